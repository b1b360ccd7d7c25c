#!/usr/bin/env python3
"""Per-kernel register / spill / LDS metadata of the gfx950 code objects embedded in csrc/*.o (or any hipcc -c output).

    tools/kernel_meta.py body-and-organ-analysis_amd/csrc/conv_ws.o [more.o ...] [--spills] [--check PATTERN ...]

--spills      only kernels with a non-zero spill count
--check P ... exit 1 if a kernel whose (demangled) name contains one of the patterns spills (the Makefile's gate for the hot
              instantiations: a spilled VGPR in an MFMA loop is scratch traffic on the critical path)
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("BOA_LLVM_BIN", "/opt/rocm/lib/llvm/bin").rstrip("/")
ARCH = os.environ.get("BOA_ARCH", "gfx950")


def code_object(path, tmp):
    fat = os.path.join(tmp, os.path.basename(path) + ".fatbin")
    co = os.path.join(tmp, os.path.basename(path) + ".co")
    if subprocess.call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path, os.path.join(tmp, "discard.o")],
                       stderr=subprocess.DEVNULL) != 0:
        return None        # host-only object
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}",
                           f"--input={fat}", f"--output={co}"])
    return co


def kernels(co):
    if co is None:
        return []
    notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    out = []
    for blk in re.split(r"\n\s+- ", notes):
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name or ".vgpr_count" not in blk:
            continue
        g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1)) if re.search(rf"\.{k}:\s+(\d+)", blk) else 0
        out.append(dict(name=name.group(1), vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"), vspill=g("vgpr_spill_count"),
                        sspill=g("sgpr_spill_count"), lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size")))
    names = subprocess.run(["c++filt"] + [k["name"] for k in out], capture_output=True, text=True).stdout.split("\n")
    for k, n in zip(out, names):
        k["pretty"] = re.sub(r"\(.*", "", n.replace("void ", ""))
    return out


def main():
    args = sys.argv[1:]
    only_spills = "--spills" in args
    pats = []
    if "--check" in args:
        i = args.index("--check")
        pats = args[i + 1:]
        args = args[:i]
    files = [a for a in args if not a.startswith("--")]
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            for k in kernels(code_object(f, tmp)):
                if only_spills and not (k["vspill"] or k["sspill"]):
                    continue
                hot = any(p in k["pretty"] for p in pats)
                flag = ""
                if hot and k["vspill"]:
                    flag = "   <-- SPILLS (hot kernel)"
                    bad += 1
                if pats and not hot and not k["vspill"]:
                    continue
                print(f"{os.path.basename(f):16s} {k['pretty'][:70]:70s} vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} sgpr {k['sgpr']:3d} "
                      f"vspill {k['vspill']:3d} sspill {k['sspill']:3d} scratch {k['scratch']:5d} lds {k['lds']:6d}{flag}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
