#!/usr/bin/env python3
"""Matrix-core utilisation per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
GRBM_GUI_ACTIVE, SQ_INSTS_VALU_MFMA_MOPS_F16) over tools/layer_prof.py (one 8-tile batch of the `total` geometry).

SQ_VALU_MFMA_BUSY_CYCLES counts cycles a SIMD's matrix pipe is busy, summed over the SIMDs the counter samples
(MI355X_MICROARCH.md: = 32 x N_mfma for 32x32x16 f16); GRBM_GUI_ACTIVE counts the cycles the kernel occupied the GPU,
summed over the XCDs.  mfma_util = MFMA_BUSY / (GUI_ACTIVE / n_xcd x n_simd) is reported with the normalisation spelled out;
`mfma_busy_per_mfma_instruction` (should be ~32 cycles x SIMD sampling factor) is the self-check of that normalisation."""
import collections
import csv
import glob
import json
import sys

root, out = sys.argv[1], sys.argv[2]
git = sys.argv[3] if len(sys.argv) > 3 else None
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
N_XCD, N_SIMD = 8, 1024
res = {}
for k, d in acc.items():
    n = max(cnt[(k, c)] for c in d)
    e = {"dispatches": n, **{c: v / cnt[(k, c)] for c, v in d.items()}}
    gui, busy = e.get("GRBM_GUI_ACTIVE"), e.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if gui and busy is not None:
        e["mfma_util_if_gui_summed_over_xcds"] = busy / (gui / N_XCD * N_SIMD)
        e["mfma_util_if_gui_is_per_device"] = busy / (gui * N_SIMD)
    res[k] = e
json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 -- python tools/layer_prof.py 8",
           "git": git, "note": __doc__,
           "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) * kv[1]["dispatches"]))},
          open(out, "w"), indent=1)
print(open(out).read()[:2500])
