#!/usr/bin/env python3
"""Per-kernel FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc, separate passes) -> JSON summary.
FETCH_SIZE / WRITE_SIZE are in KB per dispatch; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced
reads (MI355X_MICROARCH.md, HBM section): fetch_bytes_corrected = 2 x FETCH_SIZE x 1024."""
import collections
import csv
import glob
import json
import sys

root, out = sys.argv[1], sys.argv[2]
command = sys.argv[3] if len(sys.argv) > 3 else "python bench.py --size 256 --steps 1 --warmup 0 --no-cpu"
git = sys.argv[4] if len(sys.argv) > 4 else None
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{root}/pmc_{c}/**/*counter_collection.csv", recursive=True)
    acc, cnt = collections.Counter(), collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k] += float(r["Counter_Value"])
            cnt[k] += 1
    for k in acc:
        res[k]["dispatches"] = cnt[k]
        res[k][f"{c}_KB_per_dispatch"] = acc[k] / cnt[k]
for k, d in res.items():
    if "FETCH_SIZE_KB_per_dispatch" in d:
        d["fetch_MB_corrected_per_dispatch"] = 2 * d["FETCH_SIZE_KB_per_dispatch"] * 1024 / 2 ** 20
    if "WRITE_SIZE_KB_per_dispatch" in d:
        d["write_MB_per_dispatch"] = d["WRITE_SIZE_KB_per_dispatch"] * 1024 / 2 ** 20
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- " + command, "git": git,
           "note": __doc__, "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_KB_per_dispatch", 0) * kv[1].get("dispatches", 0)))},
          open(out, "w"), indent=1)
print(open(out).read()[:1500])
