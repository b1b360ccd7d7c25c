#!/usr/bin/env python3
"""BASELINE.json configs[4] shape on one GPU: 512x512x1024 @1.5 mm `total` -- 268 M voxels > 256*256*900 and z > 200 trigger the
reference's triple z-split (TS/nnunet.py:489-505, recombination :583-586): 3 x 100 tiles per model = 1 500 tile forwards.
Checks the split bookkeeping at full size: the middle third of the result equals predicting the middle part on its own."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd")]
import numpy as np  # noqa: E402
from boa_hip import synthetic  # noqa: E402
from boa_hip.device import Context  # noqa: E402
from boa_hip.task import split_bounds  # noqa: E402
from boa_hip.totalseg import TotalSegmentatorHip  # noqa: E402

ctx = Context(0)
models = [(tid, cfg, [blob]) for tid, cfg, blob, _ in synthetic.total_part_models()]
ts = TotalSegmentatorHip(ctx, models)
shape = (512, 512, 1024)
ct = synthetic.ct_phantom(shape, seed=4)
ct[:, :, 0] = 7  # nothing to crop: the border planes are non-zero
aff = np.diag([1.5, 1.5, 1.5, 1.0])
for it in range(2):
    t0 = time.perf_counter()
    seg = ts.predict(ct, affine=aff)
    dt = time.perf_counter() - t0
print(f"{shape}: {dt:.2f} s host-to-host (triple split, 1500 tile forwards), labels {len(np.unique(seg))}", flush=True)
parts, comb = split_bounds(shape[2])
(lo, hi) = parts[1]
mid = ts.predict(np.ascontiguousarray(ct[:, :, lo:hi]), affine=aff)
dst, src = comb[1]
ok = np.array_equal(seg[:, :, dst], mid[:, :, src])
print("middle third equals the middle part predicted on its own:", ok)
ts.close()
ctx.close()
sys.exit(0 if ok else 1)
