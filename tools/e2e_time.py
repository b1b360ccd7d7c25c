#!/usr/bin/env python3
"""Host-to-host wall time of the `total` task driver (upload, canonicalise, resample, 5 models, restore, download)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd")]
import numpy as np  # noqa: E402
from boa_hip import synthetic  # noqa: E402
from boa_hip.device import Context  # noqa: E402
from boa_hip.totalseg import TotalSegmentatorHip  # noqa: E402

ctx = Context(0)
models = [(tid, cfg, [blob]) for tid, cfg, blob, _ in synthetic.total_part_models()]
ts = TotalSegmentatorHip(ctx, models)
for shape, sp in (((512, 512, 512), (1.5, 1.5, 1.5)), ((512, 512, 320), (0.75, 0.75, 2.4))):
    ct = synthetic.ct_phantom(shape, seed=1)
    aff = np.diag([-sp[0], -sp[1], sp[2], 1.0])   # LPS file
    for it in range(2):
        t0 = time.perf_counter()
        seg = ts.predict(ct, affine=aff)
        dt = time.perf_counter() - t0
    print(f"{shape} @ {sp} mm (LPS): {dt:.3f} s host-to-host, labels {len(np.unique(seg))}, seg {seg.shape} {seg.dtype}", flush=True)
ts.close()
ctx.close()
