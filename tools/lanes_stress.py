#!/usr/bin/env python3
"""Stress of the multi-stream BCA pipeline: N runs of BcaPipelineHip with body_parts on a second context against the one-stream
result; prints every mismatch (volume, voxel, values).  Development aid for tests/test_gpu_lanes.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
from boa_hip import synthetic  # noqa: E402
from boa_hip.device import Context  # noqa: E402
from boa_hip.pipeline import BcaPipelineHip  # noqa: E402
import test_gpu_lanes as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mode = sys.argv[2] if len(sys.argv) > 2 else "two"       # two: parts on a second context; one: everything on ctx_b
shape = (224, 192, 256)
ct = synthetic.ct_phantom(shape, seed=11)
aff = np.diag([-1.5, -1.5, 1.5, 1.0])
ctx_a, ctx_b, ctx_c = Context(0), Context(0), Context(0)
bm = T._bca_models(2)
pipe_a = BcaPipelineHip(ctx_a, bm["body_parts"], bm["body_regions"], fast_bca=True, max_batch=8)
pipe_c = BcaPipelineHip(ctx_b, bm["body_parts"], bm["body_regions"], fast_bca=True, max_batch=8, parts_ctx=ctx_c if mode == "two" else None)
ref = pipe_a.run(ct, aff)
bad = 0
for it in range(n):
    got = pipe_c.run(ct, aff)
    for k in ("body_parts", "body_regions", "tissues"):
        d = np.argwhere(got[k] != ref[k])
        if len(d):
            bad += 1
            print(f"run {it}: {k}: {len(d)} voxels differ, first {d[:4].tolist()} got {[int(got[k][tuple(i)]) for i in d[:4]]} want {[int(ref[k][tuple(i)]) for i in d[:4]]}", flush=True)
print(f"{mode}: {n} runs, {bad} mismatching volumes")
