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
if mode == "full":     # the whole total+bca runner, two and three lanes, labels and tables
    from boa_hip import label_maps
    from boa_hip.devarray import DevArray
    from boa_hip.lanes import TotalBcaRunner
    from boa_hip.task import SegmentationTask
    lm = label_maps.measurement_label_map("total")
    parts = [(tid, cfg, [blob]) for tid, cfg, blob, _ in synthetic.total_part_models()]
    total = SegmentationTask(ctx_a, "total", parts, resample=1.5, multimodel=True, max_batch=8)
    pipe_b = BcaPipelineHip(ctx_b, bm["body_parts"], bm["body_regions"], fast_bca=True, max_batch=8)
    pipe_3 = BcaPipelineHip(ctx_b, bm["body_parts"], bm["body_regions"], fast_bca=True, max_batch=8, parts_ctx=ctx_c)
    d_ct = DevArray.from_numpy(ctx_a, ct)
    want = T._collect(TotalBcaRunner(total, pipe_a, lm).run_resident(d_ct, aff))
    bad = 0
    for it in range(n):
        run = TotalBcaRunner(total, pipe_b if it % 2 == 0 else pipe_3, lm)
        got = T._collect(run.run_resident(d_ct, aff))
        for k in want[0]:
            d = np.argwhere(got[0][k] != want[0][k])
            if len(d):
                bad += 1
                print(f"run {it} ({2 + it % 2} lanes): {k}: {len(d)} voxels differ, first {d[:3].tolist()}", flush=True)
        for j, name in ((1, "total measurements"), (2, "bca measurements"), (3, "vertebrae")):
            if got[j] != want[j]:
                bad += 1
                print(f"run {it}: {name} differ", flush=True)
    print(f"full: {n} runs, {bad} mismatches")
    sys.exit(0)
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
