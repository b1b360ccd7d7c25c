#!/usr/bin/env python3
"""BASELINE.json configs[2] end to end on one GPU: 512x512x768 @1.5 mm, `total` + `bca` (body_parts / body_regions nets
with 5 folds at 5 mm thickness, post-processing, tissues, measurements), synthetic weights, host-to-host wall times."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd")]
import numpy as np  # noqa: E402
from boa_hip import label_maps, plans, synthetic  # noqa: E402
from boa_hip import measurements as M  # noqa: E402
from boa_hip.device import Context  # noqa: E402
from boa_hip.pipeline import BcaPipelineHip  # noqa: E402
from boa_hip.totalseg import TotalSegmentatorHip  # noqa: E402

shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 512, 768)
fast = "--fast-bca" in sys.argv
ctx = Context(0)
t0 = time.perf_counter()
ct = synthetic.ct_phantom(shape, seed=3)
aff = np.diag([-1.5, -1.5, 1.5, 1.0])
models = [(tid, cfg, [blob]) for tid, cfg, blob, _ in synthetic.total_part_models()]
bca_models = {}
for name, nc, seed in (("body_parts", 7, 543), ("body_regions", 12, 542)):
    pj, dj = plans.synthetic_plans(num_classes=nc, spacing=(5.0, 1.5, 1.5))
    cfg = plans.model_config_from_plans(pj, dj)
    bca_models[name] = (cfg, [plans.weight_blob_from_state_dict(cfg.geometry, plans.synthetic_state_dict(cfg.geometry, seed + f))
                              for f in range(1 if fast else 5)])
print(f"setup {time.perf_counter() - t0:.1f} s", flush=True)


def timed(what, fn):
    t = time.perf_counter()
    r = fn()
    ctx.sync()
    print(f"{what}: {time.perf_counter() - t:.2f} s", flush=True)
    return r


ts = TotalSegmentatorHip(ctx, models)
total = timed("total (5 models, 1000 tile forwards)", lambda: ts.predict(ct, affine=aff))
ts.close()
pipe = BcaPipelineHip(ctx, bca_models["body_parts"], bca_models["body_regions"], fast_bca=fast)
out = timed(f"bca (2 nets x {1 if fast else 5} folds, post-processing, tissues, JSON)", lambda: pipe.run(ct, aff, total_seg=total))
pipe.close()
lm = label_maps.measurement_label_map("total")
# (z,y,x) views made on the device, as compute/measurements.py does (a host transpose of 201 M voxels costs ~0.5 s each)
from boa_hip.devarray import DevArray  # noqa: E402


def _measure():
    d_f = DevArray.from_numpy(ctx, ct)
    d_ct = d_f.transpose((2, 1, 0)).contiguous(np.int16, force_copy=True)
    d_f.free()
    d_s = DevArray.from_numpy(ctx, total)
    d_lab = d_s.transpose((2, 1, 0)).contiguous(force_copy=True)
    d_s.free()
    try:
        return M.total_measurements(ctx, None, None, lm, (1.5, 1.5, 1.5), d_ct=d_ct.buf, d_lab=d_lab.buf, shape=d_ct.shape)
    finally:
        d_ct.free()
        d_lab.free()


meas = timed("total-measurements (295 regions + pulmonary fat + CNR; upload + device transposes included)", _measure)
print("labels in total:", len(np.unique(total)), "regions 255:", int((out["body_regions"] == 255).sum()),
      "tissue voxels:", int((out["tissues"] > 0).sum()), "groups:", list(out["bca_measurements"]["aggregated"])[:4])
ctx.close()
