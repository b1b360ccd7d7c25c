#!/usr/bin/env python3
"""Per-layer timing of one tile batch of the `total` geometry (BOA_LAYER_PROF=1 prints from the C++ driver).
   tools/layer_prof.py [batch] [total|bca]   (bca: the body_parts net of the synthetic 5 mm plans the bench uses)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd")]
os.environ["BOA_LAYER_PROF"] = "1"
import numpy as np  # noqa: E402
from boa_hip import synthetic  # noqa: E402
from boa_hip.device import Context  # noqa: E402
from boa_hip.predictor import HipPredictor  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = Context(0)
which = sys.argv[2] if len(sys.argv) > 2 else "total"
if which == "bca":
    from boa_hip import plans
    pj, dj = plans.synthetic_plans(num_classes=7, spacing=(5.0, 1.5, 1.5))
    cfg = plans.model_config_from_plans(pj, dj)
    blob = plans.weight_blob_from_state_dict(cfg.geometry, plans.synthetic_state_dict(cfg.geometry, 543))
    print("bca patch", cfg.geometry.patch_size, file=sys.stderr)
else:
    tid, cfg, blob, _ = synthetic.total_part_models()[0]
p = HipPredictor(ctx, cfg.geometry, tile_step_size=0.8, max_batch=batch, precision=os.environ.get("LAYER_PROF_PRECISION"))
p.set_parameters([blob])
ps = [int(v) for v in cfg.geometry.patch_size]
vol = np.random.default_rng(0).standard_normal((1, ps[0] + 32, ps[1] + 32, ps[2] + 96)).astype(np.float32)
if os.environ.get("LAYER_PROF_ZERO"):     # degenerate data: what the same instruction stream does when no operand bits toggle
    vol[:] = 0
base = [[0, 0, 0], [32, 32, 96], [16, 8, 40], [32, 0, 64], [0, 32, 0], [8, 8, 8], [1, 2, 3], [30, 30, 90]]
origins = np.array([base[i % 8] for i in range(batch)], dtype=np.int32)     # (batches above 8 repeat the origins)
for it in range(2):
    print(f"--- pass {it}", file=sys.stderr)
    p.network_forward(vol, origins)
p.close()
ctx.close()
