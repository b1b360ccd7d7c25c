#!/usr/bin/env python3
"""Gather-head time of a five-fold BCA net (7 classes, 154 x 512 x 512 grid at 5 mm, step 0.5): the multi-fold instantiations of
k_gather_head_pf (development aid for A/B runs with $BOA_HIP_LIB)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd")]
import numpy as np  # noqa: E402
from boa_hip import plans  # noqa: E402
from boa_hip.device import Context  # noqa: E402
from boa_hip.predictor import HipPredictor  # noqa: E402

ctx = Context(0)
pj, dj = plans.synthetic_plans(num_classes=7, spacing=(5.0, 1.5, 1.5))
cfg = plans.model_config_from_plans(pj, dj)
blobs = [plans.weight_blob_from_state_dict(cfg.geometry, plans.synthetic_state_dict(cfg.geometry, 543 + f)) for f in range(5)]
p = HipPredictor(ctx, cfg.geometry, tile_step_size=0.5, max_batch=25)
p.set_parameters(blobs)
V = [154, 512, 512]
dvol = ctx.from_numpy(np.random.default_rng(0).standard_normal((1, *V)).astype(np.float32))
lab = ctx.zeros(int(np.prod(V)))
ctx.prof_enable(True)
for it in range(3):
    ctx.prof_reset()
    p.predict_segmentation_device(dvol, V, lab)
    ctx.sync()
    pr = ctx.prof_get()
    print({k: (round(v["ms"], 2), v["launches"]) for k, v in pr.items() if v["launches"] and k in ("head_accum", "conv_mfma")}, flush=True)
print("label checksum", int(lab.download(tuple(V), np.uint8).astype(np.int64).sum()))
