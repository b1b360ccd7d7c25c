#!/usr/bin/env python3
"""Time of the label path of one 512^3 part model (conv stack + gather head) per kernel class (development aid)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd")]
import numpy as np
from boa_hip import synthetic
from boa_hip.device import Context
from boa_hip.predictor import HipPredictor
ctx = Context(0)
tid, cfg, blob, _ = synthetic.total_part_models()[0]
p = HipPredictor(ctx, cfg.geometry, tile_step_size=0.8, max_batch=25)
p.set_parameters([blob])
x = np.random.default_rng(0).standard_normal((1, 512, 512, 512)).astype(np.float32)
dvol = ctx.from_numpy(x)
lab = ctx.zeros(512 ** 3)
ctx.prof_enable(True)
for it in range(2):
    ctx.prof_reset()
    p.predict_segmentation_device(dvol, [512, 512, 512], lab)
    ctx.sync()
    pr = ctx.prof_get()
    print({k: (round(v['ms'], 2), v['launches']) for k, v in pr.items() if v['launches']})
