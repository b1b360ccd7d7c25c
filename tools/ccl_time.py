#!/usr/bin/env python3
"""boa_ccl26 on 154 x 512 x 512 masks of several textures (development aid): time per call."""
import sys, time, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd")]
import numpy as np
from scipy import ndimage
from boa_hip.device import Context
from boa_hip._lib import check
c = Context(0)
rng = np.random.default_rng(0)
shape = (154, 512, 512)
sm = ndimage.gaussian_filter(rng.standard_normal(shape), 1.0)
masks = {"smooth half": sm > 0, "smooth sparse": sm > 0.15, "smooth dense": sm > -0.15, "noise 0.15": rng.random(shape) < 0.15,
         "noise 0.5": rng.random(shape) < 0.5, "inverse blob": ~(sm > 0.25)}
n = int(np.prod(shape))
d_r, d_s = c.alloc(n * 4), c.alloc(n * 4)
for name, m in masks.items():
    d_m = c.from_numpy(m.astype(np.uint8))
    for it in range(3):
        c.sync(); t0 = time.perf_counter()
        check(c.lib.boa_ccl26(c.h, d_m.vp, shape[0], shape[1], shape[2], d_r.vp, d_s.vp, None))
        c.sync(); t1 = time.perf_counter()
    roots = d_r.download(shape, np.int32)
    print(f"{name:14s} {1e3 * (t1 - t0):7.3f} ms   components {len(np.unique(roots[roots >= 0]))}")
    d_m.free()
c.close()
