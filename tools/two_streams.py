#!/usr/bin/env python3
"""Two volumes in flight on one GPU: two contexts (streams) driven by two host threads.  Prints volumes/s for 1 and 2
concurrent 512^3 `total` volumes (the kernels of the two streams fill each other's launch tails and the thin deep layers)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd")]
import numpy as np  # noqa: E402
from boa_hip import label_maps, synthetic  # noqa: E402
from boa_hip._lib import check  # noqa: E402
from boa_hip.device import Context  # noqa: E402
from boa_hip.predictor import HipPredictor  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
shape = [size] * 3
nvox = size ** 3
models = synthetic.total_part_models()


class Lane:
    def __init__(self, seed):
        self.ctx = Context(0)
        self.preds = []
        for tid, cfg, blob, _ in models:
            p = HipPredictor(self.ctx, cfg.geometry, tile_step_size=0.8, max_batch=8)
            p.set_parameters([blob])
            p._ensure_net(0)
            self.preds.append((tid, p))
        self.d_ct = self.ctx.from_numpy(synthetic.ct_phantom(shape, seed=seed))
        self.d_vol, self.d_lab = self.ctx.alloc(nvox * 4), self.ctx.alloc(nvox)
        self.work = {}
        self.ip = models[0][1].intensity_properties["0"]

    def step(self):
        c, ip = self.ctx, self.ip
        check(c.lib.boa_ct_normalize(c.h, self.d_ct.vp, 0, self.d_vol.vp, nvox, ip["mean"], ip["std"], ip["percentile_00_5"],
                                     ip["percentile_99_5"]))
        self.d_lab.zero()
        for tid, p in self.preds:
            p.predict_segmentation_device(self.d_vol, shape, self.d_lab, lut=label_maps.part_lut(tid), merge=True, work=self.work)

    def run(self, k):
        for _ in range(k):
            self.step()
        self.ctx.sync()


lanes = [Lane(20260928), Lane(20260929)]
for ln in lanes:
    ln.run(1)
t = time.perf_counter()
lanes[0].run(steps)
one = steps / (time.perf_counter() - t)
t = time.perf_counter()
th = [threading.Thread(target=ln.run, args=(steps,)) for ln in lanes]
for x in th:
    x.start()
for x in th:
    x.join()
two = 2 * steps / (time.perf_counter() - t)
print(f"{size}^3 total: 1 volume in flight {one:.3f} volumes/s, 2 in flight {two:.3f} volumes/s ({two / one:.2f}x)")
