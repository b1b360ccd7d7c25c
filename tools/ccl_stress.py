#!/usr/bin/env python3
"""Two contexts / host threads run boa_ccl26 on fixed masks at the same time; every result is compared with the one-stream result
(roots must be the smallest linear index of the component: a deterministic function of the mask).  Development aid."""
import ctypes as C
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd")]
import numpy as np  # noqa: E402
from scipy import ndimage  # noqa: E402
from boa_hip._lib import check  # noqa: E402
from boa_hip.device import Context  # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 200
shape = (154, 192, 224)
rng = np.random.default_rng(3)
sm = ndimage.gaussian_filter(rng.standard_normal(shape), 2.0)
masks = [sm > 0.02, sm < 0.0, rng.random(shape) < 0.3]
n = int(np.prod(shape))


def ccl(ctx, d_m, d_r, d_s):
    check(ctx.lib.boa_ccl26(ctx.h, d_m.vp, shape[0], shape[1], shape[2], d_r.vp, d_s.vp, None), "boa_ccl26")
    return d_r.download(shape, np.int32)


ctxs = [Context(0), Context(0)]
ref = []
d_m0 = [ctxs[0].from_numpy(m.astype(np.uint8)) for m in masks]
d_r0, d_s0 = ctxs[0].alloc(n * 4), ctxs[0].alloc(n * 4)
for k in range(len(masks)):
    ref.append(ccl(ctxs[0], d_m0[k], d_r0, d_s0))
bad = [0, 0]


def lane(t):
    ctx = ctxs[t]
    ctx.bind_thread()
    d_m = [ctx.from_numpy(m.astype(np.uint8)) for m in masks]
    d_r, d_s = ctx.alloc(n * 4), ctx.alloc(n * 4)
    for it in range(n_it):
        k = (it + t) % len(masks)
        got = ccl(ctx, d_m[k], d_r, d_s)
        d = np.argwhere(got != ref[k])
        if len(d):
            bad[t] += 1
            if bad[t] <= 3:
                i = tuple(d[0])
                print(f"lane {t} it {it} mask {k}: {len(d)} voxels differ; first {i}: root {int(got[i])} want {int(ref[k][i])}; linear {np.ravel_multi_index(i, shape)}", flush=True)


which = sys.argv[2] if len(sys.argv) > 2 else "two"
stop = [False]


def conv_lane():
    """the other stream's load in the failing scenario: network tile batches (MFMA conv kernels with 137 KB LDS tiles)"""
    from boa_hip import synthetic
    from boa_hip.predictor import HipPredictor
    ctx = ctxs[1]
    ctx.bind_thread()
    tid, cfg, blob, _ = synthetic.total_part_models()[0]
    p = HipPredictor(ctx, cfg.geometry, tile_step_size=0.8, max_batch=8)
    p.set_parameters([blob])
    vol = ctx.from_numpy(np.random.default_rng(0).standard_normal((1, 160, 160, 224)).astype(np.float32))
    lab = ctx.zeros(160 * 160 * 224)
    while not stop[0]:
        p.predict_segmentation_device(vol, [160, 160, 224], lab)
        ctx.sync()
    p.close()


if which in ("post", "postparts"):
    # the BCA region post-processing (label_select -> boa_ccl26 -> boa_ccl_filter_largest, four times) on a fixed label volume
    from boa_hip import bca
    lab = np.zeros(shape, np.uint8)
    lab[sm > 0.02] = 9
    lab[(sm > 0.02) & (rng.random(shape) < 0.2)] = 5
    lab[(sm < -0.25)] = 3
    ctx = ctxs[0]
    d = ctx.from_numpy(lab)
    bca.postprocess_region_segmentation_device(ctx, d, shape)
    want = d.download(shape, np.uint8)
    if which == "post":
        tc = threading.Thread(target=conv_lane)
    else:
        def parts_lane():
            c1 = ctxs[1]
            c1.bind_thread()
            pl = np.zeros(shape, np.uint8)
            pl[sm > 0.0] = 1
            pl[sm < -0.1] = 2
            while not stop[0]:
                dd = c1.from_numpy(pl)
                out = bca.postprocess_part_segmentation_device(c1, dd, shape)
                c1.sync()
                out.free()
                dd.free()
        tc = threading.Thread(target=parts_lane)
    tc.start()
    nb = 0
    if len(sys.argv) > 3 and sys.argv[3] == "roots":      # only the labelling, on the masks of the region pipeline
        masks[:] = [lab != 0, (lab == 9) | (lab == 5), lab == 5, lab == 3]
        ref[:] = [ccl(ctx, ctx.from_numpy(m.astype(np.uint8)), d_r0, d_s0) for m in masks]
        lane(0)
        stop[0] = True
        tc.join()
        print(f"{which} roots: {n_it} iterations, {bad[0]} mismatching")
        sys.exit(0)
    for it in range(n_it):
        d.upload(lab)
        bca.postprocess_region_segmentation_device(ctx, d, shape)
        got = d.download(shape, np.uint8)
        df = np.argwhere(got != want)
        if len(df):
            nb += 1
            if nb <= 5:
                i = tuple(df[0])
                print(f"it {it}: {len(df)} voxels differ; first {i}: got {int(got[i])} want {int(want[i])} (lx {i[2] % 32}, ly {i[1] % 16}, lz {i[0] % 16})", flush=True)
    stop[0] = True
    tc.join()
    print(f"{which}: {n_it} iterations, {nb} mismatching")
    sys.exit(0)
if which == "conv":
    tc = threading.Thread(target=conv_lane)
    tc.start()
    lane(0)
    stop[0] = True
    tc.join()
    print(f"conv: {n_it} CCL iterations next to a conv stream, mismatching results {bad[0]}")
    sys.exit(0)
th = [threading.Thread(target=lane, args=(t,)) for t in ((0, 1) if which == "two" else (0,))]
for x in th:
    x.start()
for x in th:
    x.join()
print(f"{which}: {n_it} iterations per lane, mismatching results per lane {bad}")
