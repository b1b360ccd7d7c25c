#!/usr/bin/env python3
"""BCA post-processing time on 512^3 label volumes: bit-mask path (csrc/ccl_bits.hip) vs byte-mask path ($BOA_MORPH_BYTES=1), on
(a) noise-like labels (what the argmax of a random-weight net gives: the worst case) and (b) the structured phantoms of
boa_hip/synthetic.py.  Prints ms per call (median of `reps`, events through the context's kernel-class timers) and checks
that both paths give the same volume."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd")]
import numpy as np  # noqa: E402
from boa_hip import bca, synthetic  # noqa: E402
from boa_hip.device import Context  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
shape = (size, size, size)
ctx = Context(0)
rng = np.random.default_rng(0)


def smooth_noise_labels(n_classes):
    """argmax of n_classes smooth random fields (a few voxels of correlation): many small components, like the synthetic nets' output"""
    from scipy import ndimage
    best = np.full(shape, -1e9, np.float32)
    lab = np.zeros(shape, np.uint8)
    for c in range(n_classes):
        f = ndimage.uniform_filter(rng.standard_normal(shape).astype(np.float32), 3)
        m = f > best
        lab[m] = c
        best = np.maximum(best, f)
    return lab


def timed(fn, d_seg):
    out = []
    res = None
    for _ in range(reps + 1):
        ctx.sync()
        t = time.perf_counter()
        res = fn(d_seg)
        ctx.sync()
        out.append((time.perf_counter() - t) * 1e3)
        if _ < reps and res is not None:
            res.free()
            res = None
    return float(np.median(out[1:])), res


cases = {
    "noise parts (7 classes)": ("parts", smooth_noise_labels(7)),
    "noise regions (12 classes)": ("regions", smooth_noise_labels(12)),
    "phantom parts": ("parts", np.ascontiguousarray(synthetic.label_phantom_parts(shape).transpose(2, 1, 0))),
    "phantom regions": ("regions", np.ascontiguousarray(synthetic.label_phantom_regions(shape).transpose(2, 1, 0))),
}
for name, (kind, seg) in cases.items():
    res = {}
    for mode in ("bits", "bytes"):
        if mode == "bytes":
            os.environ["BOA_MORPH_BYTES"] = "1"
        else:
            os.environ.pop("BOA_MORPH_BYTES", None)
        d = ctx.from_numpy(seg)
        if kind == "parts":
            ms, out = timed(lambda b: bca.postprocess_part_segmentation_device(ctx, b, shape, labels=range(1, 7)), d)
            got = out.download(shape, np.uint8)
            out.free()
        else:
            def f(b):
                bca.postprocess_region_segmentation_device(ctx, b, shape)
                return None
            # (in place: re-upload per repetition would time the copy; the filters are idempotent after the first call, so time
            #  the first call on fresh copies instead)
            times = []
            for _ in range(reps):
                d.free()
                d = ctx.from_numpy(seg)
                ctx.sync()
                t = time.perf_counter()
                f(d)
                ctx.sync()
                times.append((time.perf_counter() - t) * 1e3)
            ms = float(np.median(times))
            got = d.download(shape, np.uint8)
        d.free()
        res[mode] = (ms, got)
    same = bool((res["bits"][1] == res["bytes"][1]).all())
    print(f"{name:28s} {size}^3: bits {res['bits'][0]:8.2f} ms   bytes {res['bytes'][0]:8.2f} ms   identical {same}   "
          f"changed voxels {int((res['bits'][1] != seg).sum())}", flush=True)
ctx.close()
