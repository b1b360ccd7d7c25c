#!/usr/bin/env python3
"""Randomised bit-exact test of the tile-loop arithmetic and of the device views: (1) random volumes / patches / steps /
class counts (1..31) / with and without Gaussian / 1-3 folds / padded volumes smaller than the patch: boa_accumulate_tile +
boa_finalize_labels (fold sum, fold mean, argmax, lut, merge, crop) against the oracle's numpy statements of
predict_from_raw_data.py:483-500,611-625 -- fp16 bit patterns and labels; (2) random chains of DevArray views (transpose, flip,
slice, box, dtype conversion, scatter into a sub-box) against numpy."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
from boa_hip import sliding_window as sw  # noqa: E402
from boa_hip._lib import check  # noqa: E402
from boa_hip.devarray import DevArray  # noqa: E402
from boa_hip.device import Context  # noqa: E402
from oracle import sliding_window as osw  # noqa: E402
import test_gpu_seams as T  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = Context(0)
bad = 0
i3 = T._i3
for i in range(n_cases):
    # ---- (1) accumulate + finalize --------------------------------------------------------------------------
    patch = [int(v) for v in rng.choice([4, 6, 8, 12, 16], size=3)]
    V = [int(max(1, round(p * f))) for p, f in zip(patch, rng.choice([0.5, 1.0, 1.4, 2.3], size=3))]
    Cn = int(rng.integers(1, 32))
    step = float(rng.choice([0.5, 0.8, 1.0]))
    use_g = bool(rng.random() < 0.8)
    nf = int(rng.integers(1, 4))
    PV, below = sw.pad_amounts(V, patch)
    origins = sw.get_sliding_window_origins(PV, patch, step)
    g16 = np.ascontiguousarray(sw.compute_gaussian(tuple(patch), 1. / 8, 10)) if use_g else None
    d_g = ctx.from_numpy(g16.view(np.uint16)) if use_g else None
    nv = int(np.prod(PV))
    acc, n = ctx.alloc(Cn * nv * 2), ctx.alloc(nv * 2)
    fold = ctx.alloc(Cn * nv * 2) if nf > 1 else None
    lab = ctx.alloc(int(np.prod(V)))
    prev = rng.integers(0, 200, size=V, dtype=np.uint8)
    lab.upload(prev)
    lut = np.zeros(256, np.uint8)
    lut[:Cn] = np.concatenate([[0], rng.permutation(np.arange(1, 255))[:Cn - 1]]) if Cn > 1 else [0]
    merge = int(rng.random() < 0.5)
    o_folds = []
    try:
        for f in range(nf):
            acc.zero()
            n.zero()
            o_acc = np.zeros((Cn, *PV), np.float16)
            o_n = np.zeros(PV, np.float16)
            for o in origins:
                t = rng.normal(0, 6, size=(Cn, *patch)).astype(np.float32)
                d = ctx.from_numpy(t)
                check(ctx.lib.boa_accumulate_tile(ctx.h, d.vp, d_g.vp if d_g else None, acc.vp, n.vp, Cn, i3(patch), i3(PV), i3(o)))
                ctx.sync()
                d.free()
                osw.accumulate_tile(o_acc, o_n, t, g16, tuple(int(x) for x in o))
            assert np.array_equal(acc.download((Cn, *PV), np.uint16), o_acc.view(np.uint16)), "accumulator bits"
            assert np.array_equal(n.download(tuple(PV), np.uint16), o_n.view(np.uint16)), "n bits"
            o_folds.append(osw.finalize_logits(o_acc, o_n))
            last = f == nf - 1
            crop = (below, V) if (PV != V) else None
            T._finalize(ctx, acc, n, Cn, PV, fold=fold, fold_mode=0 if f == 0 else 1, nff=nf if (last and fold) else 0, write=0,
                        lut=lut[:Cn], merge=merge, labels=lab if last else None, crop=crop)
        want_logits = osw.ensemble_folds(o_folds) if nf > 1 else o_folds[0]
        sl = tuple(slice(b, b + v) for b, v in zip(below, V))
        am = np.argmax(want_logits[(slice(None),) + sl], axis=0)
        want = np.where(am != 0, lut[am], prev) if merge else lut[am]
        got = lab.download(tuple(V), np.uint8)
        assert np.array_equal(got, want), f"labels differ at {int((got != want).sum())} voxels"
        if fold is not None:
            assert np.array_equal(fold.download((Cn, *PV), np.uint16), want_logits.view(np.uint16)), "fold mean bits"
    except AssertionError as e:
        bad += 1
        print(f"BAD accum case {i}: V={V} patch={patch} C={Cn} step={step} gauss={use_g} folds={nf} merge={merge}: {e}", flush=True)
    for b in (acc, n, fold, lab, d_g):
        if b is not None:
            b.free()
    # ---- (2) view chains --------------------------------------------------------------------------------------
    shape = tuple(int(v) for v in rng.integers(1, 24, size=3))
    dt = rng.choice([np.uint8, np.int16, np.int32, np.float32, np.float64])
    a = (rng.normal(0, 90, size=shape)).astype(dt)
    d, ref = DevArray.from_numpy(ctx, a), a
    root = d
    desc = []
    for _ in range(int(rng.integers(1, 6))):
        op = rng.choice(["transpose", "flip", "slice"])
        if op == "transpose":
            perm = tuple(int(v) for v in rng.permutation(3))
            d, ref = d.transpose(perm), ref.transpose(perm)
        elif op == "flip":
            ax = int(rng.integers(0, 3))
            d, ref = d.flip(ax), np.flip(ref, ax)
        else:
            ax = int(rng.integers(0, 3))
            if ref.shape[ax] < 2:
                continue
            lo = int(rng.integers(0, ref.shape[ax] - 1))
            hi = int(rng.integers(lo + 1, ref.shape[ax] + 1))
            d = d.slice(ax, lo, hi)
            ref = ref[tuple(slice(lo, hi) if k == ax else slice(None) for k in range(3))]
        desc.append(op)
    odt = rng.choice([np.uint8, np.int16, np.int32, np.float32, np.float64])
    with np.errstate(invalid="ignore"):
        want = ref.astype(odt) if (np.issubdtype(odt, np.floating) or (np.all(np.abs(ref) < 120) and (odt != np.uint8 or np.all(ref >= 0)))) else None
    if want is not None:
        got = d.contiguous(odt, force_copy=True)
        ok = np.array_equal(got.download(), want)
        got.free()
        if not ok:
            bad += 1
            print(f"BAD view case {i}: shape={shape} {np.dtype(dt)}->{np.dtype(odt)} ops={desc}", flush=True)
    root.free()
print(f"{n_cases} rounds, {bad} failures")
ctx.close()
sys.exit(1 if bad else 0)
