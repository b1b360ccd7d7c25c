#!/usr/bin/env python3
"""Randomised differential test of the network / sliding-window seams against the torch-CPU fp32 oracle: random U-Net
geometries (2-5 stages, feature widths, isotropic / anisotropic kernels and strides, patch extents, class counts), tile
batches, volumes smaller than, equal to and larger than the patch in every combination, steps 0.5 / 0.8, Gaussian on / off.
Bars as in tests/test_gpu_seams.py: max |logit error| <= 3 % of the logit range (where the summed weight is >= 1e-3) and
label agreement >= 98 %; the device argmax must equal the argmax of the device logits exactly."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
from boa_hip.device import Context  # noqa: E402
from boa_hip.predictor import HipPredictor  # noqa: E402
from oracle import sliding_window as osw  # noqa: E402
from oracle.network import network_fn_from_module  # noqa: E402
import test_gpu_seams as T  # noqa: E402

PREC = os.environ.get("FUZZ_PRECISION")      # "fp32": the split-precision mode against the same oracle, bars 0.2 % / 99.95 %
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = Context(0)
bad = 0
for i in range(n_cases):
    n_st = int(rng.integers(2, 5))
    feats = tuple(int(v) for v in [32, 64, 128, 256, 320][:n_st])
    kernels, strides, div = [], [], [1, 1, 1]
    for s_ in range(n_st):
        aniso = rng.random() < 0.3
        kernels.append([1, 3, 3] if (aniso and s_ == 0) else [3, 3, 3])
        if s_ == 0:
            strides.append([1, 1, 1])
        else:
            st = [1 if (aniso and rng.random() < 0.5) else 2, 2, 2]
            strides.append(st)
            div = [d * t for d, t in zip(div, st)]
    patch = tuple(int(d * rng.integers(2 if d >= 4 else 4, 5 if d >= 8 else 9)) for d in div)
    patch = tuple(min(p, 48) // d * d for p, d in zip(patch, div))
    classes = int(rng.integers(2, 12))
    try:
        geom, blob, net = T._small_net(patch, feats, classes, 100 + i, kernels, strides)
        step = float(rng.choice([0.5, 0.8]))
        mb = int(rng.integers(1, 5))
        shape = tuple(int(max(1, round(p * f))) for p, f in zip(patch, rng.choice([0.4, 0.7, 1.0, 1.3, 1.9], size=3)))
        vol = rng.standard_normal((1, *shape)).astype(np.float32)
        p = HipPredictor(ctx, geom, tile_step_size=step, max_batch=mb, precision=PREC)
        p.set_parameters([blob])
        got = p.predict_sliding_window_return_logits(vol)
        seg = p.predict_segmentation(vol)
        ref, nw, _ = osw.predict_sliding_window_return_logits(network_fn_from_module(net, 8), vol, list(geom.patch_size),
                                                              geom.num_classes, step, return_aux=True)
        p.close()
        g32, r32 = got.astype(np.float32), ref.astype(np.float32)
        rg = float(r32.max() - r32.min())
        okw = nw.astype(np.float32) >= 1e-3
        err = float(np.abs(g32 - r32)[:, okw].max()) if okw.any() else 0.0
        agree = float((seg == ref.argmax(0)).mean())
        exact = bool(np.array_equal(seg, got.argmax(0).astype(np.uint8)))
        ok = err <= (0.002 if PREC == "fp32" else 0.03) * rg and agree >= (0.9995 if PREC == "fp32" else 0.98) and exact and got.shape == ref.shape
        tag = "ok " if ok else "BAD"
        print(f"{tag} case {i}: stages={n_st} patch={patch} kernels0={kernels[0]} strides={strides[1:]} C={classes} vol={shape} step={step} "
              f"batch={mb}: err/range={err / max(rg, 1e-9):.4f} agree={agree:.4f} argmax_exact={exact}", flush=True)
        bad += 0 if ok else 1
    except Exception as e:  # noqa: BLE001
        print(f"BAD case {i}: stages={n_st} patch={patch} kernels={kernels} strides={strides}: {type(e).__name__}: {e}", flush=True)
        bad += 1
print(f"{n_cases} cases, {bad} failures")
ctx.close()
sys.exit(1 if bad else 0)
