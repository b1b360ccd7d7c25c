#!/usr/bin/env python3
"""Randomised differential test of the conv block seam (boa_conv_block_test: MFMA conv [+ deferred InstanceNorm + LeakyReLU])
against torch-CPU fp32 on the same fp16-rounded operands: random channel counts, ragged / tiny / large extents, strides,
kernel shapes, batch sizes.  Prints the worst error per case family; exit code 1 on a tolerance violation."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
from boa_hip.device import Context  # noqa: E402
import test_gpu_seams as T  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = Context(0)
bad = 0
worst = {}
for i in range(n_cases):
    Cin = int(rng.choice([16, 32, 48, 64, 96, 128, 256, 320, 640]))
    Cout = int(rng.choice([32, 64, 96, 128, 256, 320]))
    k = (3, 3, 3) if rng.random() < 0.75 else (1, 3, 3)
    s = tuple(int(v) for v in rng.choice([1, 2], size=3, p=[0.65, 0.35]))
    if k[0] == 1:
        s = (1, s[1], s[2])
    big = rng.random() < 0.25 and Cin <= 64 and Cout <= 64
    hi = 72 if big else (24 if Cin * Cout <= 128 * 128 else 10)
    dims = tuple(int(v) for v in rng.integers(1 if rng.random() < 0.15 else 3, hi + 1, size=3))
    N = int(rng.integers(1, 4))
    norm = bool(rng.random() < 0.6)
    dout = [(d + 2 * ((kk - 1) // 2) - kk) // ss + 1 for d, kk, ss in zip(dims, k, s)]
    if norm and int(np.prod(dout)) < 2:
        norm = False                       # InstanceNorm over a single voxel is degenerate (torch raises)
    try:
        got, ref = T._conv_case(ctx, N, Cin, dims, Cout, k, s, norm=norm, seed=1000 + i)
    except Exception as e:  # noqa: BLE001
        print(f"case {i}: N={N} Cin={Cin} dims={dims} Cout={Cout} k={k} s={s} norm={norm}: EXCEPTION {type(e).__name__}: {e}")
        bad += 1
        continue
    err = np.abs(got - ref)
    tol = (6e-3 + 5e-3 * np.abs(ref)) if norm else (2e-3 + 2e-3 * np.abs(ref))
    ok = bool(np.all(err <= tol)) and np.isfinite(got).all()
    key = ("norm" if norm else "raw", "s2" if max(s) > 1 else "s1")
    worst[key] = max(worst.get(key, 0.0), float((err / tol).max()))
    if not ok:
        bad += 1
        print(f"case {i}: N={N} Cin={Cin} dims={dims} Cout={Cout} k={k} s={s} norm={norm}: max err/tol {float((err / tol).max()):.2f}")
print(f"{n_cases} cases, {bad} failures; worst err/tol per family: { {k: round(v, 3) for k, v in worst.items()} }")
ctx.close()
sys.exit(1 if bad else 0)
