#!/usr/bin/env python3
"""Same-box calibration of the conv class against the vendor libraries.

Times, on random fp16 data and on the SAME MI355X the bench runs on:
  * MIOpen fp16 `conv3d` (through torch-ROCm; channels-last-3d and contiguous, `cudnn.benchmark` on so MIOpen may
    pick its best solver) for the conv shapes of one tile batch of the `total` geometry (profiles/rNN_per_layer.txt);
  * hipBLASLt / rocBLAS fp16 GEMMs (torch.matmul) at the im2col shapes of the thin 128^3 layers
    (M = batch * 128^3, N = 32, K = 864 / 1728) and at a square 8192^3 shape (what the library reaches when the operands
    are GEMM-friendly), next to them.
Nothing here is product code; torch only times the vendor kernels.  Output: one table on stdout (commit it under profiles/).

    tools/power_sample.sh gpurun_out/vendor_power.txt python tools/vendor_calib.py 8 > gpurun_out/vendor_calib.txt
"""
import sys
import time

import torch
import torch.nn.functional as F

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)

# (spatial extent of the INPUT, cin, cout, stride) of the 3x3x3 convs of one PlainConvUNet forward at patch 128^3
LAYERS = [
    (128, 32, 32, 1), (128, 32, 64, 2), (64, 64, 64, 1), (64, 64, 128, 2), (32, 128, 128, 1), (32, 128, 256, 2),
    (16, 256, 256, 1), (16, 256, 320, 2), (8, 320, 320, 1), (8, 320, 320, 2), (4, 320, 320, 1),
    (8, 640, 320, 1), (16, 512, 256, 1), (32, 256, 128, 1), (64, 128, 64, 1), (128, 64, 32, 1),
]


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3        # us


print(f"# vendor calibration, batch {batch}, {reps} repetitions, device {torch.cuda.get_device_name(0)}, torch {torch.__version__}")
print("# MIOpen conv3d fp16 (random normal data, bias, no norm / activation: the bare convolution)")
print(f"{'layer':34s} {'contig us':>10s} {'TFLOP/s':>8s} {'ch-last us':>10s} {'TFLOP/s':>8s}")
for ext, cin, cout, s in LAYERS:
    x = torch.randn(batch, cin, ext, ext, ext, device=dev, dtype=torch.float16)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev, dtype=torch.float16) * 0.05
    b = torch.randn(cout, device=dev, dtype=torch.float16)
    oext = ext // s
    flop = 2.0 * batch * oext ** 3 * 27 * cin * cout
    res = []
    for cl in (False, True):
        try:
            xx = x.contiguous(memory_format=torch.channels_last_3d) if cl else x
            ww = w.contiguous(memory_format=torch.channels_last_3d) if cl else w
            with torch.no_grad():
                us = timed(lambda: F.conv3d(xx, ww, b, stride=s, padding=1), reps)
            res.append((us, flop / us * 1e-6))
        except Exception as e:                               # a solver that does not exist is a result too
            res.append((float("nan"), float("nan")))
            print("#   failed:", type(e).__name__, str(e)[:100])
    name = f"conv {ext}^3 {cin}->{cout} s{s}"
    print(f"{name:34s} {res[0][0]:10.1f} {res[0][1]:8.1f} {res[1][0]:10.1f} {res[1][1]:8.1f}", flush=True)
    del x, w, b

print("# hipBLASLt / rocBLAS fp16 GEMM (torch.matmul, fp32 accumulate), random normal data")
print(f"{'M x N x K':34s} {'us':>10s} {'TFLOP/s':>8s} {'A+B+C GB/s':>11s}")
M = batch * 128 ** 3
for m, n, k in ((M, 32, 864), (M, 32, 1728), (M // 8, 64, 1728), (8192, 8192, 8192), (65536, 320, 8640)):
    try:
        a = torch.randn(m, k, device=dev, dtype=torch.float16)
        bm = torch.randn(k, n, device=dev, dtype=torch.float16)
        with torch.no_grad():
            us = timed(lambda: torch.matmul(a, bm), max(3, reps // 2))
        by = 2.0 * (m * k + k * n + m * n)
        print(f"{m:>12d} x {n:>5d} x {k:>6d}      {us:10.1f} {2.0 * m * n * k / us * 1e-6:8.1f} {by / us * 1e-3:11.1f}", flush=True)
        del a, bm
    except Exception as e:
        print(f"{m} x {n} x {k}: failed {type(e).__name__} {str(e)[:100]}")
    torch.cuda.empty_cache()

# a sustained run of the thin layer (seconds, like BOA_LAYER_PROF_REPEAT) so that power_sample.sh sees the vendor kernel at its plateau
x = torch.randn(batch, 32, 128, 128, 128, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last_3d)
w = (torch.randn(32, 32, 3, 3, 3, device=dev, dtype=torch.float16) * 0.05).contiguous(memory_format=torch.channels_last_3d)
with torch.no_grad():
    us1 = timed(lambda: F.conv3d(x, w, None, padding=1), 3)
    n = max(10, int(3e6 / us1))
    t0 = time.time()
    us = timed(lambda: F.conv3d(x, w, None, padding=1), n)
print(f"# sustained: conv 128^3 32->32 channels-last x {n}: {us:.1f} us per launch = "
      f"{2.0 * batch * 128 ** 3 * 27 * 32 * 32 / us * 1e-6:.1f} TFLOP/s over {time.time() - t0:.1f} s")
