"""Per-layer error report: the fp16 production mode against the fp32 exact mode (the reference's CPU arithmetic) on one 128^3
tile of the standard `total` geometry with synthetic weights.  For every conv / transposed conv the activation the NEXT layer
consumes (InstanceNorm + LeakyReLU applied) is read back through boa_net_debug_activation in both modes.

    python tools/layer_error.py [patch] [seed]        (GPU; ~20 s)

Columns: relative RMS error = rms(a16 - a32) / rms(a32); max |error| / (max - min of a32).
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd")]

from boa_hip import plans  # noqa: E402
from boa_hip._lib import check  # noqa: E402
from boa_hip.device import Context  # noqa: E402
from boa_hip.predictor import HipPredictor  # noqa: E402


def layers(geom):
    n = len(geom.features)
    out = []
    for s in range(n):
        for c in range(geom.n_conv_enc[s]):
            out.append((0, s, c, f"enc{s}.conv{c}"))
    for k in range(n - 1):
        out.append((1, k, 0, f"up{k}"))
        for c in range(geom.n_conv_dec[k]):
            out.append((2, k, c, f"dec{k}.conv{c}"))
    return out


def activations(ctx, pred, lst):
    res = {}
    for kind, stage, conv, name in lst:
        ch, dims = C.c_int(), (C.c_int * 3)()
        rc = ctx.lib.boa_net_debug_activation(pred._net, kind, stage, conv, 0, None, C.byref(ch), dims)
        if rc != 0:
            continue
        shape = (ch.value, dims[0], dims[1], dims[2])
        buf = ctx.alloc(int(np.prod(shape)) * 4)
        check(ctx.lib.boa_net_debug_activation(pred._net, kind, stage, conv, 0, buf.vp, C.byref(ch), dims), "boa_net_debug_activation")
        res[name] = buf.download(shape, np.float32)
        buf.free()
    return res


def main():
    patch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ctx = Context(0)
    pj, dj = plans.synthetic_plans(patch=(patch,) * 3)
    cfg = plans.model_config_from_plans(pj, dj)
    geom = cfg.geometry
    blob = plans.weight_blob_from_state_dict(geom, plans.synthetic_state_dict(geom, seed))
    vol = np.random.default_rng(seed + 1).standard_normal((1, patch, patch, patch)).astype(np.float32)
    origins = np.zeros((1, 3), dtype=np.int32)
    lst = layers(geom)
    acts, logits = {}, {}
    for prec in ("fp16", "fp32"):
        p = HipPredictor(ctx, geom, max_batch=1, precision=prec)
        p.set_parameters([blob])
        logits[prec] = p.network_forward(vol, origins)[0]
        acts[prec] = activations(ctx, p, lst)
        p.close()
    print(f"{'layer':<14}{'shape':<22}{'rel RMS error':>14}{'max|err| / range':>18}")
    for _, _, _, name in lst:
        if name not in acts["fp16"]:
            continue
        a, b = acts["fp16"][name].astype(np.float64), acts["fp32"][name].astype(np.float64)
        rms = np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30)
        mx = np.abs(a - b).max() / max(b.max() - b.min(), 1e-30)
        print(f"{name:<14}{str(a.shape):<22}{rms:>14.2e}{mx:>18.2e}")
    a, b = logits["fp16"].astype(np.float64), logits["fp32"].astype(np.float64)
    print(f"{'logits':<14}{str(a.shape):<22}{np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)):>14.2e}"
          f"{np.abs(a - b).max() / (b.max() - b.min()):>18.2e}")
    print(f"label flips (argmax over {a.shape[0]} classes): {float((a.argmax(0) != b.argmax(0)).mean()):.3g}")
    ctx.close()


if __name__ == "__main__":
    main()
