#!/usr/bin/env python3
"""k_convt_deep against k_convt_mfma on the production geometry: logits must be bit-identical (same arithmetic); run twice, with
BOA_CONVT_NO_DEEP=1 the second time (subprocess), and compare the dumps."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd")]
import numpy as np
if len(sys.argv) > 1:
    from boa_hip import synthetic
    from boa_hip.device import Context
    from boa_hip.predictor import HipPredictor
    ctx = Context(0)
    tid, cfg, blob, _ = synthetic.total_part_models()[0]
    p = HipPredictor(ctx, cfg.geometry, tile_step_size=0.8, max_batch=5)
    p.set_parameters([blob])
    vol = np.random.default_rng(0).standard_normal((1, 160, 160, 224)).astype(np.float32)
    origins = np.array([[0, 0, 0], [32, 32, 96], [16, 8, 40], [32, 0, 64], [1, 2, 3]], dtype=np.int32)
    out = p.network_forward(vol, origins)
    np.save(sys.argv[1], out)
    sys.exit(0)
a, b = "/tmp/cd_a.npy", "/tmp/cd_b.npy"
subprocess.check_call([sys.executable, __file__, a])
subprocess.check_call([sys.executable, __file__, b], env=dict(os.environ, BOA_CONVT_NO_DEEP="1"))
x, y = np.load(a), np.load(b)
print("bit-identical:", np.array_equal(x.view(np.uint32), y.view(np.uint32)), "max |diff|", float(np.abs(x - y).max()), "range", float(np.ptp(x)))
