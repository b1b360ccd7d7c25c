"""Synthetic inputs for benchmarks and tests: seeded int16 HU phantoms and the five `total`
part-model geometries with seeded weights.  The real weights / plans.json are downloaded at run time by the
reference (TS/libs.py:162-416) and are not available offline."""
from __future__ import annotations

import numpy as np

from . import plans

# classes incl. background of Dataset291..295 (TS/map_to_binary.py:808-959: 24/26/18/23/26 structures)
TOTAL_PART_TASK_IDS = (291, 292, 293, 294, 295)
TOTAL_PART_NUM_CLASSES = (25, 27, 19, 24, 27)


def ct_phantom(shape, seed=20260928, dtype=np.int16):
    """Body ellipsoid of soft tissue N(40,30) with a fat shell N(-100,20), bone blobs N(700,200), lung cavities
    N(-800,60), background -1024, clipped to [-1024, 3071].  Array order (x, y, z), generated slab-wise."""
    X, Y, Z = shape
    rng = np.random.default_rng(seed)
    out = np.empty(shape, dtype=dtype)
    yy = ((np.arange(Y, dtype=np.float32) - Y / 2) / (Y * 0.40))[None, :, None]
    zz = np.arange(Z, dtype=np.float32)[None, None, :]
    for x0 in range(0, X, 32):
        x1 = min(X, x0 + 32)
        xx = ((np.arange(x0, x1, dtype=np.float32) - X / 2) / (X * 0.45))[:, None, None]
        r = np.sqrt(xx * xx + yy * yy) + 0.0 * zz
        v = np.full(r.shape, -1024.0, dtype=np.float32)
        noise = rng.standard_normal(r.shape, dtype=np.float32)
        body = r < 1.0
        v = np.where(body, -100.0 + 20.0 * noise, v)
        v = np.where(r < 0.85, 40.0 + 30.0 * noise, v)
        lung = (np.sqrt((xx - 0.35) ** 2 + yy * yy) < 0.28) | (np.sqrt((xx + 0.35) ** 2 + yy * yy) < 0.28)
        lung = lung & (zz > Z * 0.55) & (zz < Z * 0.85)
        v = np.where(lung, -800.0 + 60.0 * noise, v)
        bone = (np.abs(xx) < 0.08) & (np.abs(yy - 0.6) < 0.10)
        v = np.where(bone + 0 * zz > 0, 700.0 + 200.0 * noise, v)
        out[x0:x1] = np.clip(v, -1024, 3071).astype(dtype)
    return out


def total_part_models(patch=(128, 128, 128), features=(32, 64, 128, 256, 320, 320)):
    """[(task_id, ModelConfig, weight_blob)] for the five `total` part models with seeded synthetic weights."""
    out = []
    for k, (tid, nc) in enumerate(zip(TOTAL_PART_TASK_IDS, TOTAL_PART_NUM_CLASSES)):
        pj, dj = plans.synthetic_plans(patch=patch, features=features, num_classes=nc)
        cfg = plans.model_config_from_plans(pj, dj)
        sd = plans.synthetic_state_dict(cfg.geometry, seed=tid)
        out.append((tid, cfg, plans.weight_blob_from_state_dict(cfg.geometry, sd), (pj, dj, sd)))
    return out
