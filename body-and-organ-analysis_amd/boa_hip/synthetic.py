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


# ---- structured label phantoms ("a structured phantom (nested ellipsoids -> 117 labels) so histograms are
# non-degenerate") -- label volumes with the TOPOLOGY of real segmentations (a few large compact components per label, nesting,
# smooth boundaries, background around the body) for the aggregation / morphology benchmarks.  The argmax of a random-weight net is
# noise-like (millions of components): the worst case of every component filter and nothing a real volume looks like
# (BOA/compute/measurements.py:203-241, BCA/body_parts/postprocess.py:7-52, BCA/body_regions/postprocess.py:8-40 all see compact
# organs).  Array order (x, y, z) like ct_phantom; deterministic (seeded), generated slab-wise.
def _norm_coords(shape, x0, x1):
    X, Y, Z = shape
    xx = ((np.arange(x0, x1, dtype=np.float32) + 0.5) / X * 2 - 1)[:, None, None]
    yy = ((np.arange(Y, dtype=np.float32) + 0.5) / Y * 2 - 1)[None, :, None]
    zz = ((np.arange(Z, dtype=np.float32) + 0.5) / Z * 2 - 1)[None, None, :]
    return xx, yy, zz


def label_phantom_total(shape, n_labels=117, seed=20260929):
    """uint8 (x, y, z): `n_labels` organs = ellipsoids inside the body ellipse of ct_phantom, painted from the largest to the smallest
    so that small structures sit inside / on top of large ones (nested); ~36 % of the volume is labelled, a label is one or a few compact
    components of 1e-4 ... 3e-2 of the volume."""
    X, Y, Z = shape
    rng = np.random.default_rng(seed)
    n = int(n_labels)
    # radii (fractions of the half extents) from large organs to small vessels / vertebrae; centres inside the body ellipse
    rad = np.sort(rng.uniform(0.03, 1.0, n) ** 2.2)[::-1] * 0.42 + 0.035
    cen = np.empty((n, 3), np.float32)
    k = 0
    while k < n:
        c = rng.uniform(-0.8, 0.8, 3)
        if (c[0] / 0.9) ** 2 + (c[1] / 0.8) ** 2 < 0.75:
            cen[k] = c
            k += 1
    asp = rng.uniform(0.6, 1.6, (n, 3)).astype(np.float32)
    order = rng.permutation(n) + 1                     # label values are not sorted by size
    out = np.zeros(shape, np.uint8)
    ax = [((np.arange(d, dtype=np.float32) + 0.5) / d * 2 - 1) for d in (X, Y, Z)]

    def paint(c, r, value, clip_body):
        # only the ellipsoid's bounding box is evaluated (the work is the sum of the boxes, not labels x volume)
        lo = [int(np.searchsorted(ax[a], c[a] - r[a], "left")) for a in range(3)]
        hi = [int(np.searchsorted(ax[a], c[a] + r[a], "right")) for a in range(3)]
        if min(h - l for l, h in zip(lo, hi)) <= 0:
            return
        xx = ax[0][lo[0]:hi[0], None, None]
        yy = ax[1][None, lo[1]:hi[1], None]
        zz = ax[2][None, None, lo[2]:hi[2]]
        m = ((xx - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((zz - c[2]) / r[2]) ** 2 < 1.0
        if clip_body:
            m = m & ((xx / 0.9) ** 2 + (yy / 0.8) ** 2 < 1.0)
        out[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]][m] = value

    for i in range(n):
        paint(cen[i], rad[i] * asp[i], order[i], True)
    for i in range(n):      # a small core of every organ survives whatever was painted over it: all n labels occur
        paint(cen[i], np.full(3, 0.03, np.float32), order[i], False)
    return out


def label_phantom_parts(shape):
    """uint8 (x, y, z) body_parts labels (BCA/body_parts/definition.py:4-11): torso 1, head 2, legs 3 / 4, arms 5 / 6 -- z is the
    body axis (head at high z), x left-right."""
    X, Y, Z = shape
    out = np.zeros(shape, np.uint8)
    for x0 in range(0, X, 32):
        x1 = min(X, x0 + 32)
        xx, yy, zz = _norm_coords(shape, x0, x1)
        slab = np.zeros((x1 - x0, Y, Z), np.uint8)
        torso = ((xx / 0.55) ** 2 + (yy / 0.45) ** 2 < 1.0) & (zz > -0.25) & (zz < 0.62)
        head = (xx / 0.28) ** 2 + (yy / 0.32) ** 2 + ((zz - 0.8) / 0.2) ** 2 < 1.0
        neck = ((xx / 0.14) ** 2 + (yy / 0.14) ** 2 < 1.0) & (zz >= 0.6) & (zz < 0.7)
        for s, lab in ((-1, 3), (1, 4)):
            leg = (((xx - s * 0.27) / 0.2) ** 2 + (yy / 0.22) ** 2 < 1.0) & (zz <= -0.25)
            slab[leg] = lab
        for s, lab in ((-1, 5), (1, 6)):
            arm = (((xx - s * 0.75) / 0.13) ** 2 + (yy / 0.15) ** 2 < 1.0) & (zz > -0.35) & (zz < 0.55)
            slab[arm] = lab
        slab[torso] = 1
        slab[head | neck] = 2
        out[x0:x1] = slab
    return out


def label_phantom_regions(shape):
    """uint8 (x, y, z) body_regions labels (BCA/body_regions/definition.py:4-15), nested as in a body: subcutaneous shell 1 around
    muscle 2 around the abdominal cavity 3 / thoracic cavity 4 (with mediastinum 9 and pericardium 7 inside), bone 5 (spine, two
    femurs), glands 6, breast implant 8, brain 10, nervous system 11 (spinal canal inside the spine)."""
    X, Y, Z = shape
    out = np.zeros(shape, np.uint8)
    for x0 in range(0, X, 32):
        x1 = min(X, x0 + 32)
        xx, yy, zz = _norm_coords(shape, x0, x1)
        slab = np.zeros((x1 - x0, Y, Z), np.uint8)
        r2 = (xx / 0.9) ** 2 + (yy / 0.8) ** 2 + 0 * zz
        slab[r2 < 1.0] = 1
        slab[r2 < 0.78] = 2
        slab[(r2 < 0.5) & (zz > -0.55) & (zz < 0.05)] = 3
        thor = (r2 < 0.5) & (zz >= 0.12) & (zz < 0.62)
        slab[thor] = 4
        slab[thor & ((xx / 0.22) ** 2 + ((yy + 0.05) / 0.3) ** 2 < 1.0)] = 9
        slab[((xx + 0.05) / 0.16) ** 2 + ((yy + 0.08) / 0.18) ** 2 + ((zz - 0.3) / 0.12) ** 2 < 1.0] = 7
        slab[((xx / 0.09) ** 2 + ((yy - 0.5) / 0.1) ** 2 < 1.0) & (zz > -0.6) & (zz < 0.75)] = 5          # spine
        slab[((xx / 0.035) ** 2 + ((yy - 0.5) / 0.04) ** 2 < 1.0) & (zz > -0.6) & (zz < 0.75)] = 11        # spinal canal
        for s in (-1, 1):
            slab[(((xx - s * 0.3) / 0.06) ** 2 + (yy / 0.07) ** 2 < 1.0) & (zz <= -0.6)] = 5               # femurs
            slab[((xx - s * 0.25) / 0.07) ** 2 + ((yy + 0.1) / 0.06) ** 2 + ((zz + 0.15) / 0.07) ** 2 < 1.0] = 6   # glands (kidney-like)
        slab[((xx - 0.4) / 0.12) ** 2 + ((yy + 0.55) / 0.1) ** 2 + ((zz - 0.35) / 0.1) ** 2 < 1.0] = 8     # implant
        slab[(xx / 0.3) ** 2 + (yy / 0.36) ** 2 + ((zz - 0.86) / 0.12) ** 2 < 1.0] = 10                    # brain
        out[x0:x1] = slab
    return out
