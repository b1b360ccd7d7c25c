#!/usr/bin/env python3
"""Randomised differential test of the measurement tables against the oracle: `total` measurements (295 regions, autochthon
reference, pulmonary fat mask, CNR-adjusted regions with fat removal + erosion) and the BCA tables (tissues, per-slice
tables, aggregation groups, descriptive statistics) on random volumes: random extents and spacings, random blob layouts of
random labels, fat pockets, empty regions, with / without median filtering.  Integers exact, floats rtol 1e-9."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
from boa_hip import bca, label_maps  # noqa: E402
from boa_hip import measurements as M  # noqa: E402
from boa_hip.device import Context  # noqa: E402
from oracle import bca as obca  # noqa: E402
from oracle import measurements as OM  # noqa: E402
from test_gpu_aggregation import _cmp  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = Context(0)
lm = dict(label_maps.CLASS_MAP_TOTAL_INV)
names = list(lm)
bad = 0
for i in range(n_cases):
    shape = tuple(int(v) for v in rng.integers(12, 60, size=3))
    spacing = tuple(float(v) for v in np.round(rng.uniform(0.6, 3.0, size=3), 2))
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    seg = np.zeros(shape, np.uint8)
    must = ["autochthon_left", "autochthon_right", "aorta"] if rng.random() < 0.8 else []
    chosen = must + [names[int(k)] for k in rng.integers(0, len(names), size=int(rng.integers(3, 25)))] + \
        [nm for nm in M.LUNG_MASKS if rng.random() < 0.6]
    for nm in chosen:
        c = [float(rng.uniform(0, s)) for s in shape]
        r = [float(rng.uniform(2, max(3, s / 3))) for s in shape]
        seg[((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2 < 1] = lm[nm]
    ct = rng.normal(40, 150, size=shape).astype(np.int16)
    for nm in must:
        sel = seg == lm[nm]
        ct[sel] = rng.normal(60, 12, size=int(sel.sum())).astype(np.int16)
    pk = tuple(slice(int(a), int(a) + int(rng.integers(1, 6))) for a in [rng.integers(0, s - 1) for s in shape])
    ct[pk] = -100
    try:
        cnr = bool(rng.random() < 0.7)
        got, fat = M.total_measurements(ctx, ct, seg, lm, spacing, cnr_adjustment=cnr)
        want, wfat = OM.total_measurements(ct, seg, lm, spacing, cnr_adjustment=cnr)
        assert np.array_equal(fat, wfat), "pulmonary fat mask differs"
        _cmp(json.loads(json.dumps(got, default=float)), json.loads(json.dumps(want, default=float)), 1e-9)
        # BCA tables on the same volume: regions / parts from the label pattern
        regions = rng.choice(np.array([0, 1, 2, 3, 4, 5, 6, 7, 9, 11, 255], dtype=np.uint8), size=(4, 4, 4))
        regions = np.kron(regions, np.ones([-(-s // 4) for s in shape], dtype=np.uint8))[:shape[0], :shape[1], :shape[2]].copy()
        parts = rng.choice(np.array([0, 1, 2, 3, 4, 5, 6], dtype=np.uint8), size=(3, 3, 3))
        parts = np.kron(parts, np.ones([-(-s // 3) for s in shape], dtype=np.uint8))[:shape[0], :shape[1], :shape[2]].copy()
        med = bool(rng.random() < 0.5)
        js, tis = bca.bca_measurements(ctx, ct, regions, parts, spacing, None, return_tissues=True, median_filtering=med)
        ref_t = obca.subclassify_tissues(ct, regions, median_filtering=med, slice_axis=0)
        assert np.array_equal(tis, ref_t), "tissues differ"
        ref = obca.bca_measurements_json(ct, regions, parts, ref_t, spacing, None)
        _cmp(json.loads(json.dumps(js, default=float)), json.loads(json.dumps(ref, default=float)), 1e-9)
        print(f"ok  case {i}: shape={shape} spacing={spacing} labels={len(np.unique(seg))} cnr={cnr} median={med}", flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"BAD case {i}: shape={shape} spacing={spacing} labels={len(np.unique(seg))}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"{n_cases} cases, {bad} failures")
ctx.close()
sys.exit(1 if bad else 0)
