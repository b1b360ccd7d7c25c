#!/usr/bin/env python3
"""Randomised differential test of the BCA pipeline glue (BcaPipelineHip.run after the networks) under random file
orientations and spacings: post-processing in file order, LPS reload, tissues, body-part flags, vertebra ranges and the
bca-measurements JSON against the oracle composition done with host numpy remaps.  Integers exact, floats rtol 1e-9."""
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
from boa_hip import label_maps  # noqa: E402
from boa_hip import orientation as o  # noqa: E402
from boa_hip.device import Context  # noqa: E402
from boa_hip.pipeline import BcaPipelineHip  # noqa: E402
from oracle import bca as obca  # noqa: E402
import test_gpu_tasks as T  # noqa: E402
from test_gpu_aggregation import _cmp  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = Context(0)
perms = list(itertools.permutations(range(3)))
inv = label_maps.CLASS_MAP_TOTAL_INV
vmap = {v[len("vertebrae_"):]: k for k, v in label_maps.CLASS_MAP_TOTAL.items() if v.startswith("vertebrae_")}
bad = 0


def lps_zyx(a, aff):
    b, _ = o.with_axcodes(a, aff, "LPS")
    return np.ascontiguousarray(b.transpose(2, 1, 0))


for i in range(n_cases):
    shape = tuple(int(v) for v in rng.integers(24, 56, size=3))
    sp = tuple(float(v) for v in np.round(rng.uniform(0.7, 2.5, size=3), 2))
    perm = perms[int(rng.integers(0, 6))]
    flips = rng.integers(0, 2, size=3)
    ornt = np.array([[perm[a], -1 if flips[a] else 1] for a in range(3)], dtype=float)
    aff_ras = np.diag([sp[0], sp[1], sp[2], 1.0])
    aff = aff_ras @ o.inv_ornt_aff(ornt, shape)
    fshape = tuple(shape[int(np.argwhere(ornt[:, 0] == a)[0, 0])] for a in range(3))
    ct = T._ct(fshape, 300 + i)
    regions = np.kron(rng.choice(np.array([0, 1, 2, 3, 4, 5, 6, 7, 9, 11], dtype=np.uint8), size=(4, 4, 4)),
                      np.ones([-(-s // 4) for s in fshape], dtype=np.uint8))[:fshape[0], :fshape[1], :fshape[2]].copy()
    regions[(rng.random(fshape) < 0.002)] = 3
    parts = np.kron(rng.choice(np.array([0, 1, 2, 3, 4, 5, 6], dtype=np.uint8), size=(3, 3, 3)),
                    np.ones([-(-s // 3) for s in fshape], dtype=np.uint8))[:fshape[0], :fshape[1], :fshape[2]].copy()
    total = np.zeros(fshape, np.uint8)
    for nm in ("vertebrae_L3", "vertebrae_T9", "vertebrae_L1"):
        a = [int(rng.integers(0, s - 6)) for s in fshape]
        total[a[0]:a[0] + 6, a[1]:a[1] + 6, a[2]:a[2] + 5] = inv[nm]
    med = bool(rng.random() < 0.5)
    try:
        fsp = o.zooms_from_affine(aff)
        mp, _ = T._model(543, 7, 543, (5.0, float(fsp[1]), float(fsp[0])))
        mr, _ = T._model(542, 12, 542, (5.0, float(fsp[1]), float(fsp[0])))
        pipe = BcaPipelineHip(ctx, (mp[1], mp[2]), (mr[1], mr[2]), fast_bca=True, max_batch=4)
        out = pipe.run(ct, aff, total_seg=total, raw_parts=parts, raw_regions=regions, median_filtering=med)
        pipe.close()
        rg = obca.postprocess_region_segmentation(np.ascontiguousarray(regions.transpose(2, 1, 0)))
        pt = obca.remove_small_labeled_objects(np.ascontiguousarray(parts.transpose(2, 1, 0)))
        assert np.array_equal(out["body_regions"].transpose(2, 1, 0), rg), "body_regions"
        assert np.array_equal(out["body_parts"].transpose(2, 1, 0), pt), "body_parts"
        rg_f, pt_f = np.ascontiguousarray(rg.transpose(2, 1, 0)), np.ascontiguousarray(pt.transpose(2, 1, 0))
        ct_l, rg_l, pt_l = lps_zyx(ct, aff), lps_zyx(rg_f, aff), lps_zyx(pt_f, aff)
        _, laff = o.with_axcodes(ct, aff, "LPS")
        lsp = tuple(float(v) for v in np.sqrt(np.sum(np.asarray(laff, dtype=np.float64)[:3, :3] ** 2, axis=0)))   # as run_pipeline's image.GetSpacing()
        tis = obca.subclassify_tissues(ct_l, rg_l, median_filtering=med, slice_axis=0)
        assert np.array_equal(lps_zyx(out["tissues"], aff), tis), "tissues"
        flags = obca.examined_body_part(rg_l, lsp)
        assert out["examined_body_part"] == flags, "flags"
        vert = obca.create_vertebrae_info(lps_zyx(total, aff), vmap, flags)
        assert out["vertebrae"] == vert, "vertebrae"
        ref = obca.bca_measurements_json(ct_l, rg_l, pt_l, tis, lsp, vert or None)
        _cmp(json.loads(json.dumps(out["bca_measurements"], default=float)), json.loads(json.dumps(ref, default=float)), 1e-9)
        print(f"ok  case {i}: file shape={fshape} axcodes={o.aff2axcodes(aff)} spacing={tuple(np.round(fsp, 2))} median={med}", flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"BAD case {i}: file shape={fshape} axcodes={o.aff2axcodes(aff)}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"{n_cases} cases, {bad} failures")
ctx.close()
sys.exit(1 if bad else 0)
