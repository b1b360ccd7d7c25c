"""CPU: the oracle (oracle/) reproduces the golden vectors generated from the reference itself
(tests/golden/make_golden.py).  Bit-exact for integer / fp16-bit-pattern stages."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import bca, labels, measurements, resample, sliding_window as sw


def _npz(name):
    return np.load(os.path.join(GOLDEN, name))


def _json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def test_g1_tile_starts():
    for c in _json("g1_steps.json"):
        assert sw.compute_steps_for_sliding_window(c["size"], c["patch"], c["step"]) == c["steps"], c


def test_g2_gaussian_bits():
    z = _npz("g2_gaussian.npz")
    for k in z.files:
        ts = tuple(int(v) for v in k[2:].split("x"))
        g = sw.compute_gaussian(ts)
        assert g.dtype == np.float16
        np.testing.assert_array_equal(g.view(np.uint16), z[k])


def test_g2_gaussian_128_hash():
    meta = _json("g2_gaussian_big.json")
    for k, m in meta.items():
        ts = tuple(int(v) for v in k.split("x"))
        bits = np.ascontiguousarray(sw.compute_gaussian(ts).view(np.uint16))
        assert hashlib.sha256(bits.tobytes()).hexdigest() == m["sha256"]
        assert int(((bits & 0x7C00) == 0).sum()) == m["n_subnormal"]


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_g3_sliding_window_accumulation(case):
    z = _npz("g3_sliding_window.npz")
    tiles = z[f"{case}_tiles"]
    it = iter(range(len(tiles)))
    out = sw.predict_sliding_window_return_logits(
        lambda patch: tiles[next(it)][None], z[f"{case}_x"], [int(v) for v in z[f"{case}_patch"]],
        tiles.shape[1], float(z[f"{case}_step"]))
    np.testing.assert_array_equal(out.view(np.uint16), z[f"{case}_logits_bits"])
    np.testing.assert_array_equal(labels.argmax_labels(out), z[f"{case}_seg"])


def test_g3_sliding_window_own_conv():
    """Same, but with the tile predictions recomputed by a torch-CPU conv from the stored weights."""
    import torch
    z = _npz("g3_sliding_window.npz")
    for case in "ab":
        w, b = torch.from_numpy(z[f"{case}_w"]), torch.from_numpy(z[f"{case}_b"])
        torch.set_num_threads(1)

        def fn(p):
            return torch.nn.functional.conv3d(torch.from_numpy(p), w, b, padding=1).numpy()
        out = sw.predict_sliding_window_return_logits(fn, z[f"{case}_x"], [16, 16, 16], w.shape[0],
                                                      float(z[f"{case}_step"]))
        ref = z[f"{case}_logits_bits"].view(np.float16)
        # conv summation order may differ between runs/threads: allow 1 fp16 ulp on < 0.1 % of voxels
        neq = out.view(np.uint16) != z[f"{case}_logits_bits"]
        assert neq.mean() < 1e-3
        np.testing.assert_allclose(out.astype(np.float32), ref.astype(np.float32), rtol=2e-3, atol=1e-3)


def test_g3_no_gaussian():
    z = _npz("g3_sliding_window.npz")
    tiles = z["e_tiles"]
    it = iter(range(len(tiles)))
    out = sw.predict_sliding_window_return_logits(lambda p: tiles[next(it)][None], z["e_x"], [16, 16, 16], 2, 0.5,
                                                  use_gaussian=False)
    np.testing.assert_array_equal(out.view(np.uint16), z["e_logits_bits"])


def test_g3b_fold_ensemble():
    z = _npz("g3b_folds.npz")
    folds = [f.view(np.float16) for f in z["fold_logits_bits"]]
    np.testing.assert_array_equal(sw.ensemble_folds(folds).view(np.uint16), z["ensemble_bits"])


def test_g4_ctnorm():
    z = _npz("g4_ctnorm.npz")
    m, s, lo, hi = z["props"]
    y = labels.ct_normalize(z["x"], m, s, lo, hi)
    np.testing.assert_array_equal(y.view(np.uint32), z["y"].view(np.uint32))


def test_g5_resample():
    z = _npz("g5_resample.npz")
    for k in ["half", "twothirds", "thick", "up2", "aniso"]:
        zoom = z[f"zoom_{k}"]
        np.testing.assert_array_equal(resample.resample_img(z["ct"].astype(np.float64), zoom, 3).astype(np.int32),
                                      z[f"ct3_{k}"])
        np.testing.assert_array_equal(resample.resample_img(z["lab"].astype(np.float64), zoom, 0).astype(np.uint8),
                                      z[f"lab0_{k}"])


def test_g6_argmax_ties():
    z = _npz("g6_argmax.npz")
    np.testing.assert_array_equal(labels.argmax_labels(z["logits_bits"].view(np.float16)), z["seg"])


def test_g7_merge():
    z = _npz("g7_merge.npz")
    t = _json("g7_label_tables.json")
    inv = {v: int(k) for k, v in t["total"].items()}
    maps = [{int(k): v for k, v in t["parts"][str(tid)].items()} for tid in (291, 292, 293, 294, 295)]
    np.testing.assert_array_equal(labels.merge_parts(list(z["segs"]), maps, inv), z["combined"])


def _cmp(a, b, rtol, path=""):
    if isinstance(a, dict):
        assert set(a) == set(b), (path, set(a) ^ set(b))
        for k in a:
            _cmp(a[k], b[k], rtol, f"{path}/{k}")
    elif isinstance(a, list):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _cmp(x, y, rtol, f"{path}[{i}]")
    elif a is None or b is None or isinstance(a, (bool, str)):
        assert a == b, (path, a, b)
    else:
        assert np.isclose(a, b, rtol=rtol, atol=0), (path, a, b)


def test_g8_bca():
    z = _npz("g8_bca.npz")
    g = _json("g8_bca_measurements.json")
    tis = bca.subclassify_tissues(z["ct"], z["regions"])
    np.testing.assert_array_equal(tis, z["tissues"])
    np.testing.assert_array_equal(bca.subclassify_tissues(z["ct"], z["regions"], True, 0), z["tissues_median"])
    js = bca.bca_measurements_json(z["ct"], z["regions"], z["parts"], tis, tuple(z["spacing"]),
                                   {k: tuple(v) for k, v in g["vertebrae"].items()})
    js = json.loads(json.dumps(js, default=float))
    _cmp(js, g["json"], 1e-12)


def test_g9_measurements():
    z = _npz("g9_measurements.npz")
    g = _json("g9_measurements.json")
    am, asd = g["auto"]
    res = measurements.metrics_for_each_region(z["ct"], z["lab"], g["label_map"], am, asd, z["spacing"])
    _cmp(json.loads(json.dumps(res, default=float)), g["with_ref"], 1e-12)
    res = measurements.metrics_for_each_region(z["ct"], z["lab"], {"spleen": 1}, None, None, z["spacing"])
    _cmp(json.loads(json.dumps(res, default=float)), g["no_ref"], 1e-12)


def test_explicit_spline_zoom_matches_scipy():
    """The operation-level restatement used as blueprint for the device resampler vs scipy.ndimage.zoom."""
    from scipy import ndimage
    rng = np.random.default_rng(0)
    x = rng.integers(-1024, 1500, size=(20, 17, 23)).astype(np.float64)
    for zf in [(0.5, 0.5, 0.5), (1.0, 1.0, 0.3), (1.7, 0.8, 1.3)]:
        ref = ndimage.zoom(x, zf, order=3, mode="nearest")
        mine = resample.spline_zoom_explicit(x, ref.shape, 3)
        np.testing.assert_array_equal(mine.view(np.uint64), ref.view(np.uint64))
        lab = rng.integers(0, 9, size=x.shape).astype(np.uint8)
        np.testing.assert_array_equal(resample.spline_zoom_explicit(lab, ref.shape, 0),
                                      ndimage.zoom(lab, zf, order=0, mode="nearest"))
    # 47 -> 43 samples: the last coordinate 42 * fl(46 / 42) = 46.00000000000001 overshoots the input by one ulp; scipy does
    # not clamp it (found by tools/fuzz_voxel.py)
    x = rng.normal(size=(10, 47, 14)) * 500
    ref = ndimage.zoom(x, (0.589242864991995, 0.9166715392459119, 1.27427627251354), order=3, mode="nearest")
    assert ref.shape == (6, 43, 18)
    np.testing.assert_array_equal(resample.spline_zoom_explicit(x, ref.shape, 3).view(np.uint64), ref.view(np.uint64))


def test_g12_cropping_helpers():
    """TS get_bbox_from_mask / crop_to_bbox and nnU-Net's create_nonzero_mask, executed from the reference (G12): the
    oracle's `nonzero_bbox` and the product's host helpers (`task.get_bbox_from_mask`, `task.nonzero_bbox`) agree."""
    from boa_hip import task
    from oracle import cascade as ocas
    z = _npz("g12_cropping.npz")
    n_c, n_n = (int(v) for v in z["n_cases"])
    for i in range(n_c):
        m, ov, addon = z[f"c{i}_mask"], int(z[f"c{i}_outside"]), [int(v) for v in z[f"c{i}_addon"]]
        bbox = task.get_bbox_from_mask(m, outside_value=ov, addon=addon)
        assert bbox == z[f"c{i}_bbox"].tolist(), i
        assert ocas.get_bbox_from_mask(m, outside_value=ov, addon=addon) == bbox, i        # the cascade oracle's restatement
        np.testing.assert_array_equal(ocas.undo_crop(z[f"c{i}_crop"], m.shape, bbox)[tuple(slice(a, b) for a, b in bbox)], z[f"c{i}_crop"])
        sl = tuple(slice(a, b) for a, b in bbox)
        np.testing.assert_array_equal(z[f"c{i}_img"][sl], z[f"c{i}_crop"])
    for j in range(n_n):
        d, mask = z[f"n{j}_data"], z[f"n{j}_mask"].astype(bool)
        want = []
        for ax in range(3):
            nz = np.flatnonzero(mask.any(axis=tuple(a for a in range(3) if a != ax)))
            want.append([0, mask.shape[ax]] if nz.size == 0 else [int(nz[0]), int(nz[-1]) + 1])
        assert labels.nonzero_bbox(d) == want, j
        assert task.nonzero_bbox(d[0]) == want, j        # binary_fill_holes cannot change the bounding box


def test_g13_nnunet_resampling():
    """The reference's resample_data_or_seg_to_shape / determine_do_sep_z_and_axis / compute_new_shape (executed with the
    restated skimage resize, see make_golden.g13) against the oracle and the product's host logic."""
    from boa_hip import nnunet_resample as hnr
    from oracle import nnunet_resample as nnr
    z = _npz("g13_nnunet_resampling.npz")
    for name in z["names"]:
        meta = z[f"{name}_meta"]
        cur, new, order, sep, axis, new_shape = meta[:3], meta[3:6], int(meta[6]), bool(meta[7]), int(meta[8]), [int(v) for v in meta[9:12]]
        f16 = z[f"{name}_in"].dtype == np.uint16
        d = z[f"{name}_in"].view(np.float16) if f16 else z[f"{name}_in"]
        want = z[f"{name}_out"]
        assert list(nnr.compute_new_shape(d.shape[1:], cur, new)) == new_shape == hnr.compute_new_shape(d.shape[1:], cur, new)
        for mod in (nnr, hnr):
            s, a = mod.determine_do_sep_z_and_axis(None, cur, new)
            assert (bool(s), -1 if a is None else a) == (sep, axis), (name, mod.__name__)
        for fn in (nnr.skimage_resize, nnr.skimage_resize_explicit):
            got = nnr.resample_to_shape(d, new_shape, cur, new, order=order, resize_fn=fn)
            assert got.dtype == d.dtype
            np.testing.assert_array_equal(got.view(np.uint16) if f16 else got, want, err_msg=f"{name} {fn.__name__}")
    for row in z["decisions"]:
        cur, new = row[:3], row[3:6]
        for mod in (nnr, hnr):
            s, a = mod.determine_do_sep_z_and_axis(None, cur, new)
            assert [int(s), -1 if a is None else a] == [int(row[6]), int(row[7])]
            assert [int(v) for v in mod.compute_new_shape((37, 201, 199), cur, new)] == [int(v) for v in row[8:11]]


def test_skimage_resize_restatement_matches_scipy_grid_mode():
    """`skimage_resize_explicit` (blueprint of csrc/resample.hip k_resize_*) == scipy.ndimage.zoom(grid_mode=True,
    mode="nearest") + clip, bit for bit, 2-D and 3-D, order 1 and 3, up- and down-sampling."""
    from oracle import nnunet_resample as nnr
    rng = np.random.default_rng(3)
    for shp, osh in [((20, 17, 23), (25, 17, 30)), ((20, 17, 23), (13, 11, 9)), ((16, 31), (40, 20)), ((33, 18), (12, 27)),
                     ((9, 9, 9), (9, 14, 9)), ((5, 7), (50, 3))]:
        d = rng.standard_normal(shp) * 50
        for order in (1, 3):
            np.testing.assert_array_equal(nnr.skimage_resize_explicit(d, osh, order), nnr.skimage_resize(d, osh, order))


def test_g14_equidistant_overview():
    """create_equidistant_overview (check.py:10-36) as the reference computed it: oracle and the product's host path, exact."""
    from boa_hip import report
    from oracle import report as orep
    z = np.load(os.path.join(GOLDEN, "g14_overview.npz"))
    for i in range(3):
        img = z[f"c{i}_img"]
        segs = [(z[f"c{i}_seg{k}"], z[f"c{i}_cmap{k}"]) for k in range(2)]
        want = z[f"c{i}_out"]
        for impl in (orep.create_equidistant_overview, report.create_equidistant_overview):
            got = impl(img, segs)
            assert [r[0] for r in got] == list(z[f"c{i}_names"])
            for s_i, row in enumerate(got):
                for k in range(2):
                    assert row[1 + k].dtype == np.float64
                    np.testing.assert_array_equal(row[1 + k], want[s_i, k])


def test_find_axes_on_an_ellipse():
    """geometry.find_axes restated (ConvexHull + farthest points as the reference; minor end points without cv2): an axis-aligned
    and a rotated ellipse with known axes, within 2 pixels."""
    from boa_hip import report
    from oracle import report as orep
    yy, xx = np.mgrid[:160, :200]
    for ang, a, b in ((0.0, 70.0, 40.0), (0.5, 60.0, 30.0)):
        c, s_ = np.cos(ang), np.sin(ang)
        u, v = (xx - 100) * c + (yy - 80) * s_, -(xx - 100) * s_ + (yy - 80) * c
        m = (u / a) ** 2 + (v / b) ** 2 <= 1.0
        for impl in (orep.find_axes, report.find_axes):
            p1, p2, q1, q2 = impl(m)
            major = np.hypot(p1[0] - p2[0], p1[1] - p2[1])
            minor = np.hypot(q1[0] - q2[0], q1[1] - q2[1])
            assert abs(major - 2 * a) <= 2.5 and abs(minor - 2 * b) <= 2.5, (ang, major, minor)
        assert orep.find_axes(m) == report.find_axes(m)
