"""GPU parity tests of the body-composition / HU-measurement kernels (through the C ABI) against the golden
vectors produced by the reference (G8, G9) and against the oracle.  Integer results are bit-exact; floating
point statistics carry the tolerance stated in each test."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from boa_hip.device import Context
    c = Context(0)
    yield c
    c.close()


def _npz(name):
    return np.load(os.path.join(GOLDEN, name))


def _json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


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


def test_g8_tissues_and_bca_json(ctx):
    """tissue map bit-exact; bca-measurements.json numeric content: slice volumes exact (count x constant),
    pandas describe() statistics rtol 1e-12, mean HU rtol 1e-9 (exact integer sums vs numpy pairwise fp64)."""
    from boa_hip import bca
    z = _npz("g8_bca.npz")
    g = _json("g8_bca_measurements.json")
    js, tis = bca.bca_measurements(ctx, z["ct"], z["regions"], z["parts"], tuple(z["spacing"]),
                                   {k: tuple(v) for k, v in g["vertebrae"].items()}, return_tissues=True)
    np.testing.assert_array_equal(tis, z["tissues"])
    js = json.loads(json.dumps(js, default=float))
    _cmp(js["slices"], g["json"]["slices"], 1e-15)
    _cmp(js["slices_no_extremities"], g["json"]["slices_no_extremities"], 1e-15)
    _cmp(js, g["json"], 1e-9)


def test_tissue_counts_random_vs_oracle(ctx):
    """Ragged (non multiple-of-8) slice size -> scalar path; all label / HU combinations incl. edges of the HU windows."""
    from boa_hip import bca
    from oracle import bca as obca
    rng = np.random.default_rng(0)
    shape = (7, 13, 11)
    ct = rng.choice(np.array([-1001, -1000, -191, -190, -31, -30, -29, 0, 150, 151, 3000, 3001], np.int16), size=shape)
    regions = rng.integers(0, 12, size=shape).astype(np.uint8)
    parts = rng.integers(0, 4, size=shape).astype(np.uint8)
    d = [ctx.from_numpy(a) for a in (ct, regions, parts)]
    tis, counts, sums = bca.tissue_aggregate(ctx, d[0], d[1], d[2], shape)
    t = tis.download(shape, np.uint8)
    ref = obca.subclassify_tissues(ct, regions)
    np.testing.assert_array_equal(t, ref)
    for a, m in ((0, np.ones(shape, bool)), (1, parts == 1)):
        for k in range(1, 8):
            sel = (ref == k) & m
            np.testing.assert_array_equal(counts[:, a, k], sel.sum(axis=(1, 2)))
            np.testing.assert_array_equal(sums[:, a, k], np.where(sel, ct.astype(np.int64), 0).sum(axis=(1, 2)))


def test_g9_metrics_for_each_region(ctx):
    """Per-label HU statistics from one histogram pass vs the reference's metrics_for_each_region: counts / min /
    max / median / percentiles / mean exact, std and cnr rtol 1e-9."""
    from boa_hip import measurements as M
    z = _npz("g9_measurements.npz")
    g = _json("g9_measurements.json")
    am, asd = g["auto"]
    res = M.metrics_for_each_region(ctx, z["ct"], z["lab"], g["label_map"], am, asd, z["spacing"])
    res = json.loads(json.dumps(res, default=float))
    for region, want in g["with_ref"].items():
        got = res[region]
        assert got["present"] == want["present"]
        if not want["present"]:
            continue
        for k in ("volume_ml", "min_hu", "max_hu", "median_hu", "25th_percentile_hu", "75th_percentile_hu", "mean_hu"):
            assert got[k] == want[k], (region, k, got[k], want[k])
        assert np.isclose(got["std_hu"], want["std_hu"], rtol=1e-9, atol=0)
        assert np.isclose(got["cnr"], want["cnr"], rtol=1e-9, atol=0)
    res2 = M.metrics_for_each_region(ctx, z["ct"], z["lab"], {"spleen": 1}, None, None, z["spacing"])
    assert res2["spleen"]["cnr"] is None and res2["spleen"]["mean_hu"] == g["no_ref"]["spleen"]["mean_hu"]


def test_erosion_vs_oracle(ctx):
    """Separable 6^3 (end-padded) erosion == scipy binary_erosion restatement of erode_region, incl. borders."""
    from boa_hip import measurements as M
    from oracle import measurements as OM
    rng = np.random.default_rng(1)
    for shape, p in [((20, 24, 28), 0.97), ((9, 31, 17), 0.995), ((6, 6, 6), 1.0), ((5, 40, 40), 0.99)]:
        m = rng.random(shape) < p
        np.testing.assert_array_equal(M.erode_region(ctx, m), OM.erode_region(m))
    m = rng.random((16, 16, 16)) < 0.98
    np.testing.assert_array_equal(M.erode_region(ctx, m, 3), OM.erode_region(m, 3))
    # rows that span several mask words and end inside one (the bit-mask passes carry bits across word boundaries), other footprints
    for shape, p, k in [((7, 9, 70), 0.99, 6), ((4, 5, 100), 0.98, 5), ((3, 33, 65), 0.99, 2), ((6, 6, 97), 0.995, 9), ((2, 3, 33), 1.0, 6),
                        ((3, 4, 64), 0.97, 1)]:
        m = rng.random(shape) < p
        np.testing.assert_array_equal(M.erode_region(ctx, m, k), OM.erode_region(m, k), err_msg=f"{shape} k={k}")


def test_total_measurements_vs_oracle(ctx):
    """compute_measurements (`total` branch: autochthon reference, all labels, ct_pfav, CNR-adjusted regions)."""
    from boa_hip import label_maps
    from boa_hip import measurements as M
    from oracle import measurements as OM
    rng = np.random.default_rng(2)
    shape = (40, 56, 64)
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    seg = np.zeros(shape, np.uint8)
    lm = dict(label_maps.CLASS_MAP_TOTAL_INV)

    def blob(label, c, r):
        seg[((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2 < 1] = label
    blob(lm["autochthon_left"], (20, 40, 20), (16, 9, 9))
    blob(lm["autochthon_right"], (20, 40, 44), (16, 9, 9))
    blob(lm["aorta"], (20, 20, 32), (18, 7, 7))
    blob(lm["spleen"], (10, 12, 12), (6, 6, 6))
    for i, nm in enumerate(M.LUNG_MASKS):
        blob(lm[nm], (8 + 6 * i, 10, 50), (4, 5, 6))
    ct = rng.normal(40, 120, size=shape).astype(np.int16)
    ct[seg == lm["lung_upper_lobe_left"]] = rng.normal(-120, 60, size=(seg == lm["lung_upper_lobe_left"]).sum()).astype(np.int16)
    for nm in ("autochthon_left", "autochthon_right", "aorta"):  # homogeneous organs: the eroded masks are non-empty
        sel = seg == lm[nm]
        ct[sel] = rng.normal(60, 12, size=sel.sum()).astype(np.int16)
    ct[18:22, 38:42, 18:22] = -100  # a fat pocket inside the left autochthon (removed before the erosion)
    spacing = (0.8, 0.8, 2.0)
    got, fat = M.total_measurements(ctx, ct, seg, lm, spacing, cnr_adjustment=True)
    want, wfat = OM.total_measurements(ct, seg, lm, spacing, cnr_adjustment=True)
    np.testing.assert_array_equal(fat, wfat)
    assert got["info"]["autochthon_mean"] is not None and got["cnr_adjusted"]["aorta"]["present"]
    got = json.loads(json.dumps(got, default=float))
    want = json.loads(json.dumps(want, default=float))
    _cmp(got, want, 1e-9)


def test_ccl_and_region_postprocess_vs_oracle(ctx):
    """26-connected components by device union-find + largest-component filters vs scipy.ndimage.label."""
    from boa_hip import bca
    from oracle import bca as obca
    rng = np.random.default_rng(3)
    shape = (24, 40, 36)
    seg = np.zeros(shape, np.uint8)
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    seg[(yy - 20) ** 2 + (xx - 18) ** 2 < 15 ** 2] = 1
    seg[(yy - 20) ** 2 + (xx - 18) ** 2 < 11 ** 2] = 4
    seg[((zz - 12) ** 2 + (yy - 20) ** 2 + (xx - 18) ** 2) < 6 ** 2] = 9
    seg[((zz - 12) ** 2 + (yy - 20) ** 2 + (xx - 18) ** 2) < 3 ** 2] = 7
    seg[2:5, 2:5, 2:5] = 7          # second pericardium blob -> 255
    seg[20:22, 34:36, 30:33] = 3    # abdominal blobs
    seg[5:12, 30:38, 2:9] = 3
    seg[0, 0, 0] = 2                # isolated foreground voxel, diagonal neighbour joins it to the cube at [2:5]? no -> separate
    seg[1, 1, 1] = 2                # 26-connected to [0,0,0] and to [2,2,2]
    noise = rng.random(shape) < 0.002
    seg[noise & (seg == 0)] = 6
    got = bca.postprocess_region_segmentation(ctx, seg)
    want = obca.postprocess_region_segmentation(seg)
    np.testing.assert_array_equal(got, want)
    assert (got == 255).sum() > 0


@pytest.mark.parametrize("shape", [(33, 35, 70), (16, 16, 32), (17, 49, 33), (5, 3, 100), (48, 32, 64), (1, 1, 7)])
def test_ccl26_roots_sizes_vs_scipy(ctx, shape):
    """boa_ccl26 (tiles labelled in LDS, unions across tile faces in global memory): roots = smallest linear index of the
    26-connected component, sizes[root] = its voxel count, for noise of several densities and blobs, on shapes that cut the
    32 x 16 x 16 tiles raggedly."""
    import ctypes as C
    from scipy import ndimage
    rng = np.random.default_rng(sum(shape))
    n = int(np.prod(shape))
    masks = [rng.random(shape) < p for p in (0.03, 0.15, 0.4, 0.75, 0.97)] + [_blobs(rng, shape, 25, max(2, min(shape) // 2 + 2)),
                                                                            np.ones(shape, bool), np.zeros(shape, bool)]
    d_roots, d_sizes = ctx.alloc(n * 4), ctx.alloc(n * 4)
    for m in masks:
        d_m = ctx.from_numpy(m.astype(np.uint8))
        ncomp = C.c_int()
        from boa_hip._lib import check
        check(ctx.lib.boa_ccl26(ctx.h, d_m.vp, shape[0], shape[1], shape[2], d_roots.vp, d_sizes.vp, C.byref(ncomp)), "boa_ccl26")
        roots = d_roots.download(shape, np.int32)
        sizes = d_sizes.download((n,), np.uint32)
        d_m.free()
        lab, k = ndimage.label(m, structure=np.ones((3, 3, 3)))
        assert ncomp.value == k
        assert (roots[~m] == -1).all()
        if k:
            idx = np.arange(n).reshape(shape)
            first = ndimage.minimum(idx, lab, index=np.arange(1, k + 1)).astype(np.int64)      # smallest index per component
            np.testing.assert_array_equal(roots[m], first[lab[m] - 1])
            cnt = np.bincount(lab[m] - 1, minlength=k)
            np.testing.assert_array_equal(sizes[first], cnt)
            assert int(sizes.sum()) == int(m.sum())
    d_roots.free()
    d_sizes.free()


def _blobs(rng, shape, n_blobs, rmax):
    """random union of balls/holes: enough structure for nested contours, holes, small objects"""
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    m = np.zeros(shape, bool)
    for _ in range(n_blobs):
        c = [rng.integers(0, s) for s in shape]
        r = rng.integers(1, rmax)
        ball = (zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2 <= r * r
        if rng.random() < 0.35:
            m &= ~ball
        else:
            m |= ball
    return m


def test_fill_holes_2d_vs_scipy(ctx):
    """Slice-wise external-contour fill == binary_fill_holes with the 4-connected (cross) background, incl. diagonal
    8-connected rings (closed for the fill) and shapes touching the border."""
    import ctypes as C
    from scipy import ndimage
    from boa_hip._lib import check
    rng = np.random.default_rng(11)
    shape = (9, 61, 53)
    m = _blobs(rng, shape, 60, 12)
    m[0] = False
    m[0, 10, 10] = m[0, 11, 11] = m[0, 10, 12] = m[0, 9, 11] = True          # diamond ring: centre (10,11) is enclosed
    m[1] = rng.random(shape[1:]) < 0.45                                        # noise
    m[2] = True; m[2, 5:20, 0:7] = False; m[2, 30:40, 20:30] = False           # hole open to the border + closed hole
    n = m.size
    d_m = ctx.from_numpy(m.astype(np.uint8))
    d_i, d_t, d_o = ctx.alloc(n * 4), ctx.alloc(n), ctx.alloc(n)
    check(ctx.lib.boa_fill_holes_2d(ctx.h, d_m.vp, shape[0], shape[1], shape[2], d_i.vp, d_t.vp, d_o.vp))
    out = d_o.download(shape, np.uint8).astype(bool)
    ref = np.stack([ndimage.binary_fill_holes(m[i]) for i in range(shape[0])])
    np.testing.assert_array_equal(out, ref)
    assert out[0, 10, 11] and ref.sum() > m.sum()


def test_postprocess_part_segmentation_vs_oracle(ctx):
    from boa_hip import bca
    from oracle import bca as obca
    rng = np.random.default_rng(12)
    shape = (24, 72, 64)
    seg = np.zeros(shape, np.uint8)
    for label in (1, 2, 3, 5):
        seg[_blobs(rng, shape, 25, 14)] = label
    seg[rng.random(shape) < 0.01] = 4                                           # specks: removed as small objects
    for thr in (3000, 400):
        out = bca.postprocess_part_segmentation(ctx, seg, threshold=thr)
        ref = obca.remove_small_labeled_objects(seg, threshold=thr)
        np.testing.assert_array_equal(out, ref)
    assert (ref != seg).any() and (ref > 0).any()


@pytest.mark.parametrize("flat_axis", [0, 1, 2])
def test_median3_inplane_vs_scipy(ctx, flat_axis):
    from scipy import ndimage
    from boa_hip import bca
    rng = np.random.default_rng(13)
    shape = (6, 19, 23)
    ct = rng.integers(-1024, 3071, size=shape).astype(np.int16)
    d = ctx.from_numpy(ct)
    o = bca.median_filter_inplane(ctx, d, shape, flat_axis)
    size = [3, 3, 3]
    size[flat_axis] = 1
    np.testing.assert_array_equal(o.download(shape, np.int16), ndimage.median_filter(ct, size=size))


def test_bca_median_filtering_vs_oracle(ctx):
    """median_filtering=True: tissues from the filtered CT, mean HU from the unfiltered CT (run_pipeline hands the
    original image to the Builder)."""
    from boa_hip import bca
    from oracle import bca as obca
    z = _npz("g8_bca.npz")
    ct = z["ct"].copy()
    rng = np.random.default_rng(14)
    ct = (ct + rng.integers(-60, 60, size=ct.shape)).astype(np.int16)            # noise so the filter matters
    sp = tuple(z["spacing"])
    js, tis = bca.bca_measurements(ctx, ct, z["regions"], z["parts"], sp, None, return_tissues=True,
                                   median_filtering=True, orientation="LPS")
    ref_t = obca.subclassify_tissues(ct, z["regions"], median_filtering=True, slice_axis=0)
    np.testing.assert_array_equal(tis, ref_t)
    assert (ref_t != obca.subclassify_tissues(ct, z["regions"])).any()
    ref = obca.bca_measurements_json(ct, z["regions"], z["parts"], ref_t, sp, None)
    _cmp(json.loads(json.dumps(js, default=float)), json.loads(json.dumps(ref, default=float)), 1e-9)


def test_create_vertebrae_info_vs_oracle(ctx):
    from boa_hip import bca, label_maps
    from oracle import bca as obca
    rng = np.random.default_rng(15)
    cm = label_maps.CLASS_MAP_TOTAL
    vmap = {v[len("vertebrae_"):]: k for k, v in cm.items() if v.startswith("vertebrae_")}
    total = np.zeros((40, 16, 16), np.uint8)
    for i, (vid, lab) in enumerate(sorted(vmap.items())):
        if i % 3 == 2:
            continue                                                           # absent vertebrae
        z0 = rng.integers(0, 36)
        total[z0:z0 + rng.integers(1, 4), 3:9, 4:8] = lab
    for parts in (dict(abdomen=True, thorax=True, neck=True), dict(abdomen=True, thorax=False, neck=False),
                  dict(abdomen=False, thorax=False, neck=False)):
        assert bca.create_vertebrae_info(ctx, total, cm, parts) == obca.create_vertebrae_info(total, vmap, parts)


def test_tissue_projections_match_numpy(ctx):
    """create_tissue_heatmaps' reductions (BCA/report/plots/heatmaps.py:29-101): per-tissue sums over y and x, body
    silhouettes -- bit-exact against the numpy statements of the reference, incl. a slice width that is not a multiple
    of the wave size."""
    from boa_hip import bca
    rng = np.random.default_rng(11)
    Z, Y, X = 9, 37, 83
    tissues = rng.integers(0, 9, size=(Z, Y, X), dtype=np.uint8)
    regions = rng.choice(np.array([0, 1, 2, 3, 11, 255], dtype=np.uint8), size=(Z, Y, X))
    regions[3] = 0                                                      # an empty silhouette slice
    regions[4, :, :40] = 255
    vals = [v for n, v in bca.TISSUES if n in bca.HEATMAP_TISSUES]
    vals = [dict(bca.TISSUES)[n] for n in bca.HEATMAP_TISSUES]            # the reference's tissue order
    d_t, d_r = ctx.from_numpy(tissues), ctx.from_numpy(regions)
    cor, sag, mcor, msag = bca.tissue_projections(ctx, d_t, d_r, (Z, Y, X), vals)
    d_t.free()
    d_r.free()
    body = (regions > 0) & (regions < 255)
    np.testing.assert_array_equal(mcor, body.any(axis=1))
    np.testing.assert_array_equal(msag, body.any(axis=2))
    for t, v in enumerate(vals):
        m = tissues == v
        np.testing.assert_array_equal(cor[t], m.sum(axis=1))
        np.testing.assert_array_equal(sag[t], m.sum(axis=2))


def test_overview_and_l3_axes_on_resident_volumes(ctx):
    """report.create_equidistant_overview / major_minor_axis with the volumes resident on the device: the five slices and the
    middle L3 slice are gathered on the device; results equal golden G14 / the oracle."""
    from boa_hip import report
    from boa_hip.devarray import DevArray
    from oracle import report as orep
    z = np.load(os.path.join(GOLDEN, "g14_overview.npz"))
    for i in range(3):
        d_img = DevArray.from_numpy(ctx, z[f"c{i}_img"])
        d_segs = [(DevArray.from_numpy(ctx, z[f"c{i}_seg{k}"]), z[f"c{i}_cmap{k}"]) for k in range(2)]
        got = report.create_equidistant_overview(d_img, d_segs)
        for s_i, row in enumerate(got):
            for k in range(2):
                np.testing.assert_array_equal(row[1 + k], z[f"c{i}_out"][s_i, k])
        d_img.free()
        for d, _ in d_segs:
            d.free()
    # L3 axes: an elliptic body, vertebra L3 (label 27 here) on slices 9..14
    zz, yy, xx = np.mgrid[:24, :96, :128]
    body = (((xx - 64) / 50.0) ** 2 + ((yy - 48) / 30.0) ** 2 <= 1.0).astype(np.uint8)
    total = np.zeros(body.shape, np.uint8)
    total[9:15, 40:56, 56:72] = 27
    want = orep.major_minor_axis(total == 27, body == 1, (0.8, 0.8))
    d_t, d_b = DevArray.from_numpy(ctx, total), DevArray.from_numpy(ctx, body)
    got = report.major_minor_axis(ctx, d_t, d_b, 27, (0.8, 0.8))
    assert got == want and abs(got[0] - 100 * 0.8) <= 2.0 and abs(got[1] - 60 * 0.8) <= 2.0, (got, want)
    assert report.major_minor_axis(ctx, d_t, d_b, 99, (0.8, 0.8)) == (None, None)
    d_t.free()
    d_b.free()


def test_ccl26_is_deterministic_next_to_a_second_stream():
    """Roots are a function of the mask alone -- also while a second context / stream keeps the GPU busy with the body_parts
    post-processing (its own CCLs, hole filling).  Regression: the union-find's path-halving stores were plain stores; a line
    that stayed dirty in one XCD's L2 could take stale copies of neighbouring parent words with it when it was written back and
    undo another XCD's memory-side atomicMin -- one lost union in ~200 labellings, only when something else shared the GPU
    (tools/ccl_stress.py).  All stores to the parent array in the union kernels are agent-scope atomic stores now."""
    import ctypes as C
    import threading
    from scipy import ndimage
    from boa_hip import bca
    from boa_hip._lib import check
    from boa_hip.device import Context
    shape = (96, 160, 224)
    rng = np.random.default_rng(3)
    sm = ndimage.gaussian_filter(rng.standard_normal(shape), 2.0)
    lab = np.zeros(shape, np.uint8)
    lab[sm > 0.02] = 9
    lab[(sm > 0.02) & (rng.random(shape) < 0.2)] = 5
    lab[sm < -0.25] = 3
    masks = [lab != 0, (lab == 9) | (lab == 5), lab == 5, lab == 3]
    n = int(np.prod(shape))
    c0, c1 = Context(0), Context(0)
    stop = [False]

    def other_stream():
        c1.bind_thread()
        pl = np.zeros(shape, np.uint8)
        pl[sm > 0.0] = 1
        pl[sm < -0.1] = 2
        while not stop[0]:
            dd = c1.from_numpy(pl)
            out = bca.postprocess_part_segmentation_device(c1, dd, shape)
            c1.sync()
            out.free()
            dd.free()

    try:
        d_m = [c0.from_numpy(m.astype(np.uint8)) for m in masks]
        d_r, d_s = c0.alloc(n * 4), c0.alloc(n * 4)

        def roots(k):
            check(c0.lib.boa_ccl26(c0.h, d_m[k].vp, shape[0], shape[1], shape[2], d_r.vp, d_s.vp, None), "boa_ccl26")
            return d_r.download(shape, np.int32), d_s.download((n,), np.uint32)

        ref = [roots(k) for k in range(len(masks))]
        for k, m in enumerate(masks):       # the reference itself against scipy: root = smallest linear index of the component
            lab_k, ncomp = ndimage.label(m, structure=np.ones((3, 3, 3)))
            first = ndimage.minimum(np.arange(n).reshape(shape), lab_k, index=np.arange(1, ncomp + 1)).astype(np.int64)
            np.testing.assert_array_equal(ref[k][0][m], first[lab_k[m] - 1])
        th = threading.Thread(target=other_stream)
        th.start()
        try:
            for it in range(1200):
                k = it % len(masks)
                got_r, got_s = roots(k)
                assert np.array_equal(got_r, ref[k][0]), f"iteration {it}, mask {k}: {(got_r != ref[k][0]).sum()} roots differ"
                assert np.array_equal(got_s, ref[k][1]), f"iteration {it}, mask {k}: component sizes differ"
        finally:
            stop[0] = True
            th.join()
    finally:
        c0.close()
        c1.close()
