"""BASELINE.json's full sizes (configs[1] 512^3, configs[2] 512x512x768) on the GPU: the integer / fp16 stages against
the oracle evaluated at the same size where it finishes in seconds (tile-loop accumulation order, normalise + argmax +
merge, nearest resampling, voxel aggregation), and size-independent properties where it does not (constant logits survive
the Gaussian-weighted aggregation; a whole `total` volume is reproducible run to run).  Everything here is bit-exact."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from boa_hip.device import Context
    c = Context(0)
    yield c
    c.close()


def _i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


def test_tile_loop_accumulation_order_512(ctx):
    """configs[1] geometry: 512^3, patch 128^3, step 0.8 -> 125 tiles in the reference's x->y->z order.  One head with
    per-tile constant logits: the fp16 accumulator and n_predictions must equal the oracle's sequential fp16 `+=`
    (NN/inference/predict_from_raw_data.py:611-614) bit for bit at every one of the 134 M voxels, and the normalised
    logits (fp16 divide) too."""
    from boa_hip import sliding_window as sw
    from boa_hip._lib import check
    from oracle import sliding_window as osw
    V, P = [512, 512, 512], [128, 128, 128]
    origins = sw.get_sliding_window_origins(V, P, 0.8)
    assert len(origins) == 125
    g16 = np.ascontiguousarray(sw.compute_gaussian(tuple(P), 1. / 8, 10))
    rng = np.random.default_rng(3)
    vals = rng.normal(0, 7, size=len(origins)).astype(np.float32)
    nv = 512 ** 3
    acc, n = ctx.zeros(nv * 2), ctx.zeros(nv * 2)
    d_g = ctx.from_numpy(g16.view(np.uint16))
    d_p = ctx.alloc(128 ** 3 * 4)
    o_acc = np.zeros((1, *V), dtype=np.float16)
    o_n = np.zeros(V, dtype=np.float16)
    for v, o in zip(vals, origins):
        d_p.upload(np.full(128 ** 3, v, dtype=np.float32))
        check(ctx.lib.boa_accumulate_tile(ctx.h, d_p.vp, d_g.vp, acc.vp, n.vp, 1, _i3(P), _i3(V), _i3(o)))
        osw.accumulate_tile(o_acc, o_n, np.full((1, *P), v, dtype=np.float32), g16, tuple(int(x) for x in o))
    ctx.sync()
    np.testing.assert_array_equal(n.download(tuple(V), np.uint16), o_n.view(np.uint16))
    np.testing.assert_array_equal(acc.download((1, *V), np.uint16), o_acc.view(np.uint16))
    flag = ctx.zeros(4)
    check(ctx.lib.boa_finalize_labels(ctx.h, acc.vp, n.vp, 1, _i3(V), None, 0, 0, 1, None, 0, None, None, None, flag.vp))
    want = osw.finalize_logits(o_acc, o_n)
    np.testing.assert_array_equal(acc.download((1, *V), np.uint16), want.view(np.uint16))
    assert int(flag.download((1,), np.int32)[0]) == 0
    for b in (acc, n, d_g, d_p, flag):
        b.free()


def test_constant_logits_survive_aggregation_512(ctx):
    """Size-independent property at 512^3: when every tile predicts the same constant per class, acc / n gives that
    constant back at every voxel up to the fp16 roundings of the sums (<= 125 half-ulp steps are far below the class
    gaps), so the argmax is the largest class everywhere and the merge writes its global label."""
    from boa_hip import sliding_window as sw
    from boa_hip._lib import check
    V, P, Cn = [512, 512, 512], [128, 128, 128], 3
    origins = sw.get_sliding_window_origins(V, P, 0.8)
    nv = 512 ** 3
    acc, n = ctx.zeros(Cn * nv * 2), ctx.zeros(nv * 2)
    d_g = ctx.from_numpy(np.ascontiguousarray(sw.compute_gaussian(tuple(P), 1. / 8, 10)).view(np.uint16))
    consts = np.array([0.25, 3.0, -1.5], dtype=np.float32)
    d_p = ctx.from_numpy(np.repeat(consts, 128 ** 3))
    for o in origins:
        check(ctx.lib.boa_accumulate_tile(ctx.h, d_p.vp, d_g.vp, acc.vp, n.vp, Cn, _i3(P), _i3(V), _i3(o)))
    lab = ctx.alloc(nv)
    check(ctx.lib.boa_memset(ctx.h, lab.vp, 7, nv))
    flag = ctx.zeros(4)
    lut = np.zeros(256, dtype=np.uint8)
    lut[:3] = [0, 42, 99]
    check(ctx.lib.boa_finalize_labels(ctx.h, acc.vp, n.vp, Cn, _i3(V), None, 0, 0, 1, lut.ctypes.data_as(C.c_void_p), 1, lab.vp,
                                      None, None, flag.vp))
    got = lab.download((nv,), np.uint8)
    assert np.all(got == 42)
    # the logits themselves: where the summed weight is well inside the fp16 normal range (towards the volume's edges and
    # corners the Gaussian weights fall to 1e-6 ... 6e-8, where constant x weight rounds to a few subnormal steps -- in the
    # reference's fp16 buffers too; the argmax above is right even there)
    w = n.download((nv,), np.uint16).view(np.float16).astype(np.float32) >= 1e-3
    assert w.mean() > 0.85
    logits = acc.download((Cn, nv), np.uint16).view(np.float16).astype(np.float32)
    for c in range(Cn):
        assert np.abs(logits[c][w] - consts[c]).max() <= 4e-3 * max(1.0, abs(float(consts[c])))
    for b in (acc, n, d_g, d_p, lab, flag):
        b.free()


def test_finalize_argmax_merge_vs_numpy_512(ctx):
    """normalise + argmax + lut + merge at 512^3 against numpy on the same fp16 bit patterns (`torch.div` in fp32 rounded to
    fp16, numpy argmax = first maximum; TS/nnunet.py:553-556 merge: background never overwrites)."""
    from boa_hip._lib import check
    V, Cn = [512, 512, 512], 4
    nv = 512 ** 3
    rng = np.random.default_rng(9)
    acc = rng.normal(0, 4, size=(Cn, nv)).astype(np.float16)
    acc[:, ::7] = acc[0, ::7]                                           # ties -> lowest index
    n = rng.uniform(0.5, 30, size=nv).astype(np.float16)
    prev = rng.integers(0, 200, size=nv, dtype=np.uint8)
    d_acc, d_n, d_lab = ctx.from_numpy(acc.view(np.uint16)), ctx.from_numpy(n.view(np.uint16)), ctx.from_numpy(prev)
    flag = ctx.zeros(4)
    lut = np.zeros(256, dtype=np.uint8)
    lut[:Cn] = [0, 17, 3, 250]
    check(ctx.lib.boa_finalize_labels(ctx.h, d_acc.vp, d_n.vp, Cn, _i3(V), None, 0, 0, 0, lut.ctypes.data_as(C.c_void_p), 1,
                                      d_lab.vp, None, None, flag.vp))
    got = d_lab.download((nv,), np.uint8)
    q = (acc.astype(np.float32) / n.astype(np.float32)[None]).astype(np.float16)
    am = np.argmax(q, axis=0)
    want = np.where(am != 0, lut[am], prev)
    np.testing.assert_array_equal(got, want)
    for b in (d_acc, d_n, d_lab, flag):
        b.free()


def test_nearest_resample_512_vs_oracle(ctx):
    """Label volume 512^3 @1.5 mm -> 3 mm grid and a non-integer zoom back (TS/resampling.py order 0): bit-exact against the
    explicit index formula of scipy.ndimage.zoom(order=0, mode="nearest")."""
    from boa_hip import resample as R
    from oracle import resample as oresample
    rng = np.random.default_rng(2)
    lab = rng.integers(0, 118, size=(64, 64, 64), dtype=np.uint8).repeat(8, 0).repeat(8, 1).repeat(8, 2)
    d_in = ctx.from_numpy(lab)
    for out_shape in ((256, 256, 256), (341, 300, 427)):
        d_out = R.resample_nearest_device(ctx, d_in, lab.shape, out_shape)
        got = d_out.download(out_shape, np.uint8)
        d_out.free()
        want = oresample.spline_zoom_explicit(lab, out_shape, order=0)
        np.testing.assert_array_equal(got, want)
    d_in.free()


def test_voxel_aggregation_config3_vs_numpy(ctx):
    """configs[2] size (768 x 512 x 512 = 201 M voxels): per-label HU histogram (counts, exact HU sums) against
    numpy.bincount, and the slice-wise tissue counts against the whole-volume counts."""
    from boa_hip import bca
    from boa_hip import measurements as M
    from boa_hip import synthetic
    shape = (768, 512, 512)
    ct = np.ascontiguousarray(synthetic.ct_phantom((512, 512, 768), seed=3).transpose(2, 1, 0))
    rng = np.random.default_rng(0)
    lab = rng.integers(0, 118, size=(48, 32, 32), dtype=np.uint8).repeat(16, 0).repeat(16, 1).repeat(16, 2)
    n = ct.size
    d_ct, d_lab = ctx.from_numpy(ct), ctx.from_numpy(lab)
    hist = M.label_hu_histogram(ctx, d_ct, d_lab, n)
    counts = np.bincount(lab.ravel(), minlength=256)
    counts[0] = 0                                                       # background is never measured
    np.testing.assert_array_equal(hist.sum(axis=1), counts)
    hu = np.arange(hist.shape[1], dtype=np.int64) + M.HU_MIN
    sums = np.bincount(lab.ravel(), weights=ct.ravel().astype(np.float64), minlength=256)   # exact: |sum| < 2^53
    np.testing.assert_array_equal((hist.astype(np.int64) * hu[None]).sum(axis=1)[1:], sums[1:].astype(np.int64))
    # tissues: regions from the label pattern (values 1..11), parts all TORSO -> slice tables must add up to the volume
    regions = (lab % 12).astype(np.uint8)
    d_reg = ctx.from_numpy(regions)
    tis, cnt, hsum = bca.tissue_aggregate(ctx, d_ct, d_reg, None, shape)
    t = tis.download(shape, np.uint8)
    whole = np.bincount(t.ravel(), minlength=8)[:8]
    np.testing.assert_array_equal(cnt[:, 0, 1:].sum(axis=0).astype(np.int64), whole[1:])   # (tissue 0 = none: not counted)
    per_tissue_hu = np.bincount(t.ravel(), weights=ct.ravel().astype(np.float64), minlength=8)[:8]
    np.testing.assert_array_equal(hsum[:, 0, 1:].sum(axis=0), per_tissue_hu[1:].astype(np.int64))
    for b in (d_ct, d_lab, d_reg, tis):
        b.free()


def test_total_512_is_reproducible(ctx):
    """Two runs of the whole configs[1] volume (5 synthetic part models, 625 tile forwards) give the same label volume:
    atomics-free statistics and a fixed tile schedule make the result a function of (input, weights, tile batch)."""
    from boa_hip import label_maps, synthetic
    from boa_hip._lib import check
    from boa_hip.predictor import HipPredictor
    shape = [512, 512, 512]
    nvox = 512 ** 3
    ct = synthetic.ct_phantom(shape, seed=20260928)
    d_ct, d_vol, d_lab = ctx.from_numpy(ct), ctx.alloc(nvox * 4), ctx.alloc(nvox)
    models = synthetic.total_part_models()
    preds = []
    for tid, cfg, blob, _ in models:
        p = HipPredictor(ctx, cfg.geometry, tile_step_size=0.8, max_batch=8)
        p.set_parameters([blob])
        preds.append((tid, p))
    ip = models[0][1].intensity_properties["0"]
    work = {}
    runs = []
    for _ in range(2):
        check(ctx.lib.boa_ct_normalize(ctx.h, d_ct.vp, 0, d_vol.vp, nvox, ip["mean"], ip["std"], ip["percentile_00_5"],
                                       ip["percentile_99_5"]))
        d_lab.zero()
        for tid, p in preds:
            p.predict_segmentation_device(d_vol, shape, d_lab, lut=label_maps.part_lut(tid), merge=True, work=work)
        runs.append(d_lab.download((nvox,), np.uint8))
    np.testing.assert_array_equal(runs[0], runs[1])
    assert len(np.unique(runs[0])) > 50
    # the gather form of the tile loop (the default label path: one pass over the volume, fp16 running sums in registers) against
    # the scatter form (fp16 accumulator planes, one head launch per tile in canonical order, finalize pass -- the form
    # tests/test_gpu_head.py pins bit for bit to the oracle's accumulate) at the benchmark's full size: 125 tiles of 128^3
    for tid, p in preds:       # (all five part models: 25 / 27 / 19 / 24 / 27 classes)
        forms = []
        for fused in (True, False):
            p.use_gather_head = fused
            ctx.counters(reset=True)
            d_lab.zero()
            p.predict_segmentation_device(d_vol, shape, d_lab, lut=label_maps.part_lut(tid), merge=False, work=work)
            forms.append(d_lab.download((nvox,), np.uint8))
            assert ctx.counters()["head_valu"] == 0
        p.use_gather_head = True
        np.testing.assert_array_equal(forms[0], forms[1], err_msg=f"part model {tid}")
        assert len(np.unique(forms[0])) > 10
    for _, p in preds:
        p.close()
    for b in list(work.values()) + [d_ct, d_vol, d_lab]:
        b.free()


def test_bca_folds_gather_equals_scatter_full_size(ctx):
    """The multi-fold label path (BCA nets: fold sum / mean through the fp16 fold buffer, general epilogue of k_gather_head) at
    the size the bench runs it -- 154 x 512 x 512 at 5 mm slices, patch 128^3, step 0.5, 98 tiles per fold, 3 folds -- against
    the scatter form (accumulator planes + boa_finalize_labels)."""
    from boa_hip import plans
    from boa_hip.predictor import HipPredictor
    shape = [154, 512, 512]
    nvox = int(np.prod(shape))
    pj, dj = plans.synthetic_plans(num_classes=7, spacing=(5.0, 1.5, 1.5))
    cfg = plans.model_config_from_plans(pj, dj)
    blobs = [plans.weight_blob_from_state_dict(cfg.geometry, plans.synthetic_state_dict(cfg.geometry, 543 + f)) for f in range(3)]
    vol = np.random.default_rng(4).standard_normal([1] + shape).astype(np.float32)
    d_vol, d_lab = ctx.from_numpy(vol), ctx.alloc(nvox)
    p = HipPredictor(ctx, cfg.geometry, tile_step_size=0.5, max_batch=25)
    p.set_parameters(blobs)
    work = {}
    forms = []
    try:
        for fused in (True, False):
            p.use_gather_head = fused
            ctx.counters(reset=True)
            d_lab.zero()
            p.predict_segmentation_device(d_vol, shape, d_lab, work=work)
            forms.append(d_lab.download((nvox,), np.uint8))
            assert ctx.counters()["head_valu"] == 0
        np.testing.assert_array_equal(forms[0], forms[1])
        assert len(np.unique(forms[0])) >= 5
    finally:
        p.close()
        for b in list(work.values()) + [d_vol, d_lab]:
            b.free()


def test_ccl26_full_size_vs_scipy(ctx):
    """boa_ccl26 on 256 x 512 x 512 (8 192 LDS tiles, every XCD takes part in the face unions) against scipy.ndimage.label:
    number of components, root = smallest linear index of the component, sizes[root] = voxel count -- for a blobby mask (one giant
    component + specks, the shape of a body mask) and for its inverse (what the hole filter labels)."""
    import ctypes as C
    from scipy import ndimage
    from boa_hip._lib import check
    shape = (256, 512, 512)
    n = int(np.prod(shape))
    rng = np.random.default_rng(17)
    coarse = ndimage.gaussian_filter(rng.standard_normal((64, 128, 128)).astype(np.float32), 1.5)
    sm = ndimage.zoom(coarse, 4, order=1)
    sm += 0.02 * rng.standard_normal(shape).astype(np.float32)
    d_r, d_s = ctx.alloc(n * 4), ctx.alloc(n * 4)
    idx = np.arange(n, dtype=np.int64).reshape(shape)
    try:
        for m in (sm > 0.01, sm <= 0.01):
            d_m = ctx.from_numpy(m.astype(np.uint8))
            ncomp = C.c_int()
            check(ctx.lib.boa_ccl26(ctx.h, d_m.vp, shape[0], shape[1], shape[2], d_r.vp, d_s.vp, C.byref(ncomp)), "boa_ccl26")
            roots = d_r.download(shape, np.int32)
            sizes = d_s.download((n,), np.uint32)
            d_m.free()
            lab, k = ndimage.label(m, structure=np.ones((3, 3, 3)))
            assert ncomp.value == k and k > 100
            assert (roots[~m] == -1).all()
            first = ndimage.minimum(idx, lab, index=np.arange(1, k + 1)).astype(np.int64)
            np.testing.assert_array_equal(roots[m], first[lab[m] - 1])
            counts = np.bincount(lab.ravel(), minlength=k + 1)[1:]
            np.testing.assert_array_equal(sizes[first], counts.astype(np.uint32))
            assert int(sizes.astype(np.int64).sum()) == int(m.sum())      # nothing counted anywhere else
    finally:
        d_r.free()
        d_s.free()


def test_fill_holes_2d_full_size_vs_scipy(ctx):
    """boa_fill_holes_2d on 96 slices of 512 x 512 (the slice size the LDS bit-flood kernel is built for) against
    scipy.ndimage.binary_fill_holes per slice: blobby masks with enclosed holes, rings, noise, empty and full slices."""
    from scipy import ndimage
    from boa_hip._lib import check
    shape = (96, 512, 512)
    n = int(np.prod(shape))
    rng = np.random.default_rng(23)
    coarse = ndimage.gaussian_filter(rng.standard_normal((96, 128, 128)).astype(np.float32), (0.5, 2.0, 2.0))
    sm = ndimage.zoom(coarse, (1, 4, 4), order=1)
    m = np.abs(sm) > 0.08            # bands around the zero crossings removed: rings and enclosed lakes
    m[3] = rng.random(shape[1:]) < 0.5
    m[4] = False
    m[5] = True
    m[6] = True; m[6, 100:200, 0:50] = False; m[6, 300:320, 300:330] = False
    d_m = ctx.from_numpy(m.astype(np.uint8))
    d_i, d_t, d_o = ctx.alloc(n * 4), ctx.alloc(n), ctx.alloc(n)
    try:
        check(ctx.lib.boa_fill_holes_2d(ctx.h, d_m.vp, shape[0], shape[1], shape[2], d_i.vp, d_t.vp, d_o.vp))
        out = d_o.download(shape, np.uint8).astype(bool)
        ref = np.stack([ndimage.binary_fill_holes(m[i]) for i in range(shape[0])])
        np.testing.assert_array_equal(out, ref)
        assert ref.sum() > m.sum() + 10000
    finally:
        for b in (d_m, d_i, d_t, d_o):
            b.free()


def test_cubic_resampling_large_vs_scipy(ctx):
    """The BCA nets' thickness resampling at volume scale (512 x 512 x 256 int16 @1.5 mm -> 512 x 512 x 77 @5 mm, order 3,
    scipy's `zoom` semantics, TS/resampling.py:129-222): fp64 result bit-identical to scipy, and the int32 truncation the task
    applies."""
    from scipy import ndimage
    from boa_hip import resample
    from boa_hip.synthetic import ct_phantom
    ct = ct_phantom((512, 512, 256), seed=9)
    zoom = (1.0, 1.0, 1.5 / 5.0)
    ref = ndimage.zoom(ct.astype(np.float64), zoom, order=3, mode="nearest")
    out = resample.resample_img(ctx, ct, zoom, 3)
    assert out.shape == ref.shape == (512, 512, 77)
    np.testing.assert_array_equal(out.view(np.uint64), ref.view(np.uint64))
    np.testing.assert_array_equal(out.astype(np.int32), ref.astype(np.int32))


def test_bits_morphology_full_size(ctx, monkeypatch):
    """The bit-mask post-processing (csrc/ccl_bits.hip) at full size.  (a) remove_small_objects on 256 x 512 x 512 (8 192 tiles, a batch of
    the blobby mask and two noise densities; objects and holes) against scipy.ndimage.label; (b) the product functions on the 512^3
    structured phantoms (6 body parts, 11 nested regions) and on a speckled copy: bit path == byte path ($BOA_MORPH_BYTES=1, the boa_ccl26
    chain pinned against scipy above and in tests/test_gpu_aggregation.py)."""
    import ctypes as C
    from scipy import ndimage
    from boa_hip import bca, synthetic
    from boa_hip._lib import check
    from boa_hip.device import BufferView
    shape = (256, 512, 512)
    n = int(np.prod(shape))
    rng = np.random.default_rng(23)
    coarse = ndimage.gaussian_filter(rng.standard_normal((64, 128, 128)).astype(np.float32), 1.5)
    sm = ndimage.zoom(coarse, 4, order=1)
    sm += 0.02 * rng.standard_normal(shape).astype(np.float32)
    masks = [sm > 0.01, rng.random(shape) < 0.06]      # (26-connected site percolation sets in near 0.1: 0.06 keeps many finite clusters)
    seg = np.zeros(shape, np.uint8)
    for j, m in enumerate(masks):
        seg |= m.astype(np.uint8) << j
    lut = np.arange(256, dtype=np.uint8)
    words = int(ctx.lib.boa_bits_words(*shape))
    d_seg = ctx.from_numpy(seg)
    d_bits, d_o = ctx.alloc(words * 4 * len(masks)), ctx.alloc(n)
    S = np.ones((3, 3, 3), bool)
    removed = kept = 0
    try:
        for invert, max_size in ((0, 2999), (1, 2999)):
            check(ctx.lib.boa_bits_select(ctx.h, d_seg.vp, *shape, lut.ctypes.data_as(C.c_void_p), len(masks), d_bits.vp), "boa_bits_select")
            check(ctx.lib.boa_bits_remove_small(ctx.h, d_bits.vp, *shape, len(masks), max_size, invert), "boa_bits_remove_small")
            for j, m in enumerate(masks):
                check(ctx.lib.boa_bits_unpack(ctx.h, BufferView(d_bits, j * words * 4, words * 4).vp, *shape, d_o.vp), "boa_bits_unpack")
                got = d_o.download(shape, np.uint8).astype(bool)
                src = ~m if invert else m
                lab, k = ndimage.label(src, structure=S)
                small = np.bincount(lab.ravel()) <= max_size
                small[0] = False
                want = src & ~small[lab]
                np.testing.assert_array_equal(got, ~want if invert else want, err_msg=f"mask {j} invert {invert}")
                removed += int(small[1:].sum())
                kept += int((~small[1:]).sum())
    finally:
        for b in (d_seg, d_bits, d_o):
            b.free()
    assert removed > 1000 and kept >= 3      # thousands of small components went, the giant ones stayed
    # (b) product functions: bit path == byte path on the structured phantoms and on a speckled copy
    p3 = (512, 512, 512)
    parts = np.ascontiguousarray(synthetic.label_phantom_parts(p3).transpose(2, 1, 0))
    regions = np.ascontiguousarray(synthetic.label_phantom_regions(p3).transpose(2, 1, 0))
    speck = rng.random(p3) < 0.003
    parts_s, regions_s = parts.copy(), regions.copy()
    parts_s[speck] = rng.integers(0, 7, int(speck.sum())).astype(np.uint8)
    regions_s[speck] = rng.integers(0, 12, int(speck.sum())).astype(np.uint8)
    for a, fn in ((parts, bca.postprocess_part_segmentation), (parts_s, bca.postprocess_part_segmentation),
                  (regions, bca.postprocess_region_segmentation), (regions_s, bca.postprocess_region_segmentation)):
        monkeypatch.delenv("BOA_MORPH_BYTES", raising=False)
        got = fn(ctx, a)
        monkeypatch.setenv("BOA_MORPH_BYTES", "1")
        want = fn(ctx, a)
        np.testing.assert_array_equal(got, want)
    monkeypatch.delenv("BOA_MORPH_BYTES", raising=False)
    assert (bca.postprocess_part_segmentation(ctx, parts) == parts).all()           # the clean phantom is a fixed point
    assert (bca.postprocess_region_segmentation(ctx, regions_s) == 255).any()        # the specks are filtered
