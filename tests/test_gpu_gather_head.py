"""The gather form of the sliding-window tile loop (csrc/head_gather.hip, boa_net_predict_labels_fold): labels must be
BIT-IDENTICAL to the scatter loop (boa_net_predict_sliding_window + boa_finalize_labels), which tests/test_gpu_head.py pins bit for
bit to the reference's accumulate arithmetic (golden G3 geometries, 512^3) and test_gpu_seams.py to its argmax (G6).  Both walk
every voxel's covering tiles in ascending tile index with the same fp32 add + RTNE rounding per step; the gather form keeps the
running sums in registers instead of fp16 planes in HBM."""
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


def _pred(ctx, patch, nc, folds, step, gaussian=True, features=(32, 64), batch=3, precision=None):
    from boa_hip import plans
    from boa_hip.predictor import HipPredictor
    pj, dj = plans.synthetic_plans(patch=patch, features=features, num_classes=nc)
    geom = plans.model_config_from_plans(pj, dj).geometry
    p = HipPredictor(ctx, geom, tile_step_size=step, max_batch=batch, use_gaussian=gaussian, precision=precision)
    p.set_parameters([plans.weight_blob_from_state_dict(geom, plans.synthetic_state_dict(geom, 40 + f)) for f in range(folds)])
    return p


@pytest.mark.parametrize("patch,shape,nc,folds,step,gaussian", [
    ((32, 32, 32), (70, 50, 96), 5, 1, 0.5, True),        # 3 x 2 x 5 tiles, z a multiple of 32
    ((32, 32, 32), (44, 40, 52), 27, 1, 0.8, True),       # 27 classes, step 0.8, z extent not a multiple of 32, odd origins
    ((32, 32, 64), (33, 47, 150), 7, 3, 0.5, True),       # three folds: fp16 fold sum and mean
    ((32, 32, 32), (20, 40, 45), 4, 2, 0.5, True),        # volume smaller than the patch on axis 0: pad_nd_image + crop
    ((32, 64, 32), (64, 64, 64), 12, 1, 0.5, False),      # use_gaussian=False (weight 1)
])
@pytest.mark.parametrize("precision", ["fp16", "fp32"])
def test_gather_labels_equal_scatter_labels(ctx, patch, shape, nc, folds, step, gaussian, precision):
    """Three results must coincide: (a) the gather form; (b) the ORACLE's tile loop (oracle.sliding_window: accumulate_tile /
    finalize_logits / ensemble_folds / argmax, pinned to the reference by golden G3 / G3b / G6) fed with the device's own per-tile
    fp32 logits (k_head_mfma's logits mode: the same MFMA / bias arithmetic); (c) the scatter form whenever it ran the MFMA head
    (tile origins 8-aligned along z) -- with unaligned origins the scatter loop falls back to an fp32 VALU head whose logits
    differ in the last bits, so (c) is then only required to agree on >= 99.9 % of the voxels.
    precision "fp32" = the split-precision mode: k_gather_head_x3 against k_head_x3 (one arithmetic for every tile origin)."""
    from oracle import labels as olab
    from oracle import sliding_window as osw
    from boa_hip import sliding_window as sw
    p = _pred(ctx, patch, nc, folds, step, gaussian, precision=precision)
    x = np.random.default_rng(sum(shape)).standard_normal((1, *shape)).astype(np.float32)
    lut = (np.arange(nc) * 3 % 251).astype(np.uint8)
    lut[0] = 0
    ctx.counters(reset=True)
    p.use_gather_head = False
    want = p.predict_segmentation(x)
    want_lut = p.predict_segmentation(x, lut=lut)
    cs = ctx.counters(reset=True)
    scatter_heads = cs["head_mfma"] + cs["head_valu"] if precision == "fp16" else cs["x3"]
    if precision == "fp32":
        assert cs["conv_x3"] > 0 and cs["conv_ws"] == 0 and cs["f32"] == 0, cs
    p.use_gather_head = True
    got = p.predict_segmentation(x)
    got_lut = p.predict_segmentation(x, lut=lut)
    cnt = ctx.counters()
    assert scatter_heads > 0 and cnt["head_mfma"] == 0 and cnt["head_valu"] == 0      # the gather path launched no per-tile head
    assert len(np.unique(want)) > 2
    # (b) oracle loop over the device's per-tile logits
    PV, below = sw.pad_amounts(list(shape), list(patch))
    origins = np.asarray(sw.get_sliding_window_origins(PV, list(patch), step), dtype=np.int32)
    xp = np.zeros((1, *PV), np.float32)
    xp[(slice(None),) + tuple(slice(b, b + v) for b, v in zip(below, shape))] = x
    g = osw.compute_gaussian(tuple(patch), 1. / 8, 10) if gaussian else None
    fold_logits = []
    for f in range(folds):
        p._ensure_net(f)
        tiles = p.network_forward(xp, origins)
        acc = np.zeros((nc, *PV), np.float16)
        n = np.zeros(PV, np.float16)
        for t, o in enumerate(origins):
            osw.accumulate_tile(acc, n, tiles[t], g, tuple(int(v) for v in o))
        fold_logits.append(osw.finalize_logits(acc, n))
    logits = osw.ensemble_folds(fold_logits)
    oracle_lab = olab.argmax_labels(logits)[tuple(slice(b, b + v) for b, v in zip(below, shape))]
    p.close()
    np.testing.assert_array_equal(got, oracle_lab)
    np.testing.assert_array_equal(got_lut, lut[oracle_lab])
    if cs["head_valu"] == 0:
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(got_lut, want_lut)
    else:
        agree = float((got == want).mean())
        print(f"scatter loop ran the VALU head ({cs['head_valu']} tiles): agreement with the gather form {agree:.5f}")
        assert agree >= 0.999


def test_gather_merge_into_existing_labels(ctx):
    """merge=True (multi-model tasks, TS/nnunet.py:553-556): background never overwrites, later parts do."""
    p = _pred(ctx, (32, 32, 32), 6, 1, 0.5)
    shape = (48, 40, 64)
    x = np.random.default_rng(3).standard_normal((1, *shape)).astype(np.float32)
    lut = np.array([0, 11, 12, 13, 14, 15], dtype=np.uint8)
    outs = []
    for fused in (False, True):
        p.use_gather_head = fused
        dvol = ctx.from_numpy(x)
        lab = ctx.from_numpy(np.full(shape, 99, dtype=np.uint8))
        p.predict_segmentation_device(dvol, list(shape), lab, lut=lut, merge=True)
        outs.append(lab.download(shape, np.uint8))
        dvol.free()
        lab.free()
    p.close()
    assert (outs[0] == 99).any() and (outs[0] != 99).any()
    np.testing.assert_array_equal(outs[0], outs[1])


def test_gather_inf_flag(ctx):
    """Weights scaled so that the fp16 accumulators overflow: both forms must raise the reference's RuntimeError (:622-625)."""
    from boa_hip import plans
    from boa_hip.predictor import HipPredictor
    pj, dj = plans.synthetic_plans(patch=(32, 32, 32), features=(32, 64), num_classes=3)
    geom = plans.model_config_from_plans(pj, dj).geometry
    sd = plans.synthetic_state_dict(geom, 1)
    key = [k for k in sd if "seg_layers" in k and k.endswith("weight")][-1]
    sd[key] = sd[key] * 2e4          # (finite in fp16; the Gaussian-weighted fp16 sums overflow)
    x = np.random.default_rng(0).standard_normal((1, 40, 36, 64)).astype(np.float32)
    for fused in (False, True):
        p = HipPredictor(ctx, geom, tile_step_size=0.5, max_batch=2)
        p.set_parameters([plans.weight_blob_from_state_dict(geom, sd)])
        p.use_gather_head = fused
        with pytest.raises(RuntimeError, match="inf"):
            p.predict_segmentation(x)
        p.close()


def test_total_pipeline_same_labels_both_forms(ctx):
    """Five part models with crop + CTNormalization + merge through the task driver: the gather form (default) and the scatter
    form give the same label volume -- bit for bit when the scatter loop ran the MFMA head throughout (aligned tile origins), within
    99.9 % when it fell back to the fp32 VALU head for unaligned tiles (step 0.8 on a 32^3 patch: origins 0 / 9 / 20)."""
    from boa_hip import label_maps, plans, totalseg
    rng = np.random.default_rng(11)
    ct = rng.normal(0, 300, size=(44, 40, 52)).astype(np.int16)
    ct[ct == 0] = 1
    ct[:3] = 0
    models = []
    for tid, nc in zip(label_maps.PART_TASK_IDS, (25, 27, 19, 24, 27)):
        pj, dj = plans.synthetic_plans(patch=(32, 32, 32), features=(32, 64), num_classes=nc)
        cfg = plans.model_config_from_plans(pj, dj)
        models.append((tid, cfg, [plans.weight_blob_from_state_dict(cfg.geometry, plans.synthetic_state_dict(cfg.geometry, seed=tid))]))
    outs = []
    for fused in (True, False):
        ts = totalseg.TotalSegmentatorHip(ctx, models, step_size=0.8, max_batch=4)
        for _, _, p, _ in ts.parts:
            p.use_gather_head = fused
        ctx.counters(reset=True)
        outs.append(ts.predict(ct))
        cnt = ctx.counters()
        ts.close()
    if cnt["head_valu"] == 0:
        np.testing.assert_array_equal(outs[0], outs[1])
    else:
        agree = float((outs[0] == outs[1]).mean())
        print("gather vs scatter (VALU head on unaligned tiles) agreement", agree)
        assert agree >= 0.999


def test_task_reads_the_inf_flags_once_per_volume_and_still_raises(ctx):
    """Inside a multi-model task the predictors' inf flags go to a device.FlagRing that the task reads once per volume (one stream
    drain instead of one per model): an overflow in the THIRD of five part models must still raise the reference's RuntimeError
    (predict_from_raw_data.py:622-625) before any label leaves the task, and a clean volume afterwards must run (slots re-zeroed)."""
    from boa_hip import label_maps, plans, totalseg
    rng = np.random.default_rng(5)
    ct = rng.normal(0, 300, size=(44, 40, 52)).astype(np.int16)
    ct[ct == 0] = 1

    def models(bad):
        out = []
        for i, (tid, nc) in enumerate(zip(label_maps.PART_TASK_IDS, (25, 27, 19, 24, 27))):
            pj, dj = plans.synthetic_plans(patch=(32, 32, 32), features=(32, 64), num_classes=nc)
            cfg = plans.model_config_from_plans(pj, dj)
            sd = plans.synthetic_state_dict(cfg.geometry, seed=tid)
            if bad and i == 2:
                key = [k for k in sd if "seg_layers" in k and k.endswith("weight")][-1]
                sd[key] = sd[key] * 2e4
            out.append((tid, cfg, [plans.weight_blob_from_state_dict(cfg.geometry, sd)]))
        return out

    ts = totalseg.TotalSegmentatorHip(ctx, models(True), step_size=0.8, max_batch=4)
    with pytest.raises(RuntimeError, match="inf"):
        ts.predict(ct)
    ts.close()
    ts = totalseg.TotalSegmentatorHip(ctx, models(False), step_size=0.8, max_batch=4)
    lab = ts.predict(ct)
    assert lab.shape == ct.shape and lab.any()
    lab2 = ts.predict(ct)          # second volume through the same task: the ring's slots are reused
    np.testing.assert_array_equal(lab, lab2)
    ts.close()


@pytest.mark.parametrize("precision", ["fp16", "fp32"])
@pytest.mark.parametrize("nc", [2, 4, 5, 13])
def test_gather_ties_and_small_class_counts(ctx, nc, precision):
    """Single-fold fast path of the gather epilogue (argmax from the extremes of the running sums + exact quotients of the
    near-maximum classes): head rows duplicated so that classes TIE exactly -- across the two lanes that hold a voxel's classes and
    within one -- and class counts for which the second lane holds no class at all (nc <= 4).  Labels must equal numpy's argmax
    (first maximum) of the oracle loop over the device's per-tile logits."""
    from boa_hip import plans, sliding_window as sw
    from boa_hip.predictor import HipPredictor
    from oracle import labels as olab
    from oracle import sliding_window as osw
    patch, shape, step = (32, 32, 32), (40, 36, 64), 0.5
    pj, dj = plans.synthetic_plans(patch=patch, features=(32, 64), num_classes=nc)
    geom = plans.model_config_from_plans(pj, dj).geometry
    sd = plans.synthetic_state_dict(geom, 7)
    kw = [k for k in sd if "seg_layers" in k and k.endswith("weight")][-1]
    kb = kw[:-6] + "bias"
    dup = [(1, 0)] if nc == 2 else [(2, 0), (3, 1)] if nc == 4 else [(4, 0), (3, 1)] if nc == 5 else [(5, 2), (9, 2), (12, 7), (1, 0)]
    for dst, src in dup:           # class dst := class src (exact ties wherever src wins)
        sd[kw][dst] = sd[kw][src]
        sd[kb][dst] = sd[kb][src]
    p = HipPredictor(ctx, geom, tile_step_size=step, max_batch=3, precision=precision)
    p.set_parameters([plans.weight_blob_from_state_dict(geom, sd)])
    x = np.random.default_rng(nc).standard_normal((1, *shape)).astype(np.float32)
    got = p.predict_segmentation(x)
    origins = np.asarray(sw.get_sliding_window_origins(list(shape), list(patch), step), dtype=np.int32)
    g = osw.compute_gaussian(tuple(patch), 1. / 8, 10)
    tiles = p.network_forward(x, origins)
    p.close()
    acc = np.zeros((nc, *shape), np.float16)
    n = np.zeros(shape, np.float16)
    for t, o in enumerate(origins):
        osw.accumulate_tile(acc, n, tiles[t], g, tuple(int(v) for v in o))
    want = olab.argmax_labels(osw.finalize_logits(acc, n))
    for dst, src in dup:
        assert not (want == dst).any() or dst < src        # a duplicated higher class never wins a tie
    assert len(np.unique(want)) >= min(nc, 3) - 1
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("precision", ["fp16", "fp32"])
@pytest.mark.parametrize("split_row", [None, 1, 2])
def test_raw_partial_sums_equal_the_scatter_accumulators(ctx, precision, split_row):
    """The gather head's raw mode (tile sharding, the resampled label path): the fp16 accumulator planes and the weight plane it writes
    must be the scatter loop's, bit for bit (tile origins 8-aligned along z: both heads run on the matrix cores).  split_row = r plays
    the tile-sharded protocol on one GPU: the tiles of the rows below r are accumulated first (the lower rank: its own raw launch),
    then the upper block with the planes it shares with the lower block deferred, then boa_net_apply_deferred on top of the lower
    block's sums -- the reference's per-voxel `+=` order (predict_from_raw_data.py:611-614) -- and the planes must again match."""
    from boa_hip import sliding_window as sw
    from boa_hip._lib import check, int3
    patch, shape, nc = (32, 32, 32), (88, 48, 64), 6
    p = _pred(ctx, patch, nc, 1, 0.5, True, precision=precision)
    x = np.random.default_rng(5).standard_normal((1, *shape)).astype(np.float32)
    V, PV, below, origins = p._setup(x)
    origins = np.ascontiguousarray(origins, dtype=np.int32).reshape(-1, 3)
    assert (origins[:, 2] % 8 == 0).all()
    rows = sorted(set(int(v) for v in origins[:, 0]))
    assert len(rows) >= 4
    nvox = int(np.prod(PV))
    dvol = ctx.from_numpy(x)
    bufs = [ctx.alloc(nc * nvox * 2), ctx.alloc(nvox * 2), ctx.alloc(nc * nvox * 2), ctx.alloc(nvox * 2)]
    acc_s, n_s, acc_g, n_g = bufs
    try:
        ctx.counters(reset=True)
        p._run_fold(dvol, V, PV, below, origins, acc_s, n_s, 0)             # scatter loop
        cs = ctx.counters(reset=True)
        assert cs["head_gather"] == 0 and (cs["head_mfma"] > 0 or cs["x3"] > 0) and cs["head_valu"] == 0, cs
        p._ensure_net(0)
        acc_g.zero()
        n_g.zero()
        g = p._gaussian()

        def run(tile_mask, defer):
            org = np.ascontiguousarray(origins[tile_mask], dtype=np.int32)
            d = np.ascontiguousarray(defer, dtype=np.int32)
            st = C.c_void_p()
            check(p.lib.boa_net_predict_sliding_window_deferred(
                p._net, dvol.vp, int3(V), int3(PV), int3(below), org.ctypes.data_as(C.POINTER(C.c_int)), len(org),
                g.vp if g else None, acc_g.vp, n_g.vp, d.ctypes.data_as(C.POINTER(C.c_int)), C.byref(st)), "deferred")
            return st

        if split_row is None:
            p.lib.boa_stash_destroy(run(np.ones(len(origins), bool), np.zeros(len(origins))))
        else:
            lower = origins[:, 0] < rows[split_row]
            hi = rows[split_row - 1] + patch[0]                              # end of the lower block's last row
            p.lib.boa_stash_destroy(run(lower, np.zeros(int(lower.sum()))))
            up = origins[~lower]
            st = run(~lower, np.clip(hi - up[:, 0], 0, patch[0]))
            check(p.lib.boa_net_apply_deferred(p._net, st, g.vp if g else None, acc_g.vp, n_g.vp, int3(PV)), "apply")
            p.lib.boa_stash_destroy(st)
        cg = ctx.counters(reset=True)
        assert cg["head_gather"] == (1 if split_row is None else 3) and cg["head_mfma"] == 0 and cg["head_valu"] == 0, cg
        np.testing.assert_array_equal(acc_g.download((nc, *PV), np.uint16), acc_s.download((nc, *PV), np.uint16))
        np.testing.assert_array_equal(n_g.download(tuple(PV), np.uint16), n_s.download(tuple(PV), np.uint16))
    finally:
        for b in bufs + [dvol]:
            b.free()
        p.close()
