"""End-to-end task driver on the GPU vs the oracle pipeline: canonicalisation, device resampling
to the model spacing and back, triple z-split, multi-fold BCA nets at (sx, sy, 5 mm).

Label agreement bar: the networks run in fp16 on the device and fp32 in the oracle, so argmax flips are possible at
near-ties of random-weight logits; everything around the network (resampling, cropping, splitting, merging, restore)
is integer-exact -- a structural error would show up as a large disagreement, the bars below only leave room for
near-tie flips (>= 97 % agreement, same bar as the `total` pipeline test)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from boa_hip.device import Context
    c = Context(0)
    yield c
    c.close()


def _model(tid, nc, seed, spacing_zyx, folds=1, patch=(32, 32, 32)):
    import torch
    from boa_hip import plans
    from oracle.network import build_from_arch, network_fn_from_module
    pj, dj = plans.synthetic_plans(patch=patch, features=(32, 64), num_classes=nc, spacing=spacing_zyx)
    cfg = plans.model_config_from_plans(pj, dj)
    blobs, fns = [], []
    for f in range(folds):
        sd = plans.synthetic_state_dict(cfg.geometry, seed=seed + 17 * f)
        blobs.append(plans.weight_blob_from_state_dict(cfg.geometry, sd))
        net = build_from_arch(pj["configurations"]["3d_fullres"]["architecture"]["arch_kwargs"], 1, nc)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        fns.append(network_fn_from_module(net, 8))
    return (tid, cfg, blobs), (fns, patch, nc, cfg.intensity_properties["0"])


def _ct(shape, seed):
    rng = np.random.default_rng(seed)
    ct = rng.normal(0, 300, size=shape).astype(np.int16)
    ct[ct == 0] = 1
    return ct


def test_total_resampled_lps_input(ctx):
    """0.9 x 0.9 x 2.0 mm CT stored in LPS order: canonicalise (flip x, y), cubic resample to 1.5 mm, 5 part models,
    merge, nearest resample back, flip back."""
    from boa_hip import label_maps, totalseg
    from oracle import pipeline as opipe
    ct_ras = _ct((60, 52, 40), 1)
    ct_ras[:4] = 0                                                  # nnU-Net crop_to_nonzero has something to crop
    sp = (0.9, 0.9, 2.0)
    models, omodels = [], []
    for tid, nc in zip(label_maps.PART_TASK_IDS, (25, 27, 19, 24, 27)):
        m, o = _model(tid, nc, tid, (1.5, 1.5, 1.5))
        models.append(m)
        omodels.append(o + (label_maps.CLASS_MAP_PARTS[tid],))
    want_ras = opipe.predict_image(ct_ras, sp, omodels, label_maps.CLASS_MAP_TOTAL_INV, "total", 1.5)
    # the same volume as an LPS file: array flipped along x and y, affine diag(-sx, -sy, sz) with the matching origin
    ct_lps = np.ascontiguousarray(ct_ras[::-1, ::-1, :])
    aff = np.diag([-sp[0], -sp[1], sp[2], 1.0])
    aff[:3, 3] = [sp[0] * (ct_ras.shape[0] - 1), sp[1] * (ct_ras.shape[1] - 1), 0.0]
    ts = totalseg.TotalSegmentatorHip(ctx, models, max_batch=4)
    got = ts.predict(ct_lps, affine=aff)
    ts.close()
    assert got.shape == ct_lps.shape and got.dtype == np.uint8
    agree = float((got[::-1, ::-1, :] == want_ras).mean())
    print("total (resampled, LPS) agreement", agree, "labels", len(np.unique(got)))
    assert agree >= 0.996      # measured 0.9984


def test_force_split_matches_oracle(ctx):
    """Triple z-split bookkeeping (TS/nnunet.py:495-505, :583-586) with a single-model task at its own spacing."""
    from boa_hip.task import SegmentationTask, split_bounds
    from oracle import pipeline as opipe
    parts, comb = split_bounds(1024)                                 # [:361], [322:702], [663:]
    assert parts == [(0, 361), (322, 702), (663, 1024)]
    assert [(d.start, d.stop) for d, _ in comb] == [(0, 341), (341, 682), (682, 1024)]
    ct = _ct((36, 34, 150), 2)
    m, o = _model(900, 5, 900, (1.5, 1.5, 1.5))
    want = opipe.predict_image(ct, (1.5, 1.5, 1.5), [o + (None,)], None, "other", 1.5, multimodel=False, force_split=True)
    t = SegmentationTask(ctx, "other", [m], resample=1.5, max_batch=4)
    got = t.predict_image(ct, np.diag([1.5, 1.5, 1.5, 1.0]), force_split=True)
    nosplit = t.predict_image(ct, np.diag([1.5, 1.5, 1.5, 1.0]))
    t.close()
    agree = float((got == want).mean())
    print("force_split agreement", agree, "split vs unsplit", float((got == nosplit).mean()))
    assert agree >= 0.997      # measured 0.9993


def test_bca_nets_thickness_resampling_5_folds(ctx):
    """body_regions-like task: resample_only_thickness to 5 mm, 5 folds averaged in fp16, step 0.5, single model."""
    from boa_hip.task import SegmentationTask
    from oracle import pipeline as opipe
    ct = _ct((40, 36, 90), 3)
    sp = (0.8, 0.8, 2.0)
    m, o = _model(542, 12, 542, (5.0, 0.8, 0.8), folds=5)
    want = opipe.predict_image(ct, sp, [o + (None,)], None, "body_regions", 5.0, resample_only_thickness=True,
                               multimodel=False)
    t = SegmentationTask(ctx, "body_regions", [m], resample=5.0, resample_only_thickness=True, max_batch=4)
    assert t.step_size == 0.5
    got = t.predict_image(ct, np.diag([sp[0], sp[1], sp[2], 1.0]))
    t.close()
    assert got.shape == ct.shape
    agree = float((got == want).mean())
    print("BCA 5-fold thickness-resampled agreement", agree)
    assert agree >= 0.996      # measured 0.9985


def test_bca_pipeline_vs_oracle_composition(ctx):
    """run_pipeline numerics: nets (fast_bca: 1 fold) -> CC / contour-fill post-processing -> tissues -> LPS reload ->
    body-part detection, vertebra ranges, bca-measurements JSON.  The raw network labels are shared with the oracle
    composition (network parity is covered above), everything after them must be exact (integers) / rtol 1e-9 (means)."""
    import json
    from boa_hip import label_maps
    from boa_hip.pipeline import BcaPipelineHip
    from oracle import bca as obca
    from test_gpu_aggregation import _cmp
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "g8_bca.npz"))
    # G8 phantom is a SimpleITK-ordered LPS volume (z,y,x); store it as an LPS file: array (x,y,z), affine diag(-sx,-sy,sz)
    ct = np.ascontiguousarray(g["ct"].transpose(2, 1, 0))
    sp = [float(v) for v in g["spacing"]]
    aff = np.diag([-sp[0], -sp[1], sp[2], 1.0])
    raw_regions = np.ascontiguousarray(g["regions"].transpose(2, 1, 0)).copy()
    raw_regions[2:5, 2:5, 1:3] = 3                                   # a stray abdominal-cavity island -> 255
    raw_parts = np.ascontiguousarray(g["parts"].transpose(2, 1, 0)).copy()
    total = np.zeros(ct.shape, np.uint8)
    inv = label_maps.CLASS_MAP_TOTAL_INV
    total[10:20, 10:20, 5:12] = inv["vertebrae_L3"]
    total[10:20, 10:20, 30:36] = inv["vertebrae_T9"]
    mp, _ = _model(543, 7, 543, (5.0, sp[1], sp[0]))
    mr, _ = _model(542, 12, 542, (5.0, sp[1], sp[0]))
    pipe = BcaPipelineHip(ctx, (mp[1], mp[2]), (mr[1], mr[2]), fast_bca=True, max_batch=4)
    net_parts = pipe.tasks["body_parts"].predict_image(ct, aff)      # the network path runs (shape/dtype contract) ...
    assert net_parts.shape == ct.shape and net_parts.dtype == np.uint8
    out = pipe.run(ct, aff, total_seg=total, raw_parts=raw_parts, raw_regions=raw_regions, median_filtering=True)
    pipe.close()
    # ... and the pipeline after the networks is compared with the oracle composition
    rg = obca.postprocess_region_segmentation(np.ascontiguousarray(raw_regions.transpose(2, 1, 0)))
    pt = obca.remove_small_labeled_objects(np.ascontiguousarray(raw_parts.transpose(2, 1, 0)))
    np.testing.assert_array_equal(out["body_regions"].transpose(2, 1, 0), rg)
    np.testing.assert_array_equal(out["body_parts"].transpose(2, 1, 0), pt)
    assert (rg == 255).any()
    ct_l = g["ct"]
    tis = obca.subclassify_tissues(ct_l, rg, median_filtering=True, slice_axis=0)
    np.testing.assert_array_equal(out["tissues"].transpose(2, 1, 0), tis)
    flags = obca.examined_body_part(rg, sp)
    assert out["examined_body_part"] == flags
    vmap = {v[len("vertebrae_"):]: k for k, v in label_maps.CLASS_MAP_TOTAL.items() if v.startswith("vertebrae_")}
    vert = obca.create_vertebrae_info(np.ascontiguousarray(total.transpose(2, 1, 0)), vmap, flags)
    assert out["vertebrae"] == vert
    ref = obca.bca_measurements_json(ct_l, rg, pt, tis, sp, vert or None)
    _cmp(json.loads(json.dumps(out["bca_measurements"], default=float)), json.loads(json.dumps(ref, default=float)), 1e-9)


def test_crop_mask_and_permuted_axes(ctx):
    """crop-to-mask (TS/cropping.py:75-110) + a file whose axes are a permutation of RAS with flips: the device views must
    land every label on the voxel the host-side numpy remaps put it."""
    from boa_hip import orientation as o
    from boa_hip.task import SegmentationTask, get_bbox_from_mask
    from oracle import pipeline as opipe
    ct_ras = _ct((44, 40, 48), 7)
    sp = (1.5, 1.5, 1.5)
    m, om = _model(901, 4, 901, (1.5, 1.5, 1.5))
    # file axes = (P, I, R): array[j, k, i] = ras[i, -j, -k]
    aff = np.array([[0, 0, 1.5, -10.0], [-1.5, 0, 0, 50.0], [0, -1.5, 0, 70.0], [0, 0, 0, 1.0]])
    ornt = o.io_orientation(aff)
    ct_file = np.ascontiguousarray(o.apply_orientation(ct_ras, o.ornt_transform(o.RAS_ORNT, ornt)))
    assert o.aff2axcodes(aff) == ("P", "I", "R")
    np.testing.assert_array_equal(o.apply_orientation(ct_file, ornt), ct_ras)
    mask = np.zeros(ct_file.shape, np.uint8)
    mask[6:30, 5:38, 10:36] = 1
    t = SegmentationTask(ctx, "other", [m], resample=1.5, max_batch=4)
    got = t.predict_image(ct_file, aff, crop_mask=mask, crop_addon=(3, 3, 3))
    t.close()
    # oracle composition with host numpy remaps
    bbox = get_bbox_from_mask(mask, 0, (np.array([3, 3, 3]) / o.zooms_from_affine(aff)).astype(int))
    sl = tuple(slice(a, b) for a, b in bbox)
    crop = ct_file[sl].astype(np.int32)
    crop_ras = np.ascontiguousarray(o.apply_orientation(crop, ornt))
    want_ras = opipe.predict_image(crop_ras, sp, [om + (None,)], None, "other", 1.5, multimodel=False)
    want = np.zeros(ct_file.shape, np.uint8)
    want[sl] = o.apply_orientation(want_ras, o.ornt_transform(o.RAS_ORNT, ornt))
    assert got.shape == ct_file.shape
    outside = np.ones(ct_file.shape, bool)
    outside[sl] = False
    assert (got[outside] == 0).all() and bbox[0] == [4, 32]
    agree = float((got == want).mean())
    print("crop + permuted axes agreement", agree)
    assert agree >= 0.998      # measured 0.9997



def test_transpose_forward_plans(ctx):
    """Plans with a non-identity transpose_forward (default_preprocessor.py:57-60: the (z, y, x) array and its spacing are
    permuted before cropping / resampling, export_prediction.py:54-58 permutes the segmentation back): device views vs the
    oracle pipeline, anisotropic patch so that a wrong axis order cannot pass."""
    import torch
    from boa_hip import plans
    from boa_hip.task import SegmentationTask
    from oracle import pipeline as opipe
    from oracle.network import build_from_arch, network_fn_from_module
    tf = [1, 2, 0]
    tb = [tf.index(i) for i in range(3)]
    pj, dj = plans.synthetic_plans(patch=(16, 48, 32), features=(32, 64), num_classes=5, spacing=(1.5, 1.5, 1.5),
                                   kernels=[[1, 3, 3], [3, 3, 3]], strides=[[1, 1, 1], [1, 2, 2]])
    pj["transpose_forward"], pj["transpose_backward"] = tf, tb
    cfg = plans.model_config_from_plans(pj, dj)
    assert list(cfg.transpose_forward) == tf
    sd = plans.synthetic_state_dict(cfg.geometry, seed=77)
    net = build_from_arch(pj["configurations"]["3d_fullres"]["architecture"]["arch_kwargs"], 1, 5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    ct = _ct((52, 36, 44), 21)
    ct[:5] = 0                                   # crop_to_nonzero has something to crop (in the transposed frame)
    sp = (1.5, 1.5, 1.5)
    want = opipe.predict_image(ct, sp, [([network_fn_from_module(net, 8)], (16, 48, 32), 5, cfg.intensity_properties["0"], None)], None,
                               "other", None, multimodel=False, transpose_forward=tf)
    t = SegmentationTask(ctx, "other", [(900, cfg, [plans.weight_blob_from_state_dict(cfg.geometry, sd)])], resample=None, max_batch=4)
    got = t.predict_image(ct, np.diag([1.5, 1.5, 1.5, 1.0]))
    t.close()
    assert got.shape == ct.shape and (got[:5] == 0).all()
    agree = float((got == want).mean())
    print("transpose_forward agreement", agree)
    assert agree >= 0.996
