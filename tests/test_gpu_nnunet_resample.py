"""nnU-Net's own resampling to / from the plans' spacing on the device:
`boa_resize_skimage_f32` (order 3, whole volume or per slice) and `boa_resize_logits_argmax` (order 1 + fp16 rounding +
argmax, fused) against the oracle (oracle/nnunet_resample.py: the reference's resample_data_or_seg with skimage's resize
restated from its published algorithm -- skimage itself is absent: UNPINNED, see the oracle's header), bit for bit; and the
task driver with a model whose plans spacing differs from the CT's against the oracle pipeline."""
import ctypes as C
import os

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


def _resize(ctx, x, new_shape, slice_axis):
    from boa_hip._lib import check
    d_in = ctx.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    d_out = ctx.alloc(int(np.prod(new_shape)) * 4)
    check(ctx.lib.boa_resize_skimage_f32(ctx.h, d_in.vp, _i3(x.shape), d_out.vp, _i3(new_shape), 3, slice_axis))
    out = d_out.download(tuple(new_shape), np.float32)
    d_in.free()
    d_out.free()
    return out


def test_data_resize_matches_golden_g13(ctx):
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "g13_nnunet_resampling.npz"))
    for name in ("iso3d", "sepz_same_z", "sepz_new_z", "new_aniso"):
        meta = z[f"{name}_meta"]
        sep, axis, new_shape = bool(meta[7]), int(meta[8]), [int(v) for v in meta[9:12]]
        got = _resize(ctx, z[f"{name}_in"][0], new_shape, axis if sep else -1)
        np.testing.assert_array_equal(got.view(np.uint32), z[f"{name}_out"][0].view(np.uint32), err_msg=name)


@pytest.mark.parametrize("shape,new_shape,axis", [
    ((37, 52, 41), (45, 40, 41), -1), ((12, 60, 70), (12, 48, 77), 0), ((11, 40, 36), (19, 50, 30), 0),
    ((30, 9, 33), (41, 14, 25), 1), ((28, 31, 7), (20, 44, 7), 2), ((64, 64, 64), (51, 80, 64), -1),
])
def test_data_resize_vs_oracle_random(ctx, shape, new_shape, axis):
    from oracle import nnunet_resample as nnr
    x = (np.random.default_rng(sum(shape)).standard_normal(shape) * 2).astype(np.float32)
    want = nnr.resample_data_or_seg(x[None], new_shape, axis if axis >= 0 else None, 3, axis >= 0, 0)[0]
    got = _resize(ctx, x, new_shape, axis)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("case", [
    dict(grid=(20, 24, 22), off=(0, 0, 0), crop=(20, 24, 22), out=(20, 30, 27), axis=0, C=5),
    dict(grid=(32, 32, 32), off=(3, 0, 5), crop=(26, 32, 22), out=(33, 21, 30), axis=-1, C=12),
    dict(grid=(16, 40, 36), off=(2, 1, 0), crop=(9, 38, 36), out=(14, 30, 45), axis=0, C=3),
    dict(grid=(24, 20, 28), off=(0, 0, 0), crop=(24, 20, 28), out=(24, 20, 28), axis=-1, C=4),
])
def test_logits_resize_argmax_vs_oracle(ctx, case):
    """fp16 logits [C][grid] with the network output in a crop box -> labels on `out`: oracle = the reference's
    resample_data_or_seg (order 1, result dtype float16) + numpy argmax; planted exact ties check first-max-wins."""
    from boa_hip._lib import check
    from oracle import labels as olab
    from oracle import nnunet_resample as nnr
    rng = np.random.default_rng(case["C"])
    lg = (rng.standard_normal((case["C"], *case["grid"])) * 4).astype(np.float16)
    lg[1, ::3] = lg[0, ::3]                                             # ties between class 0 and 1
    o, c = case["off"], case["crop"]
    box = lg[:, o[0]:o[0] + c[0], o[1]:o[1] + c[1], o[2]:o[2] + c[2]]
    ax = case["axis"]
    want_lg = nnr.resample_data_or_seg(np.ascontiguousarray(box), case["out"], ax if ax >= 0 else None, 1, ax >= 0, 0)
    assert want_lg.dtype == np.float16
    want = olab.argmax_labels(want_lg)
    lut = (np.arange(256) * 7 % 251).astype(np.uint8)
    d_lg = ctx.from_numpy(lg.view(np.uint16))
    d_lab = ctx.zeros(int(np.prod(case["out"])))
    check(ctx.lib.boa_resize_logits_argmax(ctx.h, d_lg.vp, case["C"], _i3(case["grid"]), _i3(o), _i3(c), _i3(case["out"]), ax,
                                           lut.ctypes.data_as(C.c_void_p), 0, d_lab.vp))
    got = d_lab.download(tuple(case["out"]), np.uint8)
    np.testing.assert_array_equal(got, lut[want])
    # merge mode: background never overwrites what is there
    check(ctx.lib.boa_memset(ctx.h, d_lab.vp, 9, int(np.prod(case["out"]))))
    check(ctx.lib.boa_resize_logits_argmax(ctx.h, d_lg.vp, case["C"], _i3(case["grid"]), _i3(o), _i3(c), _i3(case["out"]), ax,
                                           lut.ctypes.data_as(C.c_void_p), 1, d_lab.vp))
    got = d_lab.download(tuple(case["out"]), np.uint8)
    np.testing.assert_array_equal(got, np.where(want != 0, lut[want], 9))
    d_lg.free()
    d_lab.free()


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
    return (tid, cfg, blobs), (fns, patch, nc, cfg.intensity_properties["0"], None, list(spacing_zyx))


@pytest.mark.parametrize("precision,bound", [("fp32", 2e-5), ("fp16", 5e-3)])
def test_bca_like_task_with_fixed_plan_spacing(ctx, precision, bound):
    """A body_regions-like model whose plans say (5.0, 1.0, 0.9) run on a 0.8 x 0.8 x 2.0 mm CT: TS resamples only the
    thickness to 5 mm, nnU-Net then resamples in-plane to the plans (separate z, per-slice order 3), predicts with 2 folds,
    resamples the logits back (per-slice order 1) and takes the argmax.  Exact mode: identical labels to the oracle
    pipeline (every step but the network is bit-exact; <= 2e-5 flips allowed for fp32 summation-order near-ties);
    fp16 mode: flip fraction <= 5e-3."""
    from boa_hip.task import SegmentationTask
    from oracle import pipeline as opipe
    rng = np.random.default_rng(8)
    ct = rng.normal(0, 300, size=(44, 40, 60)).astype(np.int16)
    ct[ct == 0] = 1
    ct[:, :, :4] = 0
    sp = (0.8, 0.8, 2.0)
    m, o = _model(542, 6, 542, (5.0, 1.0, 0.9), folds=2)
    want = opipe.predict_image(ct, sp, [o], None, "body_regions", 5.0, resample_only_thickness=True, multimodel=False)
    t = SegmentationTask(ctx, "body_regions", [m], resample=5.0, resample_only_thickness=True, max_batch=4, precision=precision)
    got = t.predict_image(ct, np.diag([sp[0], sp[1], sp[2], 1.0]))
    t.close()
    assert got.shape == ct.shape
    flips = float((got != want).mean())
    print(f"plan-spacing resampled BCA-like task, {precision}: label flip fraction {flips:.3g}")
    assert flips <= bound


def test_native_resolution_task_3d_resample(ctx):
    """A cascade-style model at native resolution (resample=None) whose plans spacing (1.0, 0.75, 0.75) is not the CT's
    (1.5, 0.9, 0.9): one 3-D order-3 resize in, 3-D order-1 logits resize + argmax out (exact mode)."""
    from boa_hip.task import SegmentationTask
    from oracle import pipeline as opipe
    rng = np.random.default_rng(9)
    ct = rng.normal(0, 300, size=(36, 34, 30)).astype(np.int16)
    ct[ct == 0] = 1
    sp = (0.9, 0.9, 1.5)
    m, o = _model(258, 3, 258, (1.0, 0.75, 0.75))
    want = opipe.predict_image(ct, sp, [o], None, "lung_vessels", None, multimodel=False)
    t = SegmentationTask(ctx, "lung_vessels", [m], resample=None, multimodel=False, max_batch=4, precision="fp32")
    got = t.predict_image(ct, np.diag([sp[0], sp[1], sp[2], 1.0]))
    t.close()
    flips = float((got != want).mean())
    print("native-resolution task with plan-spacing resampling: label flip fraction", flips)
    assert flips <= 2e-5
