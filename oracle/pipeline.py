"""Oracle: the `total` array pipeline on CPU (TEST INFRASTRUCTURE, see oracle/__init__.py).

TS/nnunet.py:nnUNet_predict_image (:326-829) reduced to its array steps for RAS-canonical input at the model
spacing: (x,y,z) -> (z,y,x) -> crop_to_nonzero -> CTNormalization -> per part model sliding window (step 0.8) ->
argmax -> insert crop -> merge parts -> (x,y,z).
"""
from __future__ import annotations

import numpy as np

from . import labels, sliding_window as sw


def predict_total(ct_xyz, part_models, class_map_inv, step_size=0.8):
    """part_models: list of (network_fn, patch_size, num_heads, intensity_props, part_map{idx: name})."""
    data = np.ascontiguousarray(ct_xyz.transpose(2, 1, 0))[None].astype(np.float32)
    bbox = labels.nonzero_bbox(data)
    sl = tuple(slice(a, b) for a, b in bbox)
    crop = data[(slice(None),) + sl]
    segs, maps = [], []
    for fn, patch, heads, ip, pmap in part_models:
        x = labels.ct_normalize(crop[0], ip["mean"], ip["std"], ip["percentile_00_5"], ip["percentile_99_5"])[None]
        lg = sw.predict_sliding_window_return_logits(fn, x, list(patch), heads, step_size)
        seg = np.zeros(data.shape[1:], dtype=np.uint8)
        seg[sl] = labels.argmax_labels(lg)
        segs.append(seg)
        maps.append(pmap)
    comb = labels.merge_parts(segs, maps, class_map_inv)
    return np.ascontiguousarray(comb.transpose(2, 1, 0))


def predict_part(data_xyz, models, class_map_inv, step_size, multimodel, spacing_xyz=None, transpose_forward=None):
    """One s0k_0000 sub-volume through every model (TS/nnunet.py:536-559 or :566-573).  `transpose_forward`: the plans' axis
    permutation applied to the (z, y, x) array before cropping / resampling (default_preprocessor.py:57-60) and undone on the
    segmentation (export_prediction.py:56-58).  A model entry may carry a sixth
    element, its plans' spacing (z, y, x): when it differs from the image's, nnU-Net resamples the normalised crop to the
    plans' grid (order 3) and the fold-mean logits back (order 1) before the argmax (default_preprocessor.py:82-93,
    export_prediction.py:25-33; oracle/nnunet_resample.py)."""
    from . import nnunet_resample as nnr
    data = np.ascontiguousarray(data_xyz.transpose(2, 1, 0))[None].astype(np.float32)
    tf = [0, 1, 2] if transpose_forward is None else [int(v) for v in transpose_forward]
    tb = [tf.index(i) for i in range(3)]
    data = np.ascontiguousarray(data.transpose([0] + [i + 1 for i in tf]))
    bbox = labels.nonzero_bbox(data)
    sl = tuple(slice(a, b) for a, b in bbox)
    crop = data[(slice(None),) + sl]
    segs, maps = [], []
    for entry in models:
        fns, patch, heads, ip, pmap = entry[:5]
        plan_sp = entry[5] if len(entry) > 5 else None
        fns = fns if isinstance(fns, (list, tuple)) else [fns]
        x = labels.ct_normalize(crop[0], ip["mean"], ip["std"], ip["percentile_00_5"], ip["percentile_99_5"])[None]
        shape = x.shape[1:]
        sp_zyx = None if spacing_xyz is None else [float(v) for v in list(spacing_xyz)[::-1]]
        if sp_zyx is not None:
            sp_zyx = [sp_zyx[i] for i in tf]
        new_shape = shape if (plan_sp is None or sp_zyx is None) else tuple(nnr.compute_new_shape(shape, sp_zyx, plan_sp))
        if tuple(new_shape) != tuple(shape):
            x = nnr.resample_to_shape(x, new_shape, sp_zyx, plan_sp, order=3).astype(np.float32)
        lg = sw.ensemble_folds([sw.predict_sliding_window_return_logits(fn, x, list(patch), heads, step_size)
                                for fn in fns])
        if tuple(new_shape) != tuple(shape):
            lg = nnr.resample_to_shape(lg, shape, plan_sp, sp_zyx, order=1)
        seg = np.zeros(data.shape[1:], dtype=np.uint8)
        seg[sl] = labels.argmax_labels(lg)
        segs.append(seg)
        maps.append(pmap)
    comb = labels.merge_parts(segs, maps, class_map_inv) if multimodel else segs[0]
    comb = comb.transpose(tb)
    return np.ascontiguousarray(comb.transpose(2, 1, 0))


def predict_image(ct_xyz, spacing_xyz, models, class_map_inv=None, task_name="total", resample=1.5,
                  resample_only_thickness=False, multimodel=True, force_split=False, transpose_forward=None):
    """TS/nnunet.py:nnUNet_predict_image (:453-699) for an input that is already RAS-canonical: resample (order 3 ->
    int32), optional triple z-split, predict, recombine, resample back (order 0)."""
    from . import resample as orsp
    spacing = np.array(spacing_xyz, dtype=np.float32)
    rsp = None if resample is None else [float(resample)] * 3
    if resample_only_thickness:
        rsp = [spacing[0], spacing[1], rsp[0]]
    if rsp is not None:
        img, zoom = orsp.change_spacing_array(ct_xyz, spacing, rsp, order=3, dtype=np.int32)
    else:
        img, zoom = ct_xyz, None
    step = 0.8 if (task_name == "total" and rsp is not None and rsp[0] < 3.0) else 0.5
    ss = img.shape
    sp_now = [float(v) for v in (rsp if rsp is not None else spacing)]
    if (np.prod(ss) > 512 * 512 * 900 and ss[2] > 200 and multimodel) or force_split:
        third, margin = ss[2] // 3, 20
        p1 = predict_part(img[:, :, :third + margin], models, class_map_inv, step, multimodel, sp_now, transpose_forward)
        p2 = predict_part(img[:, :, third + 1 - margin:third * 2 + margin], models, class_map_inv, step, multimodel, sp_now, transpose_forward)
        p3 = predict_part(img[:, :, third * 2 + 1 - margin:], models, class_map_inv, step, multimodel, sp_now, transpose_forward)
        seg = np.zeros(ss, dtype=np.uint8)
        seg[:, :, :third] = p1[:, :, :-margin]
        seg[:, :, third:third * 2] = p2[:, :, margin - 1:-margin]
        seg[:, :, third * 2:] = p3[:, :, margin - 1:]
    else:
        seg = predict_part(img, models, class_map_inv, step, multimodel, sp_now, transpose_forward)
    if rsp is not None and zoom is not None:
        seg, _ = orsp.change_spacing_array(seg, rsp, rsp, target_shape=ct_xyz.shape, order=0, dtype=np.uint8)
    return seg
