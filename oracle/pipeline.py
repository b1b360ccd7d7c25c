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
