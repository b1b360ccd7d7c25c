"""Oracle: logits -> labels, part merge, preprocessing scalars (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
from __future__ import annotations

import numpy as np
from scipy.ndimage import binary_fill_holes


def argmax_labels(logits_f16: np.ndarray) -> np.ndarray:
    """NN/utilities/label_handling/label_handling.py:175-178 (`predicted_probabilities.argmax(0)` in numpy on
    the fp16 logits: first maximum wins, NaN counts as maximum) and
    NN/inference/export_prediction.py:43-47 (uint8 when < 255 foreground labels)."""
    return logits_f16.argmax(0).astype(np.uint8)


def merge_parts(part_segs, part_maps, class_map_inv, shape=None):
    """TS/nnunet.py:536-556: seg_combined[seg == jdx] = class_map_inv[class_name] for every part in
    task order; later parts overwrite earlier ones.

    part_segs: list of uint8 arrays (one per part model, same shape)
    part_maps: list of {local_idx: class_name}
    class_map_inv: {class_name: global_idx}
    """
    shape = part_segs[0].shape if shape is None else shape
    out = np.zeros(shape, dtype=np.uint8)
    for seg, pmap in zip(part_segs, part_maps):
        for jdx, name in pmap.items():
            out[seg == jdx] = class_map_inv[name]
    return out


def ct_normalize(image: np.ndarray, mean, std, lower, upper) -> np.ndarray:
    """NN/preprocessing/normalization/default_normalization_schemes.py:53-67 (CTNormalization.run),
    target dtype float32: clip, subtract mean, divide by max(std, 1e-8); all in float32 in place."""
    # plans.json values are python floats (weak scalars): arithmetic stays in float32
    mean, std, lower, upper = float(mean), float(std), float(lower), float(upper)
    img = image.astype(np.float32, copy=True)
    np.clip(img, lower, upper, out=img)
    img -= mean
    img /= max(std, 1e-8)
    return img


def nonzero_bbox(data: np.ndarray):
    """NN/preprocessing/cropping/cropping.py:6-29 (`create_nonzero_mask` + acvl_utils get_bbox_from_mask):
    bbox [[lo, hi), ...] of binary_fill_holes(any channel != 0)."""
    m = data[0] != 0
    for c in range(1, data.shape[0]):
        m |= data[c] != 0
    m = binary_fill_holes(m)
    bbox = []
    for ax in range(m.ndim):
        other = tuple(a for a in range(m.ndim) if a != ax)
        nz = np.where(m.any(axis=other))[0]
        if nz.size == 0:
            bbox.append([0, m.shape[ax]])
        else:
            bbox.append([int(nz.min()), int(nz.max()) + 1])
    return bbox
