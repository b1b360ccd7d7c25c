"""Oracle: the crop-cascade of `--models all` on host arrays (TEST INFRASTRUCTURE, see oracle/__init__.py).

TS/python_api.py:673-736: a rough `total` segmentation (6 mm Dataset298, 3 mm Dataset297 with robust_crop) -> the union of the
task's crop structures as crop mask -> `nnUNet_predict_image(crop=mask, crop_addon=[20, 20, 20])` (TS/nnunet.py:415-449:
`crop_to_mask` -> bounding box + addon in voxels, TS/cropping.py:7-101) -> the task's own model at its resolution on the cropped
image -> `undo_crop` (TS/cropping.py:131-137, TS/nnunet.py:696-699) -> optionally `remove_outside_of_mask`
(TS/nnunet.py:711-716, TS/postprocessing.py:101-131: scipy `binary_dilation(mask, iterations=addon)`, labels outside cleared).
Inputs are RAS-canonical (x, y, z) arrays (as_closest_canonical is the identity then); the per-image work is
`oracle.pipeline.predict_image`.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
from scipy import ndimage

from . import pipeline


def get_bbox_from_mask(mask: np.ndarray, outside_value: float = -900, addon=0):
    """TS/cropping.py:7-38 (an int addon is used for all three axes; an empty mask gives the whole image)."""
    if isinstance(addon, (int, np.integer)):
        addon = [int(addon)] * 3
    if (mask > outside_value).sum() == 0:
        lo = [0, 0, 0]
        hi = list(mask.shape)
    else:
        idx = np.where(mask > outside_value)
        lo = [int(np.min(idx[a])) - int(addon[a]) for a in range(3)]
        hi = [int(np.max(idx[a])) + 1 + int(addon[a]) for a in range(3)]
    return [[max(0, lo[a]), min(mask.shape[a], hi[a])] for a in range(3)]


def crop_to_mask(img: np.ndarray, mask: np.ndarray, zooms, addon_mm):
    """TS/cropping.py:75-101 on arrays: addon mm -> voxels with the header's (float32) zooms, truncation towards zero."""
    addon = (np.array(addon_mm) / np.asarray(zooms, dtype=np.float32)).astype(int)
    bbox = get_bbox_from_mask(mask, outside_value=0, addon=addon)
    return img[bbox[0][0]:bbox[0][1], bbox[1][0]:bbox[1][1], bbox[2][0]:bbox[2][1]], bbox


def undo_crop(img: np.ndarray, ref_shape, bbox) -> np.ndarray:
    """TS/cropping.py:131-137."""
    out = np.zeros(ref_shape, dtype=img.dtype)
    out[bbox[0][0]:bbox[0][1], bbox[1][0]:bbox[1][1], bbox[2][0]:bbox[2][1]] = img
    return out


def remove_outside_of_mask(seg: np.ndarray, mask: np.ndarray, addon: int = 1) -> np.ndarray:
    """TS/postprocessing.py:101-131 (array form)."""
    seg = seg.copy()
    seg[ndimage.binary_dilation(mask, iterations=addon) == 0] = 0
    return seg


def totalsegmentator_cascade(ct_xyz: np.ndarray, spacing_xyz, rough_models, task_models, class_map_inv: dict,
                             crop_names: Sequence[str], task_name: str, crop_addon=(20, 20, 20), rough_resample: float = 6.0,
                             task_resample: Optional[float] = None, remove_outside: Optional[Sequence[str]] = None,
                             remove_outside_dilation: Optional[float] = None):
    """-> (segmentation on the input grid, rough organ segmentation, crop bbox or None).
    rough_models / task_models: oracle model entries [(network_fn(s), patch, heads, intensity props, part map, plan spacing)]
    (oracle.pipeline.predict_part); class_map_inv: `total` structure name -> label."""
    organ_seg = pipeline.predict_image(ct_xyz, spacing_xyz, rough_models, None, "total", rough_resample, multimodel=False)
    crop_mask = np.zeros(organ_seg.shape, dtype=np.uint8)
    for roi in crop_names:
        crop_mask[organ_seg == class_map_inv[roi]] = 1
    if crop_mask.sum() == 0:       # TS/nnunet.py:428-446: empty crop -> empty segmentation
        return np.zeros(ct_xyz.shape, dtype=np.uint8), organ_seg, None
    zooms = np.asarray(spacing_xyz, dtype=np.float32)
    crop, bbox = crop_to_mask(ct_xyz, crop_mask, zooms, crop_addon)
    crop = crop.astype(np.int32)   # crop_to_mask(..., dtype=np.int32), TS/nnunet.py:447
    seg_c = pipeline.predict_image(crop, spacing_xyz, task_models, None, task_name, task_resample, multimodel=False)
    seg = undo_crop(seg_c, ct_xyz.shape, bbox)
    if remove_outside_dilation is not None:
        rm = np.zeros(organ_seg.shape, dtype=np.uint8)
        for roi in (remove_outside or []):
            rm[organ_seg == class_map_inv[roi]] = 1
        vx = int(remove_outside_dilation / np.mean(zooms))    # TS/nnunet.py:715 (float32 header zooms)
        seg = remove_outside_of_mask(seg, rm, addon=vx)
    return seg, organ_seg, bbox
