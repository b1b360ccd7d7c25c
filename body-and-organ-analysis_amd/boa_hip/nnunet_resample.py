"""nnU-Net's own resampling between the image grid and the plans' spacing, host side (the decisions; the arithmetic is
`boa_resize_skimage_f32` / `boa_resize_logits_argmax` in libboa_hip.so):

  compute_new_shape / get_do_separate_z / get_lowres_axis / determine_do_sep_z_and_axis
        NN/preprocessing/resampling/default_resampling.py:13-62 (ANISO_THRESHOLD = 3, NN/configuration.py:7)
  data_plan(...)     NN/preprocessing/preprocessors/default_preprocessor.py:71-93: the image (already cropped to its
                     nonzero box and normalised) goes to the plans' spacing with `resampling_fn_data` (order 3, order_z 0)
  logits_plan(...)   NN/inference/export_prediction.py:25-33: the logits go back to the pre-resampling shape with
                     `resampling_fn_probabilities` (order 1, order_z 0) before the argmax

Only the kwargs every nnU-Net v2 planner writes (and the BOA / TotalSegmentator models ship) are implemented:
resample_data_or_seg_to_shape with is_seg False, order 3 (data) / 1 (probabilities), order_z 0; `force_separate_z` may be
None, True or False.  Anything else raises instead of silently resampling differently."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

ANISO_THRESHOLD = 3
_FN = "resample_data_or_seg_to_shape"
DEFAULT_KW = {"data": {"is_seg": False, "order": 3, "order_z": 0, "force_separate_z": None},
              "probabilities": {"is_seg": False, "order": 1, "order_z": 0, "force_separate_z": None}}


def compute_new_shape(old_shape: Sequence[int], old_spacing: Sequence[float], new_spacing: Sequence[float]) -> List[int]:
    assert len(old_spacing) == len(old_shape) == len(new_spacing)
    return [int(round(i / j * k)) for i, j, k in zip(old_spacing, new_spacing, old_shape)]


def get_do_separate_z(spacing, anisotropy_threshold=ANISO_THRESHOLD) -> bool:
    return bool((np.max(spacing) / np.min(spacing)) > anisotropy_threshold)


def get_lowres_axis(new_spacing) -> np.ndarray:
    return np.where(max(new_spacing) / np.array(new_spacing) == 1)[0]


def determine_do_sep_z_and_axis(force_separate_z, current_spacing, new_spacing,
                                separate_z_anisotropy_threshold=ANISO_THRESHOLD) -> Tuple[bool, Optional[int]]:
    if force_separate_z is not None:
        do_separate_z = bool(force_separate_z)
        axis = get_lowres_axis(current_spacing) if force_separate_z else None
    elif get_do_separate_z(current_spacing, separate_z_anisotropy_threshold):
        do_separate_z, axis = True, get_lowres_axis(current_spacing)
    elif get_do_separate_z(new_spacing, separate_z_anisotropy_threshold):
        do_separate_z, axis = True, get_lowres_axis(new_spacing)
    else:
        do_separate_z, axis = False, None
    if axis is not None:
        if len(axis) in (2, 3):     # e.g. (0.24, 1.25, 1.25): no separate treatment of the out-of-plane axis
            do_separate_z, axis = False, None
        else:
            axis = int(axis[0])
    return do_separate_z, axis


def checked_kwargs(cfg_extra: dict, which: str) -> dict:
    """The plans' `resampling_fn_<which>` + kwargs (`which`: data | probabilities), validated against what the device does."""
    fn = cfg_extra.get(f"resampling_fn_{which}", _FN)
    kw = dict(DEFAULT_KW[which])
    kw.update(cfg_extra.get(f"resampling_fn_{which}_kwargs") or {})
    want = DEFAULT_KW[which]
    if fn != _FN or kw["is_seg"] or kw["order"] != want["order"] or kw["order_z"] != 0:
        raise NotImplementedError(f"resampling_fn_{which} = {fn}({kw}) is not implemented on the device "
                                  f"(supported: {_FN} with is_seg False, order {want['order']}, order_z 0)")
    return kw


def slice_axis_for(kw: dict, current_spacing, new_spacing) -> int:
    """-1: one 3-D resize; else the axis that is resampled separately (per-slice 2-D resize + nearest along it)."""
    sep, axis = determine_do_sep_z_and_axis(kw["force_separate_z"], current_spacing, new_spacing)
    return axis if sep else -1
