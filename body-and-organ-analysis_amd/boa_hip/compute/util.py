"""Two small helpers of the reference's `compute/util.py` whose results the drop-in surface exposes (slice count after
thickness resampling in the run statistics; label-set masks for host-side callers).  The device path does not use
`create_mask`: label selection happens in `boa_label_hu_mask` / `boa_label_select`."""
from __future__ import annotations

from typing import Iterable, Optional, Union

import numpy as np


def convert_resampling_slices(slices: int, current_sampling: float, target_resampling: Optional[float]) -> int:
    """Number of slices after resampling `slices` slices of thickness `current_sampling` to `target_resampling`
    (Python's round-half-even, as in the reference); unchanged when no target is given."""
    if target_resampling is None:
        return slices
    return round((slices / target_resampling) * current_sampling)   # same operation order as the reference (float rounding)


def create_mask(region_data: np.ndarray, labels: Union[int, Iterable[int]]) -> np.ndarray:
    """Boolean mask of the voxels whose label is `labels` (one label) or in `labels` (several), as the reference's
    `np.isin` version (BOA/compute/util.py:25-33)."""
    data = np.asarray(region_data)
    if isinstance(labels, (int, np.integer)):
        return data == labels
    return np.isin(data, list(labels))


def require_int16_exact(values: np.ndarray, what: str = "CT") -> np.ndarray:
    """The device HU statistics (histogram over the int16 range) need integer HU that fit int16.  The reference computes
    its statistics on whatever SimpleITK / get_fdata hand over; a float-valued, scl_slope-scaled or out-of-range CT must
    not be truncated or wrapped silently, so it is refused here."""
    values = np.asarray(values)
    if values.dtype == np.int16:
        return values
    as16 = values.astype(np.int16)
    if not np.array_equal(values, as16):
        raise ValueError(f"{what}: voxel values are not exactly representable as int16 HU (float-valued, scaled or out of "
                         "range); the device HU statistics only support integer HU in [-32768, 32767]")
    return as16
