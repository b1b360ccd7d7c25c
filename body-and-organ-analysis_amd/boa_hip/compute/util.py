"""BOA/compute/util.py:17-31."""
from __future__ import annotations

import numpy as np


def convert_resampling_slices(slices: int, current_sampling: float, target_resampling):
    if target_resampling is None:
        return slices
    return round((slices / target_resampling) * current_sampling)


def create_mask(region_data: np.ndarray, labels) -> np.ndarray:
    mask = np.zeros(region_data.shape, dtype=bool)
    if isinstance(labels, int):
        mask[region_data == labels] = True
    else:
        mask[np.isin(region_data, labels)] = True
    return mask
