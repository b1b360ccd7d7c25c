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
    """Boolean mask of the voxels whose label is `labels` (one label) or in `labels` (several)."""
    data = np.asarray(region_data)
    if isinstance(labels, (int, np.integer)):
        return data == labels
    wanted = np.zeros(int(max(int(data.max(initial=0)), max(labels, default=0))) + 1, dtype=bool)
    wanted[[int(v) for v in labels if int(v) >= 0]] = True
    return wanted[data] if np.issubdtype(data.dtype, np.integer) and data.min(initial=0) >= 0 else np.isin(data, list(labels))
