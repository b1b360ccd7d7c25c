"""`total` task (TS/python_api.py:totalsegmentator task table :168-189) on the generic task driver (task.py).

Per CT: canonicalise -> resample to 1.5 mm (identity at 1.5 mm, TS/resampling.py:179-181) -> for each part model
291..295: crop_to_nonzero, CTNormalization, sliding window (step 0.8, TS/nnunet.py:507-514), fp16 Gaussian accumulation,
normalise + argmax + `seg_combined[seg == jdx] = class_map_inv[name]` merge on device (TS/nnunet.py:553-556) -> restore.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

from .device import Context
from .plans import ModelConfig
from .task import SegmentationTask, nonzero_bbox  # noqa: F401  (nonzero_bbox re-exported)

TOTAL_TASK = {"task_id": [291, 292, 293, 294, 295], "resample": 1.5, "trainer": "nnUNetTrainerNoMirroring",
              "model": "3d_fullres", "folds": [0], "step_size": 0.8}          # TS/python_api.py:183-189
TOTAL_FAST_TASK = {"task_id": [297], "resample": 3.0, "trainer": "nnUNetTrainer_4000epochs_NoMirroring",
                   "model": "3d_fullres", "folds": [0], "step_size": 0.5}     # TS/python_api.py:169-176


class TotalSegmentatorHip(SegmentationTask):
    """Holds one HipPredictor per part model; `predict(ct_xyz, spacing)` returns the merged `total` label volume."""

    def __init__(self, ctx: Context, models: Sequence[Tuple[int, ModelConfig, Sequence[np.ndarray]]],
                 step_size: float = 0.8, max_batch: int = 16, resample: float = 1.5, precision=None):
        super().__init__(ctx, "total", models, resample=resample, multimodel=True, max_batch=max_batch, precision=precision)
        if abs(step_size - self.step_size) > 1e-12:   # explicit override (the reference derives it, TS/nnunet.py:507-514)
            self.step_size = float(step_size)
            for _, _, p, _ in self.parts:
                p.tile_step_size = float(step_size)

    def predict(self, ct_xyz: np.ndarray, spacing_xyz=(1.5, 1.5, 1.5), affine=None, force_split: bool = False) -> np.ndarray:
        """(x,y,z) CT in RAS+ order with `spacing_xyz` (or any axis order with an explicit `affine`)."""
        if affine is None:
            affine = np.diag([float(spacing_xyz[0]), float(spacing_xyz[1]), float(spacing_xyz[2]), 1.0])
        return self.predict_image(ct_xyz, affine, force_split=force_split)
