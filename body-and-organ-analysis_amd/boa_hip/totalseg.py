"""`total` task pipeline on the device (host side): the array-level core of
TS/python_api.py:totalsegmentator (task table :168-189) -> TS/nnunet.py:nnUNet_predict_image (:326-829) ->
nnUNetPredictor.predict_from_files, for inputs that are already RAS-canonical and at the model spacing
(1.5 mm: TS/resampling.change_spacing returns its input, :179-181; nnU-Net's resampling is the identity).

Per CT: (x,y,z) -> nnU-Net array order (z,y,x) (NN/imageio/nibabel_reader_writer.py:51-56) -> crop_to_nonzero
bounding box (NN/preprocessing/cropping/cropping.py:19-39) -> CTNormalization on device -> for each part model
291..295: sliding window (step 0.8, TS/nnunet.py:507-514), fp16 Gaussian accumulation, normalise + argmax +
`seg_combined[seg == jdx] = class_map_inv[name]` merge on device (TS/nnunet.py:553-556) -> un-crop -> (x,y,z).
Other spacings / orientations need the resampling kernels and raise NotImplementedError.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import label_maps
from ._lib import check
from .device import Context
from .plans import ModelConfig
from .predictor import HipPredictor

TOTAL_TASK = {"task_id": [291, 292, 293, 294, 295], "resample": 1.5, "trainer": "nnUNetTrainerNoMirroring",
              "model": "3d_fullres", "folds": [0], "step_size": 0.8}          # TS/python_api.py:183-189
TOTAL_FAST_TASK = {"task_id": [297], "resample": 3.0, "trainer": "nnUNetTrainer_4000epochs_NoMirroring",
                   "model": "3d_fullres", "folds": [0], "step_size": 0.5}     # TS/python_api.py:169-176


def nonzero_bbox(data_zyx: np.ndarray) -> List[List[int]]:
    """Bounding box of data != 0 (binary_fill_holes cannot change it), cropping.py:6-29."""
    bbox = []
    for ax in range(3):
        other = tuple(a for a in range(3) if a != ax)
        nz = np.flatnonzero((data_zyx != 0).any(axis=other))
        bbox.append([0, data_zyx.shape[ax]] if nz.size == 0 else [int(nz[0]), int(nz[-1]) + 1])
    return bbox


class TotalSegmentatorHip:
    """Holds one HipPredictor per part model; `predict(ct_xyz)` returns the merged `total` label volume."""

    def __init__(self, ctx: Context, models: Sequence[Tuple[int, ModelConfig, Sequence[np.ndarray]]],
                 step_size: float = 0.8, max_batch: int = 8):
        self.ctx = ctx
        self.parts = []
        for task_id, cfg, blobs in models:
            if cfg.normalization_schemes[0] != "CTNormalization":
                raise ValueError(f"Dataset{task_id}: only CTNormalization is supported on device")
            if cfg.transpose_forward != [0, 1, 2]:
                raise NotImplementedError("plans with a non-identity transpose_forward")
            p = HipPredictor(ctx, cfg.geometry, tile_step_size=step_size, max_batch=max_batch)
            p.set_parameters(list(blobs))
            self.parts.append((task_id, cfg, p))
        self._work: Dict[str, object] = {}

    def close(self):
        for _, _, p in self.parts:
            p.close()
        for b in self._work.values():
            b.free()
        self._work = {}

    def predict_zyx_device(self, d_ct, shape, d_labels, in_dtype: int = 0):
        """Resident int16 (in_dtype 0) / float32 (1) CT [z,y,x] -> resident uint8 labels (zeroed here)."""
        ctx = self.ctx
        n = int(np.prod(shape))
        vol = self._work.get("vol")
        if vol is None or vol.nbytes < n * 4:
            if vol is not None:
                vol.free()
            vol = self._work["vol"] = ctx.alloc(n * 4)
        d_labels.zero()
        for task_id, cfg, p in self.parts:
            ip = cfg.intensity_properties["0"]
            # every part model normalises with its own plans' intensity properties (default_preprocessor.py:336-348)
            check(ctx.lib.boa_ct_normalize(ctx.h, d_ct.vp, in_dtype, vol.vp, n, ip["mean"], ip["std"],
                                           ip["percentile_00_5"], ip["percentile_99_5"]), "boa_ct_normalize")
            p.predict_segmentation_device(vol, list(shape), d_labels, lut=label_maps.part_lut(task_id), merge=True,
                                          work=self._work)

    def predict(self, ct_xyz: np.ndarray, spacing_xyz=(1.5, 1.5, 1.5)) -> np.ndarray:
        if not np.allclose(spacing_xyz, 1.5):
            raise NotImplementedError("input spacing != 1.5 mm needs the resampling kernels (next round)")
        if ct_xyz.ndim != 3:
            raise ValueError("TotalSegmentator does not work for 2D images. Use a 3D image.")
        data = np.ascontiguousarray(ct_xyz.transpose(2, 1, 0))  # nibabel (x,y,z) -> nnU-Net (z,y,x)
        data = data.astype(np.int16) if np.issubdtype(data.dtype, np.integer) else data.astype(np.float32)
        bbox = nonzero_bbox(data)
        sl = tuple(slice(a, b) for a, b in bbox)
        crop = np.ascontiguousarray(data[sl])
        d_ct = self.ctx.from_numpy(crop)
        d_lab = self.ctx.alloc(crop.size)
        try:
            self.predict_zyx_device(d_ct, crop.shape, d_lab, in_dtype=0 if crop.dtype == np.int16 else 1)
            seg_crop = d_lab.download(crop.shape, np.uint8)
        finally:
            d_ct.free()
            d_lab.free()
        seg = np.zeros(data.shape, dtype=np.uint8)  # insert_crop_into_image, export_prediction.py:44-47
        seg[sl] = seg_crop
        return np.ascontiguousarray(seg.transpose(2, 1, 0))
