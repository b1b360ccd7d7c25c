"""`bca` model end to end (host side): the array-level equivalent of BCA/commands.py:run_pipeline (:86-181).

    body_parts   = inference(ct, "body_parts")    nnUNet_predict_image(Dataset543, resample 5.0 only in thickness,
                                                   5 folds | 1 with fast_bca, BCA/tasks.py:15-48) + postprocess_part_segmentation
    body_regions = inference(ct, "body_regions")  Dataset542 (+ crop = body_parts if crop_body) + postprocess_region_segmentation
    tissues      = subclassify_tissues(ct, body_regions, median_filtering)
    reload everything in LPS (BCA/io.py:78-94) -> AggregatableBodyPart.from_body_regions -> create_vertebrae_info(total)
    -> Builder.prepare / create_json  -> bca-measurements.json, vertebrae.json

Arrays enter and leave in the CT file's own axis order with its affine (what nibabel's `get_fdata()` / `.affine` give);
SimpleITK views of the same file are the transposed arrays (z,y,x).  Every per-voxel step runs on the device; the
NIfTI files of the reference's folder contract are written by the caller (I/O is SURVEY 8f rank 3).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import bca, label_maps, orientation
from .device import Context
from .plans import ModelConfig
from .task import SegmentationTask

# BCA/tasks.py:15-48
BCA_TASKS = {
    "body_parts": {"task_id": 543, "resample": 5.0, "folds": [0, 1, 2, 3, 4], "resample_only_thickness": True,
                   "trainer": "nnUNetTrainer_1500epochs_NoMirroring"},
    "body_regions": {"task_id": 542, "resample": 5.0, "folds": [0, 1, 2, 3, 4], "resample_only_thickness": True,
                     "trainer": "nnUNetTrainerNoMirroring"},
}


def get_task_info(task_name: str, fast: bool = False) -> dict:
    t = dict(BCA_TASKS[task_name])
    if fast:
        t["folds"] = [0]
    return t


def to_lps_zyx(arr_xyz: np.ndarray, affine: np.ndarray) -> Tuple[np.ndarray, Tuple[float, float, float]]:
    """BCA/io.py:78-94 `process_image` without thickness resampling: reorient to LPS, hand over as a SimpleITK array
    (z,y,x) plus GetSpacing() (x,y,z)."""
    a, aff = orientation.with_axcodes(arr_xyz, affine, "LPS")
    sp = np.sqrt(np.sum(np.asarray(aff, dtype=np.float64)[:3, :3] ** 2, axis=0))
    return np.ascontiguousarray(a.transpose(2, 1, 0)), (float(sp[0]), float(sp[1]), float(sp[2]))


def from_lps_zyx(arr_zyx: np.ndarray, affine: np.ndarray) -> np.ndarray:
    """Inverse of `to_lps_zyx`: back to the file's axis order."""
    cur = orientation.axcodes2ornt("LPS")
    tgt = orientation.io_orientation(affine)
    return np.ascontiguousarray(orientation.apply_orientation(arr_zyx.transpose(2, 1, 0), orientation.ornt_transform(cur, tgt)))


class BcaPipelineHip:
    """parts_model / regions_model: (ModelConfig, [weight blob per fold])."""

    def __init__(self, ctx: Context, parts_model: Tuple[ModelConfig, Sequence[np.ndarray]],
                 regions_model: Tuple[ModelConfig, Sequence[np.ndarray]], fast_bca: bool = False, max_batch: int = 8):
        self.ctx = ctx
        self.tasks: Dict[str, SegmentationTask] = {}
        for name, (cfg, blobs) in (("body_parts", parts_model), ("body_regions", regions_model)):
            info = get_task_info(name, fast_bca)
            blobs = list(blobs)[:len(info["folds"])]
            if len(blobs) != len(info["folds"]):
                raise ValueError(f"{name}: {len(info['folds'])} folds expected, {len(blobs)} weight sets given")
            self.tasks[name] = SegmentationTask(ctx, name, [(info["task_id"], cfg, blobs)], resample=info["resample"],
                                                resample_only_thickness=True, multimodel=False, max_batch=max_batch)

    def close(self):
        for t in self.tasks.values():
            t.close()

    def inference(self, task_name: str, ct: np.ndarray, affine: np.ndarray, force_split: bool = False,
                  crop: Optional[np.ndarray] = None, raw: Optional[np.ndarray] = None) -> np.ndarray:
        """BCA/infer/infer.py:39-89: network labels on the input grid, then the task's post-processing applied to the
        SimpleITK view (z,y,x) of the file.  `raw` short-cuts the network (testing seam)."""
        if raw is None:
            raw = self.tasks[task_name].predict_image(ct, affine, force_split=force_split, crop_mask=crop)
        arr = np.ascontiguousarray(raw.transpose(2, 1, 0)).astype(np.uint8, copy=False)
        if task_name == "body_parts":
            out = bca.postprocess_part_segmentation(self.ctx, arr)
        elif task_name == "body_regions":
            out = bca.postprocess_region_segmentation(self.ctx, arr)
        else:
            raise ValueError(task_name)
        return np.ascontiguousarray(out.transpose(2, 1, 0))

    def run(self, ct: np.ndarray, affine: np.ndarray, total_seg: Optional[np.ndarray] = None,
            median_filtering: bool = False, examined_body_region: Optional[str] = None, crop_body: bool = False,
            force_split: bool = False, raw_parts: Optional[np.ndarray] = None, raw_regions: Optional[np.ndarray] = None) -> dict:
        """-> {"body_parts", "body_regions", "tissues" (file axis order, uint8), "bca_measurements" (dict),
        "vertebrae" (dict)}.  `total_seg`: the `total` label volume on the same grid (vertebra groups), optional."""
        affine = np.asarray(affine, dtype=np.float64)
        parts = self.inference("body_parts", ct, affine, force_split, raw=raw_parts)
        regions = self.inference("body_regions", ct, affine, force_split, crop=parts if crop_body else None,
                                 raw=raw_regions)
        # everything the report needs, in LPS (z,y,x)
        ct_l, spacing = to_lps_zyx(ct, affine)
        rg_l, _ = to_lps_zyx(regions, affine)
        pt_l, _ = to_lps_zyx(parts, affine)
        ct_l = ct_l.astype(np.int16, copy=False)
        present = None
        d_rg = self.ctx.from_numpy(rg_l)
        try:
            present = bca.slice_label_presence(self.ctx, d_rg, rg_l.shape)
        finally:
            d_rg.free()
        if examined_body_region:
            flags = dict(abdomen=False, neck=False, thorax=False)
            key = examined_body_region.lower()
            if key not in flags and key != "none":
                raise KeyError(examined_body_region.upper())
            if key in flags:
                flags[key] = True
        else:
            flags = bca.examined_body_part(present, spacing)
        vertebrae = {}
        if total_seg is not None:
            tot_l, _ = to_lps_zyx(total_seg, affine)
            vertebrae = bca.create_vertebrae_info(self.ctx, tot_l, label_maps.CLASS_MAP_TOTAL, flags)
        js, tis_l = bca.bca_measurements(self.ctx, ct_l, rg_l, pt_l, spacing, vertebrae or None, return_tissues=True,
                                         median_filtering=median_filtering, orientation="LPS",
                                         body_parts_override=flags if examined_body_region else None)
        return {"body_parts": parts, "body_regions": regions, "tissues": from_lps_zyx(tis_l, affine),
                "bca_measurements": js, "vertebrae": vertebrae, "examined_body_part": flags}
