"""One segmentation task end to end (host side of SURVEY 8 a10-a13): the array-level equivalent of
TS/nnunet.py:nnUNet_predict_image (:326-829) + nnUNetPredictor.predict_from_files for one CT.

    crop to mask (opt., TS/cropping.py:75-110) -> as_closest_canonical (TS/alignment.py:8-12)
    -> [resample_only_thickness: (sx, sy, resample)] change_spacing order 3 -> int32 (TS/nnunet.py:457-474, device)
    -> triple z-split if > 512*512*900 voxels and z > 200 and multi-model, or force_split (:489-505)
    -> per sub-volume and per model: (x,y,z) -> (z,y,x) float32 (nibabel reader), crop_to_nonzero
       (NN/preprocessing/cropping/cropping.py:19-39), CTNormalization, sliding window (step 0.8 for `total` below 3 mm,
       else 0.5, :507-514), fold mean, argmax, part -> global remap (:553-556)       -- all on the device
    -> recombine the thirds (:580-587) -> change_spacing order 0 back to the input grid (:685-687, device)
    -> undo_canonical (:691) -> undo_crop (:695) -> uint8 labels on the input grid.

Axis permutes / flips / crops are index remaps done with numpy views; every per-voxel computation runs in
libboa_hip.so.  nnU-Net's own resampling to the plans' spacing (default_preprocessor.py:82-96) must be the identity
(true for every BASELINE.json config: `total` is resampled to 1.5 mm = plans spacing, the BCA nets get (sx, sy, 5.0));
a plan with a different spacing raises NotImplementedError instead of silently skipping that step.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import label_maps, orientation
from . import resample as rs
from ._lib import check
from .device import Context
from .plans import ModelConfig
from .predictor import HipPredictor

NR_VOXELS_THR = 512 * 512 * 900  # TS/nnunet.py:484
SPLIT_MARGIN = 20                # TS/nnunet.py:497


def get_bbox_from_mask(mask: np.ndarray, outside_value=0, addon=0) -> List[List[int]]:
    """TS/cropping.py:11-38."""
    addon = [addon] * 3 if isinstance(addon, (int, np.integer)) else [int(a) for a in addon]
    sel = mask > outside_value
    if sel.sum() == 0:
        return [[0, mask.shape[0]], [0, mask.shape[1]], [0, mask.shape[2]]]
    bbox = []
    for ax in range(3):
        idx = np.flatnonzero(sel.any(axis=tuple(a for a in range(3) if a != ax)))
        bbox.append([max(0, int(idx[0]) - addon[ax]), min(mask.shape[ax], int(idx[-1]) + 1 + addon[ax])])
    return bbox


def nonzero_bbox(data_zyx: np.ndarray) -> List[List[int]]:
    """Bounding box of data != 0 (binary_fill_holes cannot change it), cropping.py:6-29."""
    bbox = []
    for ax in range(3):
        other = tuple(a for a in range(3) if a != ax)
        nz = np.flatnonzero((data_zyx != 0).any(axis=other))
        bbox.append([0, data_zyx.shape[ax]] if nz.size == 0 else [int(nz[0]), int(nz[-1]) + 1])
    return bbox


def split_bounds(nz: int) -> Tuple[List[Tuple[int, int]], List[Tuple[slice, slice]]]:
    """Triple split along z (TS/nnunet.py:495-505) and its recombination (:583-586):
    returns [(lo, hi) of each part], [(destination slice, source slice within the part)]."""
    third, m = nz // 3, SPLIT_MARGIN
    parts = [(0, third + m), (third + 1 - m, third * 2 + m), (third * 2 + 1 - m, nz)]
    comb = [(slice(0, third), slice(None, -m)), (slice(third, third * 2), slice(m - 1, -m)),
            (slice(third * 2, nz), slice(m - 1, None))]
    return parts, comb


class SegmentationTask:
    """`models`: [(task_id, ModelConfig, [weight blob per fold])]; several entries = multi-model task (`total`)
    whose part label maps are merged into the global map (later parts overwrite)."""

    def __init__(self, ctx: Context, task_name: str, models: Sequence[Tuple[int, ModelConfig, Sequence[np.ndarray]]],
                 resample: Optional[float] = None, resample_only_thickness: bool = False, multimodel: Optional[bool] = None,
                 max_batch: int = 8, part_luts: Optional[Dict[int, np.ndarray]] = None):
        self.ctx = ctx
        self.task_name = task_name
        self.resample = None if resample is None else float(resample)
        self.resample_only_thickness = bool(resample_only_thickness)
        self.multimodel = (len(models) > 1) if multimodel is None else bool(multimodel)
        # TS/nnunet.py:507-514
        self.step_size = 0.8 if (task_name == "total" and self.resample is not None and self.resample < 3.0) else 0.5
        self.parts = []
        for task_id, cfg, blobs in models:
            if cfg.normalization_schemes[0] != "CTNormalization":
                raise ValueError(f"Dataset{task_id}: only CTNormalization is supported on device")
            if list(cfg.transpose_forward) != [0, 1, 2]:
                raise NotImplementedError("plans with a non-identity transpose_forward")
            p = HipPredictor(ctx, cfg.geometry, tile_step_size=self.step_size, max_batch=max_batch)
            p.set_parameters(list(blobs))
            if self.multimodel:
                lut = part_luts[task_id] if part_luts is not None else label_maps.part_lut(task_id)
            else:
                lut = None
            self.parts.append((task_id, cfg, p, lut))
        self._work: Dict[str, object] = {}

    def close(self):
        for _, _, p, _ in self.parts:
            p.close()
        for b in self._work.values():
            b.free()
        self._work = {}

    # ---- device core: resident CT [z,y,x] -> resident uint8 labels -----------------------------------------
    def predict_zyx_device(self, d_ct, shape, d_labels, in_dtype: int = 0):
        """in_dtype 0 int16 / 1 float32 / 2 int32.  Labels are zeroed here, then every model writes (merges) into them."""
        ctx = self.ctx
        n = int(np.prod(shape))
        vol = self._work.get("vol")
        if vol is None or vol.nbytes < n * 4:
            if vol is not None:
                vol.free()
            vol = self._work["vol"] = ctx.alloc(n * 4)
        d_labels.zero()
        for task_id, cfg, p, lut in self.parts:
            ip = cfg.intensity_properties["0"]
            # every model normalises with its own plans' intensity properties (default_preprocessor.py:336-348)
            check(ctx.lib.boa_ct_normalize(ctx.h, d_ct.vp, in_dtype, vol.vp, n, ip["mean"], ip["std"],
                                           ip["percentile_00_5"], ip["percentile_99_5"]), "boa_ct_normalize")
            p.predict_segmentation_device(vol, list(shape), d_labels, lut=lut, merge=self.multimodel, work=self._work)

    def _check_plan_spacing(self, spacing_xyz):
        sp_zyx = [float(s) for s in spacing_xyz[::-1]]
        for _, cfg, _, _ in self.parts:
            if not np.allclose(sp_zyx, cfg.spacing, rtol=0, atol=1e-3):
                raise NotImplementedError(
                    f"{self.task_name}: image spacing (z,y,x) {sp_zyx} differs from the plans' spacing {list(cfg.spacing)}; "
                    "nnU-Net's internal resampling (default_preprocessor.py:82-96) is not implemented on the device")

    def predict_part_xyz(self, data_xyz: np.ndarray) -> np.ndarray:
        """One (sub-)volume as TS writes it to s0k_0000.nii.gz: (x,y,z) int16/int32/float -> uint8 labels (x,y,z)."""
        data = np.ascontiguousarray(data_xyz.transpose(2, 1, 0))   # nibabel reader: (z,y,x), float32 view of the values
        if data.dtype == np.int16:
            code = 0
        elif data.dtype == np.int32:
            code = 2
        else:
            data, code = data.astype(np.float32), 1
        bbox = nonzero_bbox(data)
        sl = tuple(slice(a, b) for a, b in bbox)
        crop = np.ascontiguousarray(data[sl])
        d_ct = self.ctx.from_numpy(crop)
        d_lab = self.ctx.alloc(max(crop.size, 1))
        try:
            self.predict_zyx_device(d_ct, crop.shape, d_lab, in_dtype=code)
            seg_crop = d_lab.download(crop.shape, np.uint8)
        finally:
            d_ct.free()
            d_lab.free()
        seg = np.zeros(data.shape, dtype=np.uint8)                  # insert_crop_into_image, export_prediction.py:44-47
        seg[sl] = seg_crop
        return seg.transpose(2, 1, 0)

    # ---- nnUNet_predict_image ------------------------------------------------------------------------------
    def predict_image(self, data: np.ndarray, affine: np.ndarray, force_split: bool = False,
                      crop_mask: Optional[np.ndarray] = None, crop_addon=(3, 3, 3), axcodes: str = "RAS") -> np.ndarray:
        """CT array in file axis order + its affine -> uint8 label array on the same grid."""
        if data.ndim == 2:
            raise ValueError("TotalSegmentator does not work for 2D images. Use a 3D image.")
        if data.ndim > 3:
            data = data[:, :, :, 0]
        if data.dtype.fields is not None:
            raise TypeError(f"Invalid dtype {data.dtype}. Expected a simple dtype, not a structured one.")
        orig_shape = data.shape
        affine = np.asarray(affine, dtype=np.float64)
        img, aff = data, affine
        bbox = None
        if crop_mask is not None:
            if crop_mask.sum() == 0:                                 # TS/nnunet.py:428-446
                return np.zeros(orig_shape, dtype=np.uint8)
            addon = (np.array(crop_addon) / orientation.zooms_from_affine(aff)).astype(int)   # mm -> voxels
            bbox = get_bbox_from_mask(crop_mask, outside_value=0, addon=addon)
            img = img[tuple(slice(a, b) for a, b in bbox)]
            aff = aff.copy()
            aff[:3, 3] = np.dot(affine, np.array([bbox[0][0], bbox[1][0], bbox[2][0], 1]))[:3]
            img = img.astype(np.int32)                                # crop_to_mask(dtype=np.int32)
        cropped_affine = aff
        img, aff, _ = orientation.as_closest_canonical(img, aff)
        resample = None if self.resample is None else [self.resample] * 3
        if self.resample_only_thickness:
            img, aff = orientation.with_axcodes(img, aff, axcodes)
            zooms = orientation.zooms_from_affine(aff)
            resample = [zooms[0], zooms[1], resample[0]]
        in_shape = img.shape
        zooms = orientation.zooms_from_affine(aff)
        if resample is not None:
            img_rsp, zoom = rs.change_spacing_array(self.ctx, np.ascontiguousarray(img), zooms, resample, order=3,
                                                    dtype=np.int32)
            sp_rsp = zooms if zoom is None else np.array(resample, dtype=np.float64)
        else:
            img_rsp, zoom, sp_rsp = img, None, zooms
        self._check_plan_spacing(sp_rsp)
        ss = img_rsp.shape
        do_split = (np.prod(ss) > NR_VOXELS_THR and ss[2] > 200 and self.multimodel) or force_split
        if do_split:
            parts, comb = split_bounds(ss[2])
            seg = np.zeros(ss, dtype=np.uint8)
            for (lo, hi), (dst, src) in zip(parts, comb):
                seg[:, :, dst] = self.predict_part_xyz(img_rsp[:, :, lo:hi])[:, :, src]
        else:
            seg = self.predict_part_xyz(img_rsp)
        if resample is not None and zoom is not None:
            seg, _ = rs.change_spacing_array(self.ctx, np.ascontiguousarray(seg), sp_rsp.astype(np.float32), resample,
                                             target_shape=in_shape, order=0, dtype=np.uint8)
        if self.resample_only_thickness:
            # the labels are in `axcodes` order; TS hands them to undo_canonical as they are (RAS is the default, a no-op)
            seg = orientation.apply_orientation(
                seg, orientation.ornt_transform(orientation.axcodes2ornt(axcodes), orientation.RAS_ORNT))
        seg = orientation.undo_canonical(seg, cropped_affine)
        if bbox is not None:                                         # undo_crop, TS/cropping.py:126-132
            full = np.zeros(orig_shape, dtype=np.uint8)
            full[tuple(slice(a, b) for a, b in bbox)] = seg
            seg = full
        if seg.shape != tuple(orig_shape[:3]):
            raise ValueError(f"shape mismatch after restore: {seg.shape} vs {orig_shape}")   # check_if_shape_and_affine_identical
        return np.ascontiguousarray(seg, dtype=np.uint8)
