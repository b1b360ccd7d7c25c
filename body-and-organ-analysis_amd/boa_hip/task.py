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

Axis permutes / flips / crops are index remaps done with device views; every per-voxel computation runs in
libboa_hip.so.  nnU-Net's own resampling to the plans' spacing (default_preprocessor.py:82-93) and of the logits back
(export_prediction.py:25-33) happens per model inside `predict_zyx_device` whenever the array's spacing differs from the
model's plans (`boa_hip/nnunet_resample.py`); it is the identity at every BASELINE.json config (`total` is resampled to
1.5 mm = plans spacing) and active for real-world BCA / cascade inputs whose in-plane spacing differs from the plans'.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import label_maps, orientation
from . import nnunet_resample as nr
from . import resample as rs
from ._lib import check, int3
from .devarray import DevArray
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
                 max_batch: int = 16, part_luts: Optional[Dict[int, np.ndarray]] = None, precision: Optional[str] = None):
        self.ctx = ctx
        self.task_name = task_name
        self.resample = None if resample is None else float(resample)
        self.resample_only_thickness = bool(resample_only_thickness)
        # {key: DeviceBuffer} shared between tasks that resample the SAME resident volume to the same grid (the two BCA
        # nets both take the CT to (sx, sy, 5 mm)): the owner of the dict frees the buffers.  None = resample every time.
        self.resample_cache: Optional[dict] = None
        self.multimodel = (len(models) > 1) if multimodel is None else bool(multimodel)
        # TS/nnunet.py:507-514
        self.step_size = 0.8 if (task_name == "total" and self.resample is not None and self.resample < 3.0) else 0.5
        self.parts = []
        for task_id, cfg, blobs in models:
            if cfg.normalization_schemes[0] != "CTNormalization":
                raise ValueError(f"Dataset{task_id}: only CTNormalization is supported on device")
            tf, tb = [int(v) for v in cfg.transpose_forward], [int(v) for v in cfg.transpose_backward]
            if sorted(tf) != [0, 1, 2] or [tf[i] for i in tb] != [0, 1, 2]:
                raise ValueError(f"Dataset{task_id}: transpose_forward {tf} / transpose_backward {tb} are not inverse permutations")
            if self.parts and tf != [int(v) for v in self.parts[0][1].transpose_forward]:
                raise NotImplementedError("models of one task with different transpose_forward")
            p = HipPredictor(ctx, cfg.geometry, tile_step_size=self.step_size, max_batch=max_batch, precision=precision)
            p.set_parameters(list(blobs))
            if self.multimodel:
                lut = part_luts[task_id] if part_luts is not None else label_maps.part_lut(task_id)
            else:
                lut = None
            self.parts.append((task_id, cfg, p, lut))
        self._work: Dict[str, object] = {}
        # tile_shard.TileShard: several ranks share every volume (tile rows split, overlap slabs exchanged); each rank
        # must be given the same input and ends with the same labels.  None = this process handles whole volumes.
        self.shard = None
        # tile_shard.ShardComm: the part models of a multi-model task are dealt out to the ranks (model k -> rank k mod
        # world, SURVEY 8e granularity 2); every rank must be given the same input and ends with the same labels.  Each
        # model runs exactly as on one GPU (same tiles, same batches), so the result is bit-identical to the one-GPU run.
        self.model_shard = None

    def close(self):
        for _, _, p, _ in self.parts:
            p.close()
        for b in [v for v in self._work.values() if hasattr(v, "free")]:
            b.free()
        self._work = {}

    # ---- device core: resident CT [z,y,x] -> resident uint8 labels -----------------------------------------
    def predict_zyx_device(self, d_ct, shape, d_labels, in_dtype: int = 0, spacing_zyx=None):
        """in_dtype 0 int16 / 1 float32 / 2 int32.  Labels are zeroed here, then every model writes (merges) into them.
        `spacing_zyx`: voxel spacing of the array (nnU-Net axis order); when it differs from a model's plans spacing the
        normalised volume is resampled to the plans' grid and the logits back (nnU-Net's own resampling,
        default_preprocessor.py:82-93 / export_prediction.py:25-33) -- None = the array is at the plans' spacing."""
        ctx = self.ctx
        n = int(np.prod(shape))
        vol = self._work.get("vol")
        if vol is None or vol.nbytes < n * 4:
            if vol is not None:
                vol.free()
            vol = self._work["vol"] = ctx.alloc(n * 4)
        d_labels.zero()
        # The predictors of this task take their inf-flag slots from the ring in `self._work` (predictor.py discovers it there), so EVERY
        # path below owns it: reset before the first model, one read after the last (a path that skipped the check would drop the
        # reference's "Encountered inf in predicted array" error and let unchecked slots pile up -- ADVICE r5).
        ring = self._work.get("flag_ring")
        if ring is None:
            from .device import FlagRing
            ring = self._work["flag_ring"] = FlagRing(ctx)
        ring.reset()
        if self.model_shard is not None and self.model_shard.world > 1 and self.multimodel:
            self._predict_zyx_model_sharded(d_ct, shape, d_labels, in_dtype, vol, n, spacing_zyx)
        elif self.shard is not None and self.shard.comm.world > 1 and self.multimodel and len(self.parts) > 1 and self._units_applicable(shape, spacing_zyx):
            self._predict_zyx_unit_sharded(d_ct, shape, d_labels, in_dtype, n)
        else:
            for k in range(len(self.parts)):
                self._run_model(k, d_ct, shape, d_labels, in_dtype, vol, n, spacing_zyx, merge=self.multimodel, shard=self.shard)
        ring.check()   # the models' inf flags of this rank, one read per volume

    # ---- (row x model) units: all ranks busy on one multi-model volume ---------------------------------------------------------
    def _units_applicable(self, shape, spacing_zyx) -> bool:
        """Every model runs on the array as it is (no plan-spacing resampling: that path all-reduces whole logit volumes)."""
        if spacing_zyx is None:
            return True
        return all(nr.compute_new_shape(shape, spacing_zyx, cfg.spacing) == list(shape) for _, cfg, _, _ in self.parts)

    def _predict_zyx_unit_sharded(self, d_ct, shape, d_labels, in_dtype, n):
        """The part models of a multi-model task are independent until the label merge (TS/nnunet.py:542-556), so the work units of a
        shared volume are (model, tile row) pairs: tile_shard.plan_units cuts the model-major unit list into one contiguous run
        per rank (512^3 `total`: 5 x 5 units -> 3-4 per rank on 8 GPUs, where tile rows alone keep 5 busy).  A model's rows then
        live on a subset of the ranks, which exchange that model's overlap slabs; every rank takes part in the label all-reduce.
        The models are software-pipelined: model k + 1's tiles are queued before model k's exchange is waited for, every model
        with its own normalised copy of the CT (the models have different intensity properties).  Labels are bit-identical to
        the one-GPU run in `exact` mode: per model the accumulation order is the reference's, the merge order is the part order."""
        import dataclasses
        from . import sliding_window as sw
        ctx, world = self.ctx, self.shard.comm.world
        rows, weights = [], []
        for _, cfg, p, _ in self.parts:
            PV, _b = sw.pad_amounts(list(shape), p.geom.patch_size)
            o = sw.get_sliding_window_origins(PV, p.geom.patch_size, p.tile_step_size)
            r = len(set(int(v) for v in o[:, 0]))
            rows.append(r)
            weights.append(len(o) / r)
        from .tile_shard import plan_units
        units = plan_units(rows, world, weights)
        prev = None
        for k in range(len(self.parts)):
            task_id, cfg, p, lut = self.parts[k]
            vol = self._work.get(f"vol{k & 1}")
            if vol is None or vol.nbytes < n * 4:
                if vol is not None:
                    vol.free()
                vol = self._work[f"vol{k & 1}"] = ctx.alloc(n * 4)
            ip = cfg.intensity_properties["0"]
            check(ctx.lib.boa_ct_normalize(ctx.h, d_ct.vp, in_dtype, vol.vp, n, ip["mean"], ip["std"], ip["percentile_00_5"],
                                           ip["percentile_99_5"]), "boa_ct_normalize")
            shard_k = dataclasses.replace(self.shard, assignment=units[k])
            job = p.begin_segmentation_sharded(vol, list(shape), d_labels, lut, True, self._work, shard_k,
                                               after_first_start=(prev.finish if prev is not None else None))
            prev = job
        if prev is not None:
            prev.finish()

    def _run_model(self, k, d_ct, shape, d_out, in_dtype, vol, n, spacing_zyx, merge, shard):
        """One model of the task on the resident array: CTNormalization with the model's own intensity properties
        (default_preprocessor.py:336-348), nnU-Net's resampling to the plans' spacing when it differs from the array's, sliding
        window, labels written (merge=False) or merged (merge=True) into `d_out`."""
        ctx = self.ctx
        task_id, cfg, p, lut = self.parts[k]
        ip = cfg.intensity_properties["0"]
        check(ctx.lib.boa_ct_normalize(ctx.h, d_ct.vp, in_dtype, vol.vp, n, ip["mean"], ip["std"],
                                       ip["percentile_00_5"], ip["percentile_99_5"]), "boa_ct_normalize")
        new_shape = list(shape) if spacing_zyx is None else nr.compute_new_shape(shape, spacing_zyx, cfg.spacing)
        if new_shape == list(shape):      # resample_data_or_seg returns its input unchanged (default_resampling.py:194-196)
            p.predict_segmentation_device(vol, list(shape), d_out, lut=lut, merge=merge, work=self._work, shard=shard)
            return
        # normalise BEFORE resampling (default_preprocessor.py:82-84), order 3 in; order 1 back, then argmax
        ax_in = nr.slice_axis_for(nr.checked_kwargs(cfg.extra, "data"), spacing_zyx, cfg.spacing)
        ax_out = nr.slice_axis_for(nr.checked_kwargs(cfg.extra, "probabilities"), cfg.spacing, spacing_zyx)
        vol_r = ctx.alloc(int(np.prod(new_shape)) * 4)
        try:
            check(ctx.lib.boa_resize_skimage_f32(ctx.h, vol.vp, int3(shape), vol_r.vp, int3(new_shape), 3, ax_in),
                  "boa_resize_skimage_f32")
            p.predict_segmentation_device(vol_r, new_shape, d_out, lut=lut, merge=merge, work=self._work,
                                          shard=shard, resample_to=(list(shape), ax_out))
        finally:
            vol_r.free()

    def _predict_zyx_model_sharded(self, d_ct, shape, d_labels, in_dtype, vol, n, spacing_zyx=None):
        from . import tile_shard as ts
        ctx, comm = self.ctx, self.model_shard
        part = self._work.get("part")
        if part is None or part.nbytes < n:
            if part is not None:
                part.free()
            part = self._work["part"] = ctx.alloc(n)
        for k in range(len(self.parts)):
            check(ctx.lib.boa_memset(ctx.h, part.vp, 0, n), "boa_memset")
            if k % comm.world == comm.rank:   # exactly the one-GPU computation of this model (incl. its plan-spacing resampling)
                self._run_model(k, d_ct, shape, part, in_dtype, vol, n, spacing_zyx, merge=False, shard=None)
            # the owner's label volume reaches every rank (sum over disjoint supports), then the reference's merge in part
            # order: `seg_combined[seg == jdx] = class_map_inv[name]` (TS/nnunet.py:553-556)
            ts.all_reduce_labels(ctx, comm, part, n)
            check(ctx.lib.boa_label_overlay(ctx.h, part.vp, n, d_labels.vp), "boa_label_overlay")

    # ---- one (sub-)volume, device resident ----------------------------------------------------------------------
    def _predict_part_device(self, part_xyz: DevArray, dst_xyz: DevArray, src_lo: int, src_hi: int, spacing_zyx=None):
        """`part_xyz`: (x,y,z) view of the (resampled) CT that TS would write to s0k_0000.nii.gz.  Its labels for the
        part-local z range [src_lo, src_hi) are written into `dst_xyz` (a zero-initialised (x,y,z) view of the same
        x/y extent and src_hi - src_lo slices).  nibabel reader view change, crop_to_nonzero and insert_crop_into_image
        (export_prediction.py:44-47) are views + one device copy each way."""
        dt = part_xyz.dtype
        want = dt if dt in (np.dtype(np.int16), np.dtype(np.int32)) else np.dtype(np.float32)
        code = {np.dtype(np.int16): 0, np.dtype(np.float32): 1, np.dtype(np.int32): 2}[want]
        zyx = part_xyz.transpose((2, 1, 0)).contiguous(want, force_copy=False)
        # plans' transpose_forward (default_preprocessor.py:57-60: data.transpose([0, *[i + 1 for i in transpose_forward]]) before
        # cropping / resampling; export_prediction.py:56-58 transposes the segmentation back): a device view + one copy
        tf = [int(v) for v in self.parts[0][1].transpose_forward]
        ident = tf == [0, 1, 2]
        work = zyx if ident else zyx.transpose(tuple(tf)).contiguous(want, force_copy=True)
        sp_t = None if spacing_zyx is None else [spacing_zyx[i] for i in tf]
        bbox_t = work.nonzero_bbox()
        full = all(b == [0, n] for b, n in zip(bbox_t, work.shape))
        crop = work if full else work.box(bbox_t).contiguous()
        d_lab = DevArray.empty(self.ctx, crop.shape, np.uint8)
        try:
            self.predict_zyx_device(crop.buf, crop.shape, d_lab.buf, in_dtype=code, spacing_zyx=sp_t)
            # back to (z, y, x): axis i of the transposed array is axis tf[i] of the original
            bbox = [None, None, None]
            for i in range(3):
                bbox[tf[i]] = bbox_t[i]
            lab_zyx = d_lab if ident else d_lab.transpose(tuple(int(v) for v in self.parts[0][1].transpose_backward))
            k0, k1 = max(bbox[0][0], src_lo), min(bbox[0][1], src_hi)
            if k0 < k1:
                src = lab_zyx.slice(0, k0 - bbox[0][0], k1 - bbox[0][0]).transpose((2, 1, 0))
                dst = dst_xyz.slice(2, k0 - src_lo, k1 - src_lo).slice(1, bbox[1][0], bbox[1][1]).slice(0, bbox[2][0], bbox[2][1])
                src.copy_to(dst)
        finally:
            d_lab.free()
            if crop is not work:
                crop.free()
            if work is not zyx:
                work.free()
            if zyx.buf is not part_xyz.buf:
                zyx.free()

    def predict_part_xyz(self, data_xyz: np.ndarray) -> np.ndarray:
        """One (sub-)volume as TS writes it to s0k_0000.nii.gz: (x,y,z) array -> uint8 labels (x,y,z)."""
        src = DevArray.from_numpy(self.ctx, self._supported(data_xyz))
        seg = DevArray.zeros(self.ctx, src.shape, np.uint8)
        try:
            self._predict_part_device(src, seg, 0, src.shape[2])
            return seg.download()
        finally:
            src.free()
            seg.free()

    @staticmethod
    def _supported(data: np.ndarray) -> np.ndarray:
        data = np.ascontiguousarray(data)
        if data.dtype in (np.uint8, np.int16, np.int32, np.float32, np.float64):
            return data
        return data.astype(np.float64)   # what get_fdata() hands to the reference

    # ---- nnUNet_predict_image ------------------------------------------------------------------------------
    def predict_image(self, data, affine: np.ndarray, force_split: bool = False,
                      crop_mask: Optional[np.ndarray] = None, crop_addon=(3, 3, 3), axcodes: str = "RAS",
                      return_device: bool = False):
        """CT array in file axis order + its affine -> uint8 label array on the same grid.  The volume is uploaded
        once; reorientation, crops, splits and restores are device views / copies, resampling and inference run on the
        device, only the final label volume comes back.  `data` may be a resident DevArray (not freed here);
        `return_device=True` returns the labels as a contiguous DevArray (caller frees) instead of downloading."""
        resident = isinstance(data, DevArray)
        if not resident:
            if data.ndim == 2:
                raise ValueError("TotalSegmentator does not work for 2D images. Use a 3D image.")
            if data.ndim > 3:
                data = data[:, :, :, 0]
            if data.dtype.fields is not None:
                raise TypeError(f"Invalid dtype {data.dtype}. Expected a simple dtype, not a structured one.")
        ctx = self.ctx
        orig_shape = tuple(int(v) for v in data.shape)
        affine = np.asarray(affine, dtype=np.float64)
        aff = affine
        bbox = None
        if crop_mask is not None and crop_mask.sum() == 0:              # TS/nnunet.py:428-446
            return DevArray.zeros(ctx, orig_shape, np.uint8) if return_device else np.zeros(orig_shape, dtype=np.uint8)
        owned = []                                                     # device arrays to release

        def own(a):
            owned.append(a)
            return a
        try:
            view = data if resident else own(DevArray.from_numpy(ctx, self._supported(data)))
            cast = None
            if crop_mask is not None:
                addon = (np.array(crop_addon) / orientation.zooms_from_affine(aff)).astype(int)   # mm -> voxels
                bbox = get_bbox_from_mask(crop_mask, outside_value=0, addon=addon)
                view = view.box(bbox)
                aff = aff.copy()
                aff[:3, 3] = np.dot(affine, np.array([bbox[0][0], bbox[1][0], bbox[2][0], 1]))[:3]
                cast = np.int32                                          # crop_to_mask(dtype=np.int32)
            cropped_affine = aff
            ornt = orientation.io_orientation(aff)                       # as_closest_canonical
            if not np.array_equal(ornt, orientation.RAS_ORNT):
                aff = aff.dot(orientation.inv_ornt_aff(ornt, view.shape))
                view = view.apply_orientation(ornt)
            resample = None if self.resample is None else [self.resample] * 3
            thick_ornt = None
            if self.resample_only_thickness:
                cur = "".join(orientation.aff2axcodes(aff))
                if cur != "".join(axcodes):
                    thick_ornt = orientation.ornt_transform(orientation.axcodes2ornt(cur), orientation.axcodes2ornt(axcodes))
                    aff = aff.dot(orientation.inv_ornt_aff(thick_ornt, view.shape))
                    view = view.apply_orientation(thick_ornt)
                zooms = orientation.zooms_from_affine(aff)
                resample = [zooms[0], zooms[1], resample[0]]
            in_shape = view.shape
            zooms = orientation.zooms_from_affine(aff)
            zoom = None
            if resample is not None:                                     # change_spacing (TS/resampling.py:165-181)
                new_spacing = np.array(resample)
                zoom = zooms / new_spacing
                if np.array_equal(zooms, new_spacing):
                    zoom = None
            if zoom is not None:
                out_shape = rs.zoomed_shape(in_shape, zoom)
                key = None
                if self.resample_cache is not None and resident:
                    key = (data.buf.ptr, view.offset, view.shape, view.strides, str(view.dtype), str(cast), tuple(out_shape),
                           tuple(float(z) for z in zoom))   # (the cache lives for ONE resident volume: pipeline.run_resident)
                if key is not None and key in self.resample_cache:
                    img_rsp = DevArray(ctx, self.resample_cache[key], out_shape, np.int32)      # (not owned: the cache's)
                else:
                    src = view.contiguous(cast)
                    if src.buf is not view.buf:
                        own(src)
                    if src.dtype == np.uint8:
                        src = own(src.contiguous(np.int16, force_copy=True))
                    buf = rs.resample_cubic_device(ctx, src.buf, src.dtype, src.shape, out_shape, np.int32)
                    img_rsp = DevArray(ctx, buf, out_shape, np.int32)
                    if key is not None:
                        self.resample_cache[key] = buf
                    else:
                        own(img_rsp)
                sp_rsp = np.array(resample, dtype=np.float64)
            else:
                img_rsp = view if cast is None else own(view.contiguous(cast))
                sp_rsp = zooms
            ss = img_rsp.shape
            # nnU-Net reads the spacing from the header of the file TS wrote (float32 pixdim), in array axis order (z, y, x)
            sp_zyx = [float(np.float32(v)) for v in sp_rsp[::-1]]
            seg = own(DevArray.zeros(ctx, ss, np.uint8))
            do_split = (np.prod(ss) > NR_VOXELS_THR and ss[2] > 200 and self.multimodel) or force_split
            if do_split:
                parts, comb = split_bounds(ss[2])
                for (lo, hi), (dst, srcsl) in zip(parts, comb):
                    a, b, _ = srcsl.indices(hi - lo)
                    self._predict_part_device(img_rsp.slice(2, lo, hi), seg.slice(2, dst.start, dst.stop), a, b, sp_zyx)
            else:
                self._predict_part_device(img_rsp, seg, 0, ss[2], sp_zyx)
            if zoom is not None:                                         # back to the input grid (TS/nnunet.py:685-687)
                if tuple(in_shape) != tuple(ss):
                    buf = rs.resample_nearest_device(ctx, seg.buf, ss, in_shape)
                    seg = own(DevArray(ctx, buf, in_shape, np.uint8))
            out = seg
            if thick_ornt is not None:
                # the labels are in `axcodes` order; back to RAS before undo_canonical
                out = out.apply_orientation(orientation.ornt_transform(orientation.axcodes2ornt(axcodes), orientation.RAS_ORNT))
            out = out.apply_orientation(orientation.ornt_transform(orientation.RAS_ORNT, orientation.io_orientation(cropped_affine)))
            if bbox is not None:                                         # undo_crop, TS/cropping.py:126-132
                full = own(DevArray.zeros(ctx, orig_shape, np.uint8))
                out.copy_to(full.box(bbox))
                out = full
            if tuple(out.shape) != orig_shape[:3]:
                raise ValueError(f"shape mismatch after restore: {out.shape} vs {orig_shape}")   # check_if_shape_and_affine_identical
            if return_device:
                res = out.contiguous(force_copy=(resident and out.buf is data.buf))
                keep = id(res.buf)
                owned[:] = [a for a in owned if id(a.buf) != keep]
                return res
            return out.download()
        finally:
            seen = set()
            for a in owned:
                if id(a.buf) not in seen:
                    seen.add(id(a.buf))
                    a.free()


def remove_outside_of_mask(ctx: Context, seg: np.ndarray, mask: np.ndarray, addon: int = 1) -> np.ndarray:
    """TS/postprocessing.py:101-131 on arrays of the same grid: dilate `mask != 0` `addon` times with the 6-neighbour cross
    (scipy.ndimage.binary_dilation(mask, iterations=addon)), clear `seg` outside.  Runs on the device."""
    if seg.shape != mask.shape:
        raise ValueError("segmentation and mask must have the same shape")
    if addon < 1:
        raise NotImplementedError("remove_outside_of_mask with addon < 1 (scipy dilates until convergence)")
    n = int(seg.size)
    Z, Y, X = (int(v) for v in seg.shape)
    d_seg = ctx.from_numpy(np.ascontiguousarray(seg, dtype=np.uint8))
    d_m = ctx.from_numpy(np.ascontiguousarray(mask != 0, dtype=np.uint8))
    d_o, d_t = ctx.alloc(n), ctx.alloc(n)
    try:
        check(ctx.lib.boa_binary_dilate_cross(ctx.h, d_m.vp, d_o.vp, d_t.vp, Z, Y, X, int(addon)), "boa_binary_dilate_cross")
        check(ctx.lib.boa_mask_assign(ctx.h, d_o.vp, n, 1, 0, d_seg.vp), "boa_mask_assign")          # seg[mask == 0] = 0
        return d_seg.download(seg.shape, np.uint8)
    finally:
        for b in (d_seg, d_m, d_o, d_t):
            b.free()


def run_cascade_task(ctx: Context, task: str, data: np.ndarray, affine: np.ndarray, rough_models, task_models,
                     crop_names: Sequence[str], crop_addon=(3, 3, 3), max_batch: int = 16, rough_resample: float = 6.0,
                     remove_outside: Optional[Sequence[str]] = None, remove_outside_dilation: Optional[float] = None,
                     precision: Optional[str] = None) -> np.ndarray:
    """Crop-cascade task of `--models all` (TS/python_api.py:670-757): a rough `total` segmentation at 6 mm (single model
    Dataset298; 3 mm / Dataset297 with robust_crop: `rough_resample`), labels = the `total` map -> crop mask = union of the
    `crop_names` structures -> the task's own model at native resolution on the cropped image -> labels on the input grid
    -> optionally cleared outside the dilated union of the `remove_outside` structures (TS/nnunet.py:711-716).
    rough_models / task_models: [(task_id, ModelConfig, [weight blob per fold])] as `model_store.load_task_models` gives."""
    rough = SegmentationTask(ctx, "total", rough_models, resample=rough_resample, multimodel=False, max_batch=max_batch, precision=precision)
    try:
        organ_seg = rough.predict_image(data, affine)
    finally:
        rough.close()
    inv = label_maps.CLASS_MAP_TOTAL_INV
    crop_mask = np.isin(organ_seg, [inv[n] for n in crop_names]).astype(np.uint8)
    t = SegmentationTask(ctx, task, task_models, resample=None, multimodel=False, max_batch=max_batch, precision=precision)
    try:
        seg = t.predict_image(data, affine, crop_mask=crop_mask, crop_addon=crop_addon)
    finally:
        t.close()
    if remove_outside_dilation is not None:
        remove_mask = np.isin(organ_seg, [inv[n] for n in (remove_outside or [])]).astype(np.uint8)
        # header zooms are float32 (nibabel): int(mm / np.mean(img.header.get_zooms()))
        zooms = np.asarray(orientation.zooms_from_affine(np.asarray(affine, dtype=np.float64)), dtype=np.float32)
        vx = int(remove_outside_dilation / np.mean(zooms))
        seg = remove_outside_of_mask(ctx, seg, remove_mask, addon=vx)
    return seg
