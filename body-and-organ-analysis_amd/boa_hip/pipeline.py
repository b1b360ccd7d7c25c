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

import os
import time
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import bca, label_maps, orientation
from .devarray import DevArray
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


_PROF = bool(os.environ.get("BOA_PIPE_PROF"))


class _Stage:
    """`with _Stage(ctx, "name"):` prints the wall time of a pipeline stage when BOA_PIPE_PROF is set."""

    def __init__(self, ctx, name):
        self.ctx, self.name = ctx, name

    def __enter__(self):
        if _PROF:
            self.ctx.sync()
            self.t = time.perf_counter()

    def __exit__(self, *a):
        if _PROF:
            self.ctx.sync()
            print(f"[pipe] {self.name}: {time.perf_counter() - self.t:.3f} s", flush=True)


class BcaPipelineHip:
    """parts_model / regions_model: (ModelConfig, [weight blob per fold])."""

    def __init__(self, ctx: Context, parts_model: Tuple[ModelConfig, Sequence[np.ndarray]],
                 regions_model: Tuple[ModelConfig, Sequence[np.ndarray]], fast_bca: bool = False, max_batch: int = 16,
                 parts_ctx: Optional[Context] = None, precision: Optional[str] = None):
        """`parts_ctx`: a second Context of the same GPU (own stream and pool).  The body_parts net and its post-processing
        then run on it, driven by a worker thread, while body_regions runs on `ctx` (the two only meet in the tissue stage;
        with `crop_body` the second net needs the first one's output and the pipeline stays on one stream).  Same kernels on
        the same data either way: the results are bit-identical."""
        self.ctx = ctx
        self.parts_ctx = parts_ctx if parts_ctx is not ctx else None
        if self.parts_ctx is not None and self.parts_ctx.device != ctx.device:
            raise ValueError("BcaPipelineHip: parts_ctx must be on the same GPU")
        self.tasks: Dict[str, SegmentationTask] = {}
        for name, model in (("body_parts", parts_model), ("body_regions", regions_model)):
            if model is None:      # this task's output is reloaded from an earlier run (recompute=False): no network needed
                continue
            cfg, blobs = model
            info = get_task_info(name, fast_bca)
            blobs = list(blobs)[:len(info["folds"])]
            if len(blobs) != len(info["folds"]):
                raise ValueError(f"{name}: {len(info['folds'])} folds expected, {len(blobs)} weight sets given")
            tctx = self.parts_ctx if (name == "body_parts" and self.parts_ctx is not None) else ctx
            self.tasks[name] = SegmentationTask(tctx, name, [(info["task_id"], cfg, blobs)], resample=info["resample"],
                                                resample_only_thickness=True, multimodel=False, max_batch=max_batch, precision=precision)

        # (agg_shard.AggComm, tile_shard.ShardComm): several ranks share every volume also in the aggregation half -- CC
        # filters of body_regions, tissue pass and per-slice tables run per z-slab; None = whole volumes here
        self.agg = None

    def close(self):
        for t in self.tasks.values():
            t.close()

    def _inference_device(self, task_name: str, d_ct: DevArray, affine: np.ndarray, force_split: bool, crop, raw,
                          done=None) -> DevArray:
        """BCA/infer/infer.py:39-89 on resident data: network labels on the input grid (file axis order), then the task's
        post-processing applied to the SimpleITK view (z,y,x) of the file; returns the cleaned labels in file order."""
        ctx = self.tasks[task_name].ctx if task_name in self.tasks else self.ctx
        if done is not None:   # `inference(recompute=False)` found <task>.nii.gz: already post-processed (infer.py:58-61)
            return DevArray.from_numpy(ctx, np.ascontiguousarray(done, dtype=np.uint8))
        if raw is None:
            with _Stage(ctx, f"{task_name}: networks"):
                d_raw = self.tasks[task_name].predict_image(d_ct, affine, force_split=force_split, crop_mask=crop, return_device=True)
        else:
            d_raw = DevArray.from_numpy(ctx, np.ascontiguousarray(raw, dtype=np.uint8))
        with _Stage(ctx, f"{task_name}: post-processing"):
            zyx = d_raw.transpose((2, 1, 0)).contiguous(force_copy=True)
            d_raw.free()
            if task_name == "body_parts":
                # (the network's label values are known from its class count: no presence query, no host round trip)
                task = self.tasks.get(task_name) if raw is None else None
                labels = range(1, int(task.parts[0][1].geometry.num_classes)) if task is not None and not task.multimodel else None
                buf = bca.postprocess_part_segmentation_device(ctx, zyx.buf, zyx.shape, labels=labels)
                zyx.free()
                zyx = DevArray(ctx, buf, zyx.shape, np.uint8)
            elif task_name == "body_regions":
                if self.agg is not None and self.agg[0].world > 1:
                    bca.postprocess_region_segmentation_device_sharded(ctx, self.agg, zyx.buf, zyx.shape)
                else:
                    bca.postprocess_region_segmentation_device(ctx, zyx.buf, zyx.shape)
            else:
                zyx.free()
                raise ValueError(task_name)
            out = zyx.transpose((2, 1, 0)).contiguous(force_copy=True)
            zyx.free()
            return out

    def inference(self, task_name: str, ct: np.ndarray, affine: np.ndarray, force_split: bool = False,
                  crop: Optional[np.ndarray] = None, raw: Optional[np.ndarray] = None) -> np.ndarray:
        """`inference()` for one BCA task on host arrays (file axis order) -> cleaned labels (file axis order)."""
        d_ct = DevArray.from_numpy(self.tasks[task_name].ctx if task_name in self.tasks else self.ctx, SegmentationTask._supported(ct))
        try:
            out = self._inference_device(task_name, d_ct, np.asarray(affine, dtype=np.float64), force_split, crop, raw)
            try:
                return out.download()
            finally:
                out.free()
        finally:
            d_ct.free()

    def _both_nets_two_streams(self, d_ct, affine, force_split, raw_regions, done_regions, keep, crop_body=False):
        """body_parts on `parts_ctx` (worker thread), body_regions on `ctx` (calling thread); hand-over by value: the CT is
        copied into the second pool before the net starts, the cleaned labels back after that stream synchronised.
        `crop_body`: body_regions needs the body mask first -- the two nets then run one after the other."""
        import threading
        ctx, pctx = self.ctx, self.parts_ctx
        ctx.sync()                                    # the CT is complete before the other stream copies it
        ct_p = d_ct.to_context(pctx)
        box: dict = {}

        def lane():
            try:
                pctx.bind_thread()
                box["parts"] = self._inference_device("body_parts", ct_p, affine, force_split, None, None, None)
                pctx.sync()
            except BaseException as e:  # noqa: BLE001  (re-raised on the calling thread)
                box["error"] = e

        def parts_here():
            if "error" in box:
                raise RuntimeError("the body_parts lane failed") from box["error"]
            d = box["parts"].to_context(ctx)
            keep.append(d)
            ctx.sync()                                # the copy is done before its source goes back to the other pool
            box.pop("parts").free()
            return d

        th = threading.Thread(target=lane, name="boa-lane-body-parts")
        th.start()
        d_parts = None
        try:
            if crop_body:
                th.join()
                d_parts = parts_here()
            crop = d_parts.download() if crop_body else None
            d_regions = self._inference_device("body_regions", d_ct, affine, force_split, crop, raw_regions, done_regions)
            keep.append(d_regions)
        except BaseException:
            th.join()
            if "parts" in box:             # the lane finished but this thread failed: its result must not outlive the call
                box.pop("parts").free()
            raise
        finally:
            th.join()
            ct_p.free()
            if "error" in box and "parts" in box:
                box.pop("parts").free()
        if d_parts is None:
            d_parts = parts_here()
        return d_parts, d_regions

    def _lps_zyx(self, d: DevArray, affine: np.ndarray, dtype=None) -> DevArray:
        """BCA/io.py:78-94 `process_image` as a device view: reorient to LPS, SimpleITK array order (z,y,x), contiguous."""
        cur = "".join(orientation.aff2axcodes(affine))
        v = d if cur == "LPS" else d.apply_orientation(orientation.ornt_transform(orientation.axcodes2ornt(cur), orientation.axcodes2ornt("LPS")))
        return v.transpose((2, 1, 0)).contiguous(dtype, force_copy=True)

    def run(self, ct: np.ndarray, affine: np.ndarray, total_seg: Optional[np.ndarray] = None,
            median_filtering: bool = False, examined_body_region: Optional[str] = None, crop_body: bool = False,
            force_split: bool = False, raw_parts: Optional[np.ndarray] = None, raw_regions: Optional[np.ndarray] = None,
            done_parts: Optional[np.ndarray] = None, done_regions: Optional[np.ndarray] = None) -> dict:
        """-> {"body_parts", "body_regions", "tissues" (file axis order, uint8), "bca_measurements" (dict),
        "vertebrae" (dict)}.  `total_seg`: the `total` label volume on the same grid (vertebra groups), optional.
        The CT is uploaded once; every stage (nets, post-processing, LPS reload, tissues, tables) works on resident
        buffers (`run_resident`), the three label volumes are downloaded at the end."""
        if np.asarray(ct).dtype != np.int16:
            from .compute.util import require_int16_exact
            require_int16_exact(ct, "bca: CT")    # tissue rules / HU sums run on int16 HU: never truncate or wrap silently
        d_ct = DevArray.from_numpy(self.ctx, SegmentationTask._supported(ct))
        d_tot = None if total_seg is None else DevArray.from_numpy(self.ctx, np.ascontiguousarray(total_seg, dtype=np.uint8))
        res = None
        try:
            res = self.run_resident(d_ct, affine, d_tot, median_filtering, examined_body_region, crop_body, force_split, raw_parts,
                                    raw_regions, done_parts, done_regions)
            return {"body_parts": res["body_parts"].download(), "body_regions": res["body_regions"].download(),
                    "tissues": res["tissues"].download(), "bca_measurements": res["bca_measurements"],
                    "vertebrae": res["vertebrae"], "examined_body_part": res["examined_body_part"]}
        finally:
            for a in [d_ct, d_tot] + ([res[k] for k in ("body_parts", "body_regions", "tissues")] if res else []):
                if a is not None:
                    a.free()

    def run_resident(self, d_ct: DevArray, affine: np.ndarray, d_total: Optional[DevArray] = None,
                     median_filtering: bool = False, examined_body_region: Optional[str] = None, crop_body: bool = False,
                     force_split: bool = False, raw_parts: Optional[np.ndarray] = None, raw_regions: Optional[np.ndarray] = None,
                     done_parts: Optional[np.ndarray] = None, done_regions: Optional[np.ndarray] = None) -> dict:
        """`run` on device-resident inputs (CT and `total` labels in the file's axis order; neither is freed here): the three
        label volumes come back as contiguous DevArrays in file axis order (the caller frees them), the tables as dicts.
        `d_total` may be a callable returning the DevArray: it is called when the vertebra table needs the `total` labels,
        i.e. after both nets and their post-processing (boa_hip/lanes.py runs `total` on a second stream meanwhile)."""
        ctx = self.ctx
        affine = np.asarray(affine, dtype=np.float64)
        live = []
        keep = []
        # both nets see the same CT at the same (sx, sy, 5 mm) grid: the cubic resampling runs once (unless the second net
        # works on a body crop)
        rs_cache: dict = {}
        two_streams = self.parts_ctx is not None and raw_parts is None and done_parts is None and "body_parts" in self.tasks
        if not two_streams:
            for t in self.tasks.values():
                t.resample_cache = rs_cache
        try:
            if two_streams:
                d_parts, d_regions = self._both_nets_two_streams(d_ct, affine, force_split, raw_regions, done_regions, keep, crop_body)
            else:
                d_parts = self._inference_device("body_parts", d_ct, affine, force_split, None, raw_parts, done_parts)
                keep.append(d_parts)
                crop = d_parts.download() if crop_body else None
                d_regions = self._inference_device("body_regions", d_ct, affine, force_split, crop, raw_regions, done_regions)
                keep.append(d_regions)
            with _Stage(ctx, "LPS reload, body-part flags, vertebrae, tissues + tables, JSON"):
                _, laff = orientation.with_axcodes(np.empty(d_ct.shape, dtype=np.uint8), affine, "LPS")
                sp = np.sqrt(np.sum(np.asarray(laff, dtype=np.float64)[:3, :3] ** 2, axis=0))
                spacing = (float(sp[0]), float(sp[1]), float(sp[2]))
                ct_l = self._lps_zyx(d_ct, affine, np.int16)
                rg_l = self._lps_zyx(d_regions, affine)
                pt_l = self._lps_zyx(d_parts, affine)
                live += [ct_l, rg_l, pt_l]
                present = bca.slice_label_presence(ctx, rg_l.buf, rg_l.shape)
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
                if callable(d_total):
                    d_total = d_total()
                if d_total is not None:
                    tot_l = self._lps_zyx(d_total, affine)
                    live.append(tot_l)
                    vertebrae = bca.create_vertebrae_info(ctx, None, label_maps.CLASS_MAP_TOTAL, flags, d_total=tot_l.buf,
                                                          shape=tot_l.shape)
                if self.agg is not None and self.agg[0].world > 1 and not median_filtering:
                    js, d_tis = bca.bca_measurements_device_sharded(ctx, self.agg, ct_l.buf, rg_l.buf, pt_l.buf, ct_l.shape, spacing,
                                                                    vertebrae or None, "LPS", flags if examined_body_region else None)
                else:
                    js, d_tis = bca.bca_measurements_device(ctx, ct_l.buf, rg_l.buf, pt_l.buf, ct_l.shape, spacing, vertebrae or None,
                                                            True, median_filtering, "LPS", flags if examined_body_region else None)
                tis_l = DevArray(ctx, d_tis, ct_l.shape, np.uint8)
                live.append(tis_l)
                # back to the file's axis order: (z,y,x) LPS -> (x,y,z) LPS -> file orientation
                back = orientation.ornt_transform(orientation.axcodes2ornt("LPS"), orientation.io_orientation(affine))
                tissues = tis_l.transpose((2, 1, 0)).apply_orientation(back).contiguous(force_copy=True)
                keep.append(tissues)
            out = {"body_parts": d_parts, "body_regions": d_regions, "tissues": tissues, "bca_measurements": js,
                   "vertebrae": vertebrae, "examined_body_part": flags}
            keep = []
            return out
        finally:
            for t in self.tasks.values():
                t.resample_cache = None
            for b in rs_cache.values():
                b.free()
            seen = set()
            for a in live + keep:
                if id(a.buf) not in seen:
                    seen.add(id(a.buf))
                    a.free()
