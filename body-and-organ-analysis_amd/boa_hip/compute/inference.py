"""File-level mirror of BOA/compute/inference.py: `compute_all_models` (:50-144) with the reference's signature,
outputs and folder contract, on the device engine.

Supported models: `total`, `bca`, `body_parts`, `body_regions` and the crop-cascade models of `--models all`
(lung_vessels, cerebral_bleed, hip_implant, pleural_pericard_effusion, liver_vessels, and the licensed
heartchambers_highres: rough 6 mm / robust 3 mm `total` -> crop -> native-resolution model [-> remove_outside_of_mask],
TS/python_api.py:670-757).  Anything else raises NotImplementedError -- nothing is silently skipped.  Weights come from
`$nnUNet_results` (boa_hip/model_store.py); nothing is downloaded.
"""
from __future__ import annotations

import json
import logging
import pathlib
from typing import Any, Dict, Iterable, Optional, Union

import numpy as np

from .. import label_maps, model_store, nifti, orientation
from ..device import Context
from ..pipeline import BcaPipelineHip
from ..task import SegmentationTask, run_cascade_task
from .config import resolve_device
from .constants import BASE_MODELS
from .measurements import compute_measurements
from .util import convert_resampling_slices

logger = logging.getLogger(__name__)

_CTX: Dict[int, Context] = {}
BODY_PARTS_MAP = {1: "torso", 2: "head", 3: "arms", 4: "legs", 5: "others", 6: "breasts"}   # BCA/body_parts/definition.py


def get_context(device: Optional[str] = None) -> Context:
    """One Context per GPU per process.  device: the reference's `device` parameter ("gpu", "gpu:1", "hip", ...)."""
    dev = resolve_device(device)
    if dev in ("cpu", "mps"):
        raise RuntimeError(f"device {dev!r} requested: this engine only runs on an MI355X (no CPU fallback)")
    idx = int(dev.split(":")[1]) if ":" in dev else 0
    if idx not in _CTX or _CTX[idx].h is None:
        _CTX[idx] = Context(idx)
    return _CTX[idx]


def range_warning(ct_image_data: np.ndarray) -> None:
    if np.any(ct_image_data < -1024) or np.any(ct_image_data > 3071):
        logger.warning("Unexpected CT values found in input image: got %s-%s, expected -1024-3071. The values have been "
                       "clipped to the expected range. Please check the segmentations to ensure that everything is correct.",
                       np.min(ct_image_data), np.max(ct_image_data))


def print_and_collect_image_info(ct_path: pathlib.Path):
    data, affine, hdr = nifti.load(ct_path)
    if data.ndim != 3:
        raise ValueError(f"Only 3D CT scans are supported not {data.ndim}D.")
    logger.info("Input image:   %s", ct_path)
    logger.info("Image size:    %s", hdr.get_data_shape())
    logger.info("Image dtype:   %s", hdr.get_data_dtype())
    logger.info("Voxel spacing: %s", hdr.get_zooms())
    logger.info("Input Axcodes: %s", orientation.aff2axcodes(affine))
    lps, laff = orientation.with_axcodes(data, affine, "LPS")
    range_warning(nifti.fdata(data, hdr))
    return lps.shape, tuple(orientation.zooms_from_affine(laff))


def _ct_array(data, hdr):
    """What the segmentation tasks see: get_fdata() values; integer-valued unscaled files stay int16 (same values)."""
    f = nifti.fdata(data, hdr)
    if data.dtype == np.int16 and np.array_equal(f, data):
        return data
    return f


def compute_all_models(
    ct_path: pathlib.Path,
    segmentation_folder: pathlib.Path,
    models_to_compute: Union[Iterable[str], str],
    totalsegmentator_params: Dict[str, Any],
    fast_bca: bool = False,
    bca_params: Optional[Dict[str, Any]] = None,
    force_split_threshold: int = 400,
    recompute: bool = True,
    cnr_adjustment: bool = True,
) -> Dict[str, int]:
    ct_path, segmentation_folder = pathlib.Path(ct_path), pathlib.Path(segmentation_folder)
    totalsegmentator_params = dict(totalsegmentator_params or {})
    bca_params = dict(bca_params or {})
    totalsegmentator_params.pop("preview", None)   # previews are rendering, not on this path
    models_to_compute = [models_to_compute] if isinstance(models_to_compute, str) else list(models_to_compute)
    shape, spacing = print_and_collect_image_info(ct_path)
    measurement_models = [m for m in models_to_compute if m not in BASE_MODELS]
    stats = {
        "num_voxels": int(shape[0] * shape[1] * shape[2]),
        "num_slices": int(shape[2]),
        "num_slices_resampled": convert_resampling_slices(slices=shape[-1], current_sampling=spacing[-1], target_resampling=1.5),
    }
    unsupported = [m for m in measurement_models if m != "total" and m not in model_store.CASCADE_MODELS]
    if unsupported:
        raise NotImplementedError(f"models {unsupported} are not implemented on the device")
    segmentation_folder.mkdir(parents=True, exist_ok=True)
    if totalsegmentator_params.get("nr_thr_saving"):      # the reference's saving workers = deflate threads of nifti.save here
        nifti.SAVE_THREADS = max(1, int(totalsegmentator_params["nr_thr_saving"]))
    ctx = get_context(totalsegmentator_params.get("device"))
    data, affine, hdr = nifti.load(ct_path)
    ct = _ct_array(data, hdr)
    fast = bool(totalsegmentator_params.get("fast", False))
    for name in measurement_models:
        _segment_one(ctx, name, ct, affine, hdr, segmentation_folder, fast, recompute)
    _write_total_measurements(ctx, ct_path, segmentation_folder, measurement_models, cnr_adjustment, recompute)
    bca_slices = convert_resampling_slices(slices=shape[-1], current_sampling=spacing[-1], target_resampling=5.0)
    for name in sorted(BASE_MODELS & set(models_to_compute)):
        _run_bca_model(ctx, name, ct, affine, hdr, segmentation_folder, fast_bca, bca_params,
                       split=bca_slices > force_split_threshold, slices=bca_slices, threshold=force_split_threshold,
                       recompute=recompute)
    return stats


def _segment_one(ctx, name, ct, affine, hdr, folder, fast, recompute):
    """One TotalSegmentator-style model -> `<name>.nii.gz` with the label table in the header extension."""
    target = folder / f"{name}.nii.gz"
    logger.info("Computing model %s...", name)
    if target.is_file() and not recompute:
        logger.info("The model was already computed, skipping...")
        return
    if name == "total":
        key = "total_fast" if fast else "total"
        task = SegmentationTask(ctx, "total", model_store.load_task_models(key), resample=model_store.TASKS[key]["resample"],
                                multimodel=not fast)
        try:
            seg = task.predict_image(ct, affine)
        finally:
            task.close()
        names = label_maps.CLASS_MAP_TOTAL
    else:
        if fast:
            raise ValueError(f"task {name} does not work with option --fast")   # TS/python_api.py:242 ff.
        info = model_store.TASKS[name]
        rough = model_store.rough_model_key(name)
        seg = run_cascade_task(ctx, name, ct, affine, model_store.load_task_models(rough),
                               model_store.load_task_models(name), info["crop"], model_store.effective_crop_addon(name),
                               rough_resample=model_store.TASKS[rough]["resample"], remove_outside=info.get("remove_outside"),
                               remove_outside_dilation=info.get("remove_outside_dilation"))
        names = label_maps.class_map(name)
    nifti.save(target, seg, affine, like=hdr, extensions=[(0, nifti.label_xml(names))])


def _write_total_measurements(ctx, ct_path, folder, models, cnr_adjustment, recompute):
    target = folder / "total-measurements.json"
    if not models or (target.is_file() and not recompute):
        logger.info("The total measurements were already computed, skipping...")
        return
    table = compute_measurements(ct_path=ct_path, segmentation_folder=folder, models=models, cnr_adjustment=cnr_adjustment, ctx=ctx)
    with target.open("w") as fh:
        json.dump(table, fh, indent=2)


def _run_bca_model(ctx, name, ct, affine, hdr, folder, fast_bca, bca_params, split, slices, threshold, recompute=True):
    """`bca` (both nets + tissues + tables) or one of its two networks on its own.  As BCA/infer/infer.py:58-61 an
    existing `<task>.nii.gz` is reloaded instead of recomputed when `recompute` is False (it already went through the
    task's post-processing when it was written), and only the network(s) that are needed get loaded."""
    if split:
        logger.info("Splitting the image into parts as the number of slices %s is more than %s", slices, threshold)
    wanted = ("body_parts", "body_regions") if name == "bca" else (name,)
    done = {}
    for task in wanted:
        f = folder / f"{task}.nii.gz"
        if not recompute and f.is_file():
            logger.info("Loading already computed %s...", task)
            done[task] = np.ascontiguousarray(nifti.load(f)[0], dtype=np.uint8)
    if name != "bca" and name in done:
        return
    models = {t: (model_store.load_task_models(t, fast_bca)[0][1:3] if t not in done else None) for t in wanted}
    pipe = BcaPipelineHip(ctx, models.get("body_parts"), models.get("body_regions"), fast_bca=fast_bca)
    try:
        if name != "bca":
            nifti.save(folder / f"{name}.nii.gz", pipe.inference(name, ct, affine, force_split=split), affine, like=hdr)
            return
        total_file = folder / "total.nii.gz"
        out = pipe.run(ct, affine, total_seg=nifti.load(total_file)[0] if total_file.is_file() else None, force_split=split,
                       median_filtering=bool(bca_params.get("median_filtering", False)),
                       examined_body_region=bca_params.get("examined_body_region"),
                       done_parts=done.get("body_parts"), done_regions=done.get("body_regions"))
        for volume in ("body_parts", "body_regions", "tissues"):
            if volume in done:
                continue
            nifti.save(folder / f"{volume}.nii.gz", out[volume], affine, like=hdr)
        if out["vertebrae"]:
            with (folder / "vertebrae.json").open("w") as fh:
                json.dump(out["vertebrae"], fh, indent=2)
        with (folder / "bca-measurements.json").open("w") as fh:
            json.dump(out["bca_measurements"], fh, indent=2, default=float)
    finally:
        pipe.close()
