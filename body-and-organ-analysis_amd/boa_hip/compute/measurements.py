"""File-level mirror of BOA/compute/measurements.py:compute_measurements (:244-343): reads the CT and the
segmentation NIfTIs of the output folder, runs the per-label HU statistics on the device (boa_hip/measurements.py)
and writes ct_pfav.nii.gz as `ct_pfav` does (:196-198)."""
from __future__ import annotations

import logging
import pathlib
from typing import Any, Dict, List, Optional

import numpy as np

from .. import label_maps, nifti
from .. import measurements as M
from ..devarray import DevArray
from ..device import Context

logger = logging.getLogger(__name__)


def compute_measurements(ct_path: pathlib.Path, segmentation_folder: pathlib.Path, models: List[str],
                         cnr_adjustment: bool, ctx: Optional[Context] = None) -> Dict[str, Any]:
    measurements: Dict[str, Any] = {"segmentations": {}, "info": {}}
    if len(models) == 0:
        return measurements
    logger.info("Computing measurements for the computed segmentations: %s", models)
    if ctx is None:
        from .inference import get_context
        ctx = get_context(None)
    segmentation_folder = pathlib.Path(segmentation_folder)

    def path_of(model_name):
        return segmentation_folder / f"{'total' if model_name == 'total' else label_maps.output_name(model_name)}.nii.gz"

    # every segmentation file of the run is read and inflated on worker threads while the CT is uploaded and the earlier models
    # are measured (the reference reads them one after the other, measurements.py:257-270)
    from concurrent.futures import ThreadPoolExecutor
    ordered = sorted(models, key=lambda m: m != "total")
    pool = ThreadPoolExecutor(max_workers=max(1, min(4, len(ordered))))
    pending = {m: pool.submit(nifti.load, path_of(m)) for m in ordered if path_of(m).exists()}
    pool.shutdown(wait=False)
    data, _, hdr = nifti.load(ct_path)
    # SimpleITK view (z,y,x) of the file, int16 HU, made on the device (a 512^3 host transpose costs ~0.5 s)
    from .util import require_int16_exact
    d_file = DevArray.from_numpy(ctx, data if (data.dtype == np.int16 and not nifti.is_scaled(hdr))
                                 else require_int16_exact(nifti.fdata(data, hdr), str(ct_path)))
    d_ct = d_file.transpose((2, 1, 0)).contiguous(np.int16, force_copy=True)
    d_file.free()
    spacing = tuple(float(v) for v in hdr.get_zooms())
    am = asd = None
    for model_name in ordered:
        # (the reference looks for ADDITIONAL_MODELS_OUTPUT_NAME[model] while inference writes <model>.nii.gz, so e.g.
        #  lung_vessels -> lung_vessels_airways.nii.gz is never found and silently skipped: kept as is, :263-270)
        if model_name not in pending:
            continue
        seg, saff, shdr = pending.pop(model_name).result()
        if not np.isclose(spacing, tuple(float(v) for v in shdr.get_zooms())).all():
            raise ValueError("The spacing of the image and of the segmentation should be the same")
        label_map = label_maps.measurement_label_map(model_name)
        d_sfile = DevArray.from_numpy(ctx, np.ascontiguousarray(seg, dtype=np.uint8))
        d_seg = d_sfile.transpose((2, 1, 0)).contiguous(force_copy=True)
        d_sfile.free()
        try:
            if d_seg.shape != d_ct.shape:
                raise ValueError("The spacing of the image and of the segmentation should be the same")
            if model_name == "total":
                meas, d_mask = M.total_measurements(ctx, None, None, label_map, spacing, cnr_adjustment=cnr_adjustment,
                                                    d_ct=d_ct.buf, d_lab=d_seg.buf, shape=d_ct.shape, mask_on_device=True)
                measurements["segmentations"].update(meas["segmentations"])
                if "cnr_adjusted" in meas:
                    measurements["cnr_adjusted"] = meas["cnr_adjusted"]
                am, asd = meas["info"].get("autochthon_mean"), meas["info"].get("autochthon_std")
                fat = DevArray(ctx, d_mask, d_ct.shape, np.uint8)
                try:
                    nifti.save(segmentation_folder / "ct_pfav.nii.gz", fat.transpose((2, 1, 0)).download(), saff, like=shdr)
                finally:
                    fat.free()
            else:
                hist = M.label_hu_histogram(ctx, d_ct.buf, d_seg.buf, d_ct.size)
                measurements["segmentations"][model_name] = M._metrics_from_hist(hist, label_map, am, asd, spacing)
                if cnr_adjustment and model_name in label_maps.cnr_adjusted_regions():
                    if am is None or asd is None:   # (:307-313)
                        logger.warning("Skipping CNR-adjusted measurements for %s: autochthon reference is unavailable (the "
                                       "'total' model did not run or did not produce a usable autochthon mask).", model_name)
                    else:
                        adj = M.cnr_adjusted_region_metrics(ctx, d_ct.buf, d_seg.buf, d_ct.shape, label_map,
                                                            label_maps.cnr_adjusted_regions()[model_name], hist, am, asd, spacing)
                        measurements.setdefault("cnr_adjusted", {}).update(adj)
        finally:
            d_seg.free()
    d_ct.free()
    measurements["info"]["autochthon_mean"] = am
    measurements["info"]["autochthon_std"] = asd
    return measurements
