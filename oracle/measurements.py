"""Oracle: per-label HU measurements (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows BOA/compute/measurements.py.  Arrays are (z, y, x); spacing is sitk (x, y, z).
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage

ADIPOSE_TISSUE = (-200, -40)


def create_mask(region_data, labels):
    """BOA/compute/util.py:25-31."""
    if isinstance(labels, (int, np.integer)):
        return region_data == labels
    return np.isin(region_data, labels)


def get_region_minus_fat(ct, mask):
    """BOA/compute/measurements.py:29-39."""
    return np.logical_and(mask, np.logical_or(ct < ADIPOSE_TISSUE[0], ct > ADIPOSE_TISSUE[1]))


def erode_region(mask, kernel_value=6):
    """BOA/compute/measurements.py:61-71: skimage binary_erosion with a 6^3 ones footprint padded at the
    end to 7^3 (pad_footprint(pad_end=True)).  PARITY UNPINNED vs skimage 0.26 (absent here): restated as
    scipy.ndimage.binary_erosion with the same centred 7^3 structure (ones in [0:6]^3), border_value
    True (skimage's erosion pads with the maximum, so the image border does not erode)."""
    k = np.ones([kernel_value] * 3, dtype=bool)
    if kernel_value % 2 == 0:
        k = np.pad(k, [(0, 1)] * 3)
    return ndimage.binary_erosion(mask, structure=k, border_value=1)


def autochthon_reference(ct, right_mask, left_mask):
    """BOA/compute/measurements.py:42-58."""
    m = erode_region(get_region_minus_fat(ct, np.logical_or(right_mask, left_mask)))
    if m.sum() == 0:
        return None, None
    return float(np.mean(ct[m])), float(np.std(ct[m]))


def metrics_for_region(ct, mask, auto_mean, auto_std, spacing, cnr_adjustment=False, region_name=""):
    """BOA/compute/measurements.py:74-123."""
    out = {}
    if np.sum(mask) == 0:
        return {"present": False}
    if cnr_adjustment:
        if "autochthon" in region_name:
            mask = get_region_minus_fat(ct, mask)
        mask = erode_region(mask)
    if np.sum(mask) == 0:
        return {"present": False}
    ml = np.prod(spacing) / 1000.0
    out["present"] = True
    hu = ct[mask]
    out["volume_ml"] = np.sum(mask) * ml
    out["mean_hu"] = float(np.mean(hu))
    out["std_hu"] = float(np.std(hu))
    out["min_hu"] = float(np.min(hu))
    out["median_hu"] = float(np.median(hu))
    out["max_hu"] = float(np.max(hu))
    for p in (25, 75):
        out[f"{p}th_percentile_hu"] = float(np.percentile(hu, p))
    if auto_mean is not None and auto_std is not None:
        if cnr_adjustment and region_name.partition("_")[0] == "autochthon":
            out["cnr"] = None
        else:
            out["cnr"] = (np.mean(hu) - auto_mean) / auto_std
    else:
        out["cnr"] = None
    return out


def metrics_for_each_region(ct, region_data, label_map, auto_mean, auto_std, spacing, cnr_adjustment=False):
    """BOA/compute/measurements.py:203-241."""
    res = {}
    for region, label in label_map.items():
        res[region] = metrics_for_region(ct, create_mask(region_data, label), auto_mean, auto_std, spacing,
                                         cnr_adjustment, region)
    if "autochthon_left" in label_map and "autochthon_right" in label_map:
        m = create_mask(region_data, [label_map["autochthon_left"], label_map["autochthon_right"]])
        res["autochthon"] = metrics_for_region(ct, m, auto_mean, auto_std, spacing, cnr_adjustment, "autochthon")
    return res


LUNG_MASKS = ["lung_upper_lobe_left", "lung_lower_lobe_left", "lung_upper_lobe_right",
              "lung_middle_lobe_right", "lung_lower_lobe_right"]


def ct_pfav(ct, region_data, label_map, auto_mean, auto_std, spacing):
    """BOA/compute/measurements.py:126-200 -> (measurements, fat_mask uint8 of all lungs)."""
    def lung(ids):
        m = create_mask(region_data, ids)
        fat = np.logical_and(m, np.logical_and(ct >= ADIPOSE_TISSUE[0], ct <= ADIPOSE_TISSUE[1]))
        return fat, metrics_for_region(ct, fat, auto_mean, auto_std, spacing)

    out = {}
    for name in LUNG_MASKS:
        _, out["ct_pfav_" + name] = lung([label_map[name]])
    for side in ("left", "right"):
        _, out[f"ct_pfav_lobe_{side}"] = lung([label_map[n] for n in LUNG_MASKS if n.endswith(side)])
    fat, out["ct_pfav_lungs"] = lung([label_map[n] for n in LUNG_MASKS])
    return out, fat.astype(np.uint8)


def total_measurements(ct, total_seg, label_map, spacing, cnr_adjustment=True,
                       cnr_regions=("aorta", "autochthon_left", "autochthon_right")):
    """BOA/compute/measurements.py:244-343 for models == ["total"] (array part)."""
    meas = {"segmentations": {}, "info": {}}
    am, asd = autochthon_reference(ct, create_mask(total_seg, label_map["autochthon_right"]),
                                   create_mask(total_seg, label_map["autochthon_left"]))
    seg = metrics_for_each_region(ct, total_seg, label_map, am, asd, spacing)
    pf, fat = ct_pfav(ct, total_seg, label_map, am, asd, spacing)
    meas["segmentations"]["total"] = {**seg, **pf}
    if cnr_adjustment and am is not None and asd is not None:
        sub = {r: v for r, v in label_map.items() if r in cnr_regions}
        meas.setdefault("cnr_adjusted", {}).update(
            metrics_for_each_region(ct, total_seg, sub, am, asd, spacing, cnr_adjustment=True))
    meas["info"]["autochthon_mean"] = am
    meas["info"]["autochthon_std"] = asd
    return meas, fat
