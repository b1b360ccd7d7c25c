"""Oracle: body-composition aggregations (TEST INFRASTRUCTURE, see oracle/__init__.py).

Arrays are SimpleITK order (z, y, x); spacing is sitk order (x, y, z).
"""
from __future__ import annotations

import numpy as np
import pandas as pd
from scipy import ndimage

# BCA/body_regions/definition.py:4-15, BCA/body_parts/definition.py:4-11, BCA/tissue/definition.py:6-30
REGION = dict(SUBCUTANEOUS_TISSUE=1, MUSCLE=2, ABDOMINAL_CAVITY=3, THORACIC_CAVITY=4, BONE=5, GLANDS=6,
              PERICARDIUM=7, BREAST_IMPLANT=8, MEDIASTINUM=9, BRAIN=10, NERVOUS_SYSTEM=11)
PART_TORSO = 1
TISSUES = [("MUSCLE", 1), ("BONE", 2), ("SAT", 3), ("VAT", 4), ("IMAT", 5), ("PAT", 6), ("EAT", 7)]
HU_ALL, HU_ADIPOSE, HU_MUSCLE = (-1000, 3000), (-190, -30), (-29, 150)
# enum order of TISSUE_DERIVATION_RULES: (tissue value, HU range, region value)
RULES = [
    (1, HU_MUSCLE, REGION["MUSCLE"]),
    (2, HU_ALL, REGION["BONE"]),
    (3, HU_ADIPOSE, REGION["SUBCUTANEOUS_TISSUE"]),
    (4, HU_ADIPOSE, REGION["ABDOMINAL_CAVITY"]),
    (5, HU_ADIPOSE, REGION["MUSCLE"]),
    (6, HU_ADIPOSE, REGION["MEDIASTINUM"]),
    (7, HU_ADIPOSE, REGION["PERICARDIUM"]),
]
COLS = ["Bone", "Muscle", "TAT", "IMAT", "SAT", "VAT", "PAT", "EAT"]


def _tname(name):
    return name.capitalize() if name in ("BONE", "MUSCLE") else name


def subclassify_tissues(image_data: np.ndarray, seg_data: np.ndarray, median_filtering=False,
                        slice_axis=0) -> np.ndarray:
    """BCA/tissue/subclassification.py:10-53."""
    if median_filtering:
        kernel = [3, 3, 3]
        kernel[slice_axis] = 1
        image_data = ndimage.median_filter(image_data, size=kernel)
    out = np.zeros_like(seg_data)
    for t, (lo, hi), region in RULES:
        m = np.logical_and(image_data >= lo, image_data <= hi) & (seg_data == region)
        out[m] = t
    return out


def slicewise_measurements(tissue_data, part_data, spacing_xyz):
    """BCA/report/builder.py:403-444: two DataFrames (all, no extremities)."""
    ml = np.prod(spacing_xyz) / 1000.0
    data = {_tname(n): (tissue_data == v).sum(axis=(1, 2)) * ml for n, v in TISSUES}
    df = pd.DataFrame(data)
    df["TAT"] = df.SAT + df.VAT + df.IMAT + df.PAT + df.EAT
    df["slice_idx"] = range(len(df))
    df = df[["slice_idx"] + COLS]
    torso = part_data == PART_TORSO
    data = {_tname(n): np.logical_and(torso, tissue_data == v).sum(axis=(1, 2)) * ml for n, v in TISSUES}
    d2 = pd.DataFrame(data)
    d2["TAT"] = d2.SAT + d2.VAT + d2.IMAT + d2.PAT + d2.EAT
    d2["slice_idx"] = range(len(d2))
    d2 = d2[["slice_idx"] + COLS]
    return df, d2


def examined_body_part(region_data, spacing_xyz, min_abdomen=200, min_neck=100, min_thorax=200):
    """BCA/report/builder.py:44-112 -> dict(abdomen, neck, thorax)."""
    thick = spacing_xyz[2]
    depth = region_data.shape[0]
    res = dict(abdomen=False, neck=False, thorax=False)
    abd = (region_data == REGION["ABDOMINAL_CAVITY"]).any(axis=(1, 2))
    sl = np.where(abd)[0]
    n_abd = sl.max() - sl.min() + 1 if sl.size else 0
    if n_abd * thick >= min_abdomen:
        res["abdomen"] = True
    med = np.where((region_data == REGION["MEDIASTINUM"]).any(axis=(1, 2)))[0]
    n_above = depth - med.max() if med.size else 0
    if n_above * thick >= min_neck:
        res["neck"] = True
    thx = np.isin(region_data, [REGION["THORACIC_CAVITY"], REGION["MEDIASTINUM"], REGION["PERICARDIUM"]]).any(
        axis=(1, 2))
    ts = np.where(thx)[0]
    inter = np.logical_and(abd, thx).any()
    n_thx = ts.max() - ts.min() + 1 if ts.size else 0
    if inter and n_thx * thick >= min_thorax:
        res["thorax"] = True
    return res


def aggregation_groups(region_data, depth, parts, vertebrae=None):
    """BCA/report/builder.py:170-216."""
    groups = [("Whole Scan", 0, depth)]

    def rng(mask):
        s = np.where(mask.any(axis=(1, 2)))[0]
        return int(s.min()), int(s.max()) + 1

    if parts["abdomen"]:
        groups.append(("Abdominal Cavity", *rng(region_data == REGION["ABDOMINAL_CAVITY"])))
    if parts["thorax"]:
        groups.append(("Thoracic Cavity", *rng(np.isin(
            region_data, [REGION["THORACIC_CAVITY"], REGION["MEDIASTINUM"], REGION["PERICARDIUM"]]))))
        groups.append(("Mediastinum", *rng(region_data == REGION["MEDIASTINUM"])))
        groups.append(("Pericardium", *rng(region_data == REGION["PERICARDIUM"])))
    if parts["abdomen"] and parts["thorax"]:
        groups.insert(1, ("Ventral Cavity", groups[1][1], groups[2][2]))
    if vertebrae:
        for name, g in vertebrae.items():
            groups.append((name, g[0], g[1]))
    return groups


def descriptive_statistics(slicewise: pd.DataFrame, image_data, tissue_data) -> pd.DataFrame:
    """BCA/report/builder.py:263-307."""
    sw = slicewise.drop("slice_idx", axis=1)
    m = sw.describe()
    m.drop("count", inplace=True)
    m.index = ["Mean", "StdDev", "Minimum", "25%", "Median", "75%", "Maximum"]
    m.loc["Total"] = sw.sum()
    for n, v in TISSUES:
        md = image_data[tissue_data == v]
        m.loc["MeanHU", _tname(n)] = np.mean(md) if md.size else None
    md = image_data[np.isin(tissue_data, [5, 3, 4, 6, 7])]
    m.loc["MeanHU", "TAT"] = np.mean(md) if md.size else None
    return m.replace({np.nan: None})


_ROW = {"Mean": "mean", "StdDev": "std", "Minimum": "min", "25%": "q1", "Median": "q2", "75%": "q3",
        "Maximum": "max", "Total": "sum", "MeanHU": "mean_hu"}


def bca_measurements_json(image_data, region_data, part_data, tissue_data, spacing_xyz, vertebrae=None):
    """Numeric content of bca-measurements.json: BCA/report/builder.py:397-444,163-261,520-598 and
    BCA/commands.py (examined_body_part = from_body_regions)."""
    df, d2 = slicewise_measurements(tissue_data, part_data, spacing_xyz)
    parts = examined_body_part(region_data, spacing_xyz)
    groups = aggregation_groups(region_data, image_data.shape[0], parts, vertebrae)
    torso = part_data == PART_TORSO
    agg = {}
    for name, lo, hi in groups:
        m = descriptive_statistics(df[(df.slice_idx >= lo) & (df.slice_idx < hi)], image_data[lo:hi],
                                   tissue_data[lo:hi])
        m2 = descriptive_statistics(d2[(d2.slice_idx >= lo) & (d2.slice_idx < hi)], image_data[lo:hi],
                                    np.where(torso[lo:hi], tissue_data[lo:hi], 0))
        key = name.lower().replace(" ", "_").replace("-", "_")
        agg[key] = {
            "num_slices": int(hi - lo), "min_slice_idx": int(lo), "max_slice_idx": int(hi),
            "measurements": m.rename(index=_ROW, columns={c: c.lower() for c in m.columns}).to_dict(),
            "measurements_no_extremities": m2.rename(index=_ROW, columns={c: c.lower() for c in m2.columns}).to_dict(),
        }

    def recs(d):
        return d.rename(columns={c: c.lower() for c in d.columns}).drop("slice_idx", axis=1).astype(float).to_dict(
            "records")

    return {"slices": recs(df), "slices_no_extremities": recs(d2), "aggregated": agg, "body_parts": parts}


def filter_largest_unique_segment(seg, mask):
    """BCA/body_regions/postprocess.py:8-15 with skimage.measure.label (default full connectivity = 26)
    restated via scipy.ndimage.label(structure=ones(3,3,3)).  Non-largest components -> 255.
    PARITY UNPINNED vs skimage (absent); tie-breaking among equal areas follows a stable sort by area
    descending over ascending label id, as `sorted(props, key=area, reverse=True)` does."""
    lab, n = ndimage.label(mask, structure=np.ones((3, 3, 3)))
    if n > 1:
        areas = np.bincount(lab.ravel())[1:]
        order = sorted(range(1, n + 1), key=lambda i: int(areas[i - 1]), reverse=True)
        for i in order[1:]:
            seg[lab == i] = 255


def postprocess_region_segmentation(seg):
    """BCA/body_regions/postprocess.py:18-40."""
    seg = seg.copy()
    filter_largest_unique_segment(seg, seg > 0)
    filter_largest_unique_segment(seg, (seg == REGION["THORACIC_CAVITY"]) | (seg == REGION["MEDIASTINUM"]) |
                                  (seg == REGION["PERICARDIUM"]))
    for r in (REGION["PERICARDIUM"], REGION["ABDOMINAL_CAVITY"]):
        filter_largest_unique_segment(seg, seg == r)
    return seg


def remove_small_labeled_objects(mask: np.ndarray, threshold: int = 3000) -> np.ndarray:
    """BCA/body_parts/postprocess.py:7-52.  cv2 and skimage are absent here (PARITY UNPINNED vs those libraries); restated
    from their documented semantics with scipy:
      * findContours(RETR_EXTERNAL) + drawContours(FILLED) per slice fills each outer contour of the 8-connected
        foreground, i.e. foreground plus all background not 4-connected to the slice border
        = scipy.ndimage.binary_fill_holes (default cross structuring element);
      * remove_small_objects(max_size=threshold-1, connectivity=3) clears 26-connected components with
        <= threshold-1 voxels."""
    out = np.zeros(mask.shape, dtype=mask.dtype)
    ones = np.ones((3, 3, 3))
    for label in np.unique(mask):
        if label <= 0:
            continue
        filled = np.stack([ndimage.binary_fill_holes(mask[i] == label) for i in range(mask.shape[0])])
        for _ in range(2):  # objects, then (after inversion) holes
            lab, n = ndimage.label(filled, structure=ones)
            sizes = np.bincount(lab.ravel())
            small = sizes <= threshold - 1
            small[0] = False
            filled[small[lab]] = False
            filled = ~filled
        out[filled] = label
    return out


def create_vertebrae_info(total_zyx: np.ndarray, vertebrae_map: dict, parts: dict) -> dict:
    """BCA/commands.py:24-45.  vertebrae_map: {"C1": label, ...}; parts: {"abdomen","thorax","neck"} flags."""
    info = {}
    for vid, label in vertebrae_map.items():
        idx = np.where((total_zyx == label).any(axis=(1, 2)))[0]
        if len(idx) == 0:
            continue
        if ("C" in vid and not parts["neck"]) or ("T" in vid and not parts["thorax"]) or ("L" in vid and not parts["abdomen"]):
            continue
        info[vid] = (int(idx.min()), int(idx.max() + 1))
    return info
