"""Body-composition aggregation on the device (host side).

Mirrors the numeric part of the reference's BCA pipeline for arrays already in SimpleITK order (z, y, x):
  BCA/tissue/subclassification.py:10-63   subclassify_tissues           -> boa_tissue_aggregate (fused)
  BCA/report/builder.py:397-444           slice-wise tissue volumes      -> per-slice counts from the same pass
  BCA/report/builder.py:44-112            AggregatableBodyPart.from_body_regions
  BCA/report/builder.py:163-307           aggregation groups + descriptive statistics + mean HU per tissue
  BCA/report/builder.py:520-598           create_json (numeric content of bca-measurements.json)
  BCA/body_regions/postprocess.py:8-40    largest-connected-component filters (boa_ccl26 + boa_ccl_filter_largest)
The per-voxel work (one pass over CT + regions + parts, 5 B/voxel) runs in libboa_hip.so; the host only turns
Z x 2 x 8 integer tables into the tiny DataFrames/dicts the reference produces (pandas `describe()` is called on
the same per-slice table, so the statistics follow pandas' arithmetic exactly).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np

from ._lib import check
from .device import Context, DeviceBuffer

REGION = dict(SUBCUTANEOUS_TISSUE=1, MUSCLE=2, ABDOMINAL_CAVITY=3, THORACIC_CAVITY=4, BONE=5, GLANDS=6, PERICARDIUM=7,
              BREAST_IMPLANT=8, MEDIASTINUM=9, BRAIN=10, NERVOUS_SYSTEM=11)  # BCA/body_regions/definition.py:4-15
TISSUES = [("MUSCLE", 1), ("BONE", 2), ("SAT", 3), ("VAT", 4), ("IMAT", 5), ("PAT", 6), ("EAT", 7)]  # tissue/definition.py
COLS = ["Bone", "Muscle", "TAT", "IMAT", "SAT", "VAT", "PAT", "EAT"]
_ROW = {"Mean": "mean", "StdDev": "std", "Minimum": "min", "25%": "q1", "Median": "q2", "75%": "q3", "Maximum": "max",
        "Total": "sum", "MeanHU": "mean_hu"}


def _tname(name):
    return name.capitalize() if name in ("BONE", "MUSCLE") else name


class DeviceVolume:
    """int16 CT / uint8 label volumes resident on the device, (z, y, x)."""

    def __init__(self, ctx: Context, arr: np.ndarray):
        assert arr.ndim == 3
        self.ctx = ctx
        self.shape = tuple(int(s) for s in arr.shape)
        self.dtype = arr.dtype
        self.buf = ctx.from_numpy(np.ascontiguousarray(arr))

    def free(self):
        self.buf.free()


def slice_axis_from_orientation(orientation) -> int:
    """subclassification.py:24-35: the array axis (of the (z,y,x) SimpleITK array) that gets kernel size 1."""
    if orientation is None:
        raise ValueError("Orientation must be set")
    orientation = tuple(orientation)
    if "I" in orientation:
        return orientation[::-1].index("I")
    if "S" in orientation:
        return orientation[::-1].index("S")
    raise ValueError(f"The orientation {orientation} does not contain neither I nor S.")


def median_filter_inplane(ctx: Context, ct: DeviceBuffer, shape, flat_axis: int = 0) -> DeviceBuffer:
    """scipy.ndimage.median_filter(image, size=[3,3,3] with 1 on `flat_axis`) on the device (int16)."""
    Z, Y, X = (int(s) for s in shape)
    out = ctx.alloc(Z * Y * X * 2)
    check(ctx.lib.boa_median3_inplane(ctx.h, ct.vp, Z, Y, X, int(flat_axis), out.vp), "boa_median3_inplane")
    return out


def tissue_aggregate(ctx: Context, ct: DeviceBuffer, regions: DeviceBuffer, parts: Optional[DeviceBuffer], shape,
                     want_tissues: bool = True, ct_rules: Optional[DeviceBuffer] = None):
    """One pass: tissues (uint8, optional), counts uint32 [Z,2,8], hu_sums int64 [Z,2,8] ([.,0,.] all voxels,
    [.,1,.] body_parts == TORSO).  ct_rules: the (median-filtered) CT the derivation rules look at, default ct."""
    Z, Y, X = (int(s) for s in shape)
    tis = ctx.alloc(Z * Y * X) if want_tissues else None
    cnt = ctx.alloc(Z * 16 * 4)
    sums = ctx.alloc(Z * 16 * 8)
    check(ctx.lib.boa_tissue_aggregate(ctx.h, ct.vp, ct_rules.vp if ct_rules else None, regions.vp,
                                       parts.vp if parts else None, tis.vp if tis else None, Z, Y, X, cnt.vp, sums.vp),
          "boa_tissue_aggregate")
    counts = cnt.download((Z, 2, 8), np.uint32)
    hu_sums = sums.download((Z, 2, 8), np.int64)
    cnt.free()
    sums.free()
    return tis, counts, hu_sums


HEATMAP_TISSUES = ("BONE", "MUSCLE", "IMAT", "SAT", "VAT", "PAT", "EAT")   # BCA/report/plots/heatmaps.py:38-46


def tissue_projections(ctx: Context, tissues: DeviceBuffer, regions: DeviceBuffer, shape, values):
    """create_tissue_heatmaps' reductions (BCA/report/plots/heatmaps.py:29-101) in one device pass:
    -> (coronal uint32 [T,Z,X] = per-tissue sum over y, sagittal uint32 [T,Z,Y] = sum over x, body silhouettes bool [Z,X],
    bool [Z,Y]) for the tissue label values `values`.  Normalisation to the maximum, colour map and resize are rendering."""
    Z, Y, X = (int(s) for s in shape)
    vals = np.ascontiguousarray(values, dtype=np.uint8)
    T = int(vals.size)
    d_c, d_s = ctx.alloc(T * Z * X * 4), ctx.alloc(T * Z * Y * 4)
    d_mc, d_ms = ctx.alloc(Z * X), ctx.alloc(Z * Y)
    try:
        check(ctx.lib.boa_tissue_projections(ctx.h, tissues.vp, regions.vp, Z, Y, X, vals.ctypes.data_as(C.c_void_p), T,
                                             d_c.vp, d_s.vp, d_mc.vp, d_ms.vp), "boa_tissue_projections")
        return (d_c.download((T, Z, X), np.uint32), d_s.download((T, Z, Y), np.uint32),
                d_mc.download((Z, X), np.uint8).astype(bool), d_ms.download((Z, Y), np.uint8).astype(bool))
    finally:
        for b in (d_c, d_s, d_mc, d_ms):
            b.free()


def slice_label_presence(ctx: Context, labels: DeviceBuffer, shape) -> np.ndarray:
    Z, Y, X = (int(s) for s in shape)
    d = ctx.alloc(Z * 256)
    check(ctx.lib.boa_slice_label_presence(ctx.h, labels.vp, Z, Y, X, d.vp), "boa_slice_label_presence")
    out = d.download((Z, 256), np.uint8).astype(bool)
    d.free()
    return out


def examined_body_part(present: np.ndarray, spacing_xyz, min_abdomen=200, min_neck=100, min_thorax=200) -> Dict[str, bool]:
    """AggregatableBodyPart.from_body_regions (builder.py:44-112) from the per-slice presence table."""
    thick = spacing_xyz[2]
    depth = present.shape[0]
    res = dict(abdomen=False, neck=False, thorax=False)
    abd = present[:, REGION["ABDOMINAL_CAVITY"]]
    sl = np.where(abd)[0]
    n_abd = sl.max() - sl.min() + 1 if sl.size else 0
    if n_abd * thick >= min_abdomen:
        res["abdomen"] = True
    med = np.where(present[:, REGION["MEDIASTINUM"]])[0]
    n_above = depth - med.max() if med.size else 0
    if n_above * thick >= min_neck:
        res["neck"] = True
    thx = present[:, REGION["THORACIC_CAVITY"]] | present[:, REGION["MEDIASTINUM"]] | present[:, REGION["PERICARDIUM"]]
    ts = np.where(thx)[0]
    n_thx = ts.max() - ts.min() + 1 if ts.size else 0
    if np.logical_and(abd, thx).any() and n_thx * thick >= min_thorax:
        res["thorax"] = True
    return res


def aggregation_groups(present: np.ndarray, depth: int, parts: Dict[str, bool], vertebrae=None):
    """Slice ranges of generate_aggregated_measurements (builder.py:170-216)."""
    def rng(col):
        s = np.where(col)[0]
        return int(s.min()), int(s.max()) + 1

    groups = [("Whole Scan", 0, depth)]
    if parts["abdomen"]:
        groups.append(("Abdominal Cavity", *rng(present[:, REGION["ABDOMINAL_CAVITY"]])))
    if parts["thorax"]:
        thx = present[:, REGION["THORACIC_CAVITY"]] | present[:, REGION["MEDIASTINUM"]] | present[:, REGION["PERICARDIUM"]]
        groups.append(("Thoracic Cavity", *rng(thx)))
        groups.append(("Mediastinum", *rng(present[:, REGION["MEDIASTINUM"]])))
        groups.append(("Pericardium", *rng(present[:, REGION["PERICARDIUM"]])))
    if parts["abdomen"] and parts["thorax"]:
        if groups[1][0] != "Abdominal Cavity":
            raise ValueError("Something went wrong for Abdominal Cavity")
        if groups[2][0] != "Thoracic Cavity":
            raise ValueError("Something went wrong for Thoracic Cavity")
        groups.insert(1, ("Ventral Cavity", groups[1][1], groups[2][2]))
    if vertebrae:
        for name, g in vertebrae.items():
            groups.append((name, g[0], g[1]))
    return groups


def _slicewise(counts_a: np.ndarray, ml_per_voxel: float):
    import pandas as pd
    data = {_tname(n): counts_a[:, v].astype(np.int64) * ml_per_voxel for n, v in TISSUES}
    df = pd.DataFrame(data)
    df["TAT"] = df.SAT + df.VAT + df.IMAT + df.PAT + df.EAT
    df["slice_idx"] = range(len(df))
    return df[["slice_idx"] + COLS]


_STAT_ROWS = ["Mean", "StdDev", "Minimum", "25%", "Median", "75%", "Maximum", "Total", "MeanHU"]


def _descriptive(x_all: np.ndarray, cols, counts_a, sums_a, lo, hi) -> dict:
    """_descriptive_statistics_from_measurements (builder.py:263-307) for slices [lo, hi), returned the way run_pipeline
    stores it (`m.rename(...).to_dict()`: {column: {statistic: float | None}}): pandas `describe()` of the slice-wise
    volumes (mean, std with ddof 1, min, linear-interpolated quartiles, max) + Total + MeanHU.  Evaluated with the numpy
    operations pandas itself dispatches to (sum / count, sqrt(sum((x - mean)^2) / (n - 1)), numpy's linear percentile) and
    written into plain dicts: the same numbers as `DataFrame.describe()` + `.loc[]` + `.where()` + `.to_dict()` without
    their frame bookkeeping (~8 ms per group and variant, 0.4 s per volume, with the GPU idle behind it).
    `x_all` [slices, len(cols)] float64 = the slice-wise table without its index column."""
    # one contiguous row per tissue: numpy then reduces each row with the pairwise summation pandas gets on a column
    x = np.ascontiguousarray(x_all[max(lo, 0):max(hi, 0)].T)      # [tissues, slices]; rows with lo <= slice_idx < hi
    n = x.shape[1]
    out = np.full((len(_STAT_ROWS), len(cols)), np.nan)
    if n:
        mean = x.sum(axis=1) / n
        out[0] = mean
        if n > 1:
            d = mean[:, None] - x
            out[1] = np.sqrt((d * d).sum(axis=1) / (n - 1))
        out[2] = x.min(axis=1)
        out[3:6] = np.percentile(x, [25, 50, 75], axis=1)
        out[6] = x.max(axis=1)
    out[7] = x.sum(axis=1)
    c = counts_a[lo:hi].astype(np.int64).sum(axis=0)
    s = sums_a[lo:hi].sum(axis=0)
    mean_hu = {_tname(nme): ((float(s[v]) / float(c[v])) if c[v] else None) for nme, v in TISSUES}
    adip = [5, 3, 4, 6, 7]
    ca, sa = int(c[adip].sum()), int(s[adip].sum())
    mean_hu["TAT"] = (float(sa) / float(ca)) if ca else None
    keys = [_ROW[r] for r in _STAT_ROWS[:-1]]
    res = {}
    for ci, col in enumerate(cols):
        vals = out[:len(keys), ci].tolist()
        m = {k: (None if v != v else v) for k, v in zip(keys, vals)}
        m[_ROW["MeanHU"]] = mean_hu[col]
        res[col.lower()] = m
    return res


def bca_measurements_from_tables(counts, hu_sums, present, spacing_xyz, vertebrae=None, body_parts_override=None) -> dict:
    ml = np.prod(spacing_xyz) / 1000.0
    depth = counts.shape[0]
    df = _slicewise(counts[:, 0], ml)
    d2 = _slicewise(counts[:, 1], ml)
    # run_pipeline: `builder.examined_body_part` is the detected flag set unless examined_body_region is given (:150-165)
    parts = dict(body_parts_override) if body_parts_override is not None else examined_body_part(present, spacing_xyz)
    groups = aggregation_groups(present, depth, parts, vertebrae)
    agg = {}
    x1 = df[COLS].to_numpy(dtype=np.float64)          # (slice_idx is 0 .. depth-1: rows [lo, hi) are the group's slices)
    x2 = d2[COLS].to_numpy(dtype=np.float64)
    for name, lo, hi in groups:
        key = name.lower().replace(" ", "_").replace("-", "_")
        agg[key] = {
            "num_slices": int(hi - lo), "min_slice_idx": int(lo), "max_slice_idx": int(hi),
            "measurements": _descriptive(x1, COLS, counts[:, 0], hu_sums[:, 0], lo, hi),
            "measurements_no_extremities": _descriptive(x2, COLS, counts[:, 1], hu_sums[:, 1], lo, hi),
        }

    def recs(d):
        return d.rename(columns={c: c.lower() for c in d.columns}).drop("slice_idx", axis=1).astype(float).to_dict("records")

    return {"slices": recs(df), "slices_no_extremities": recs(d2), "aggregated": agg, "body_parts": parts}


def bca_measurements(ctx: Context, ct: np.ndarray, regions: np.ndarray, parts: np.ndarray, spacing_xyz,
                     vertebrae=None, return_tissues: bool = False, median_filtering: bool = False, orientation="LPS",
                     body_parts_override=None):
    """CT (z,y,x) int16 + body_regions + body_parts -> bca-measurements dict (+ tissues array).
    median_filtering: subclassify on the 3x3 in-plane median of the CT (run_pipeline(median_filtering=True))."""
    if ct.shape != regions.shape or ct.shape != parts.shape:
        raise ValueError("image, body_regions and body_parts must have the same shape")
    shape = ct.shape
    d_ct = ctx.from_numpy(np.ascontiguousarray(ct, dtype=np.int16))
    d_rg = ctx.from_numpy(np.ascontiguousarray(regions, dtype=np.uint8))
    d_pt = ctx.from_numpy(np.ascontiguousarray(parts, dtype=np.uint8))
    try:
        out, tis = bca_measurements_device(ctx, d_ct, d_rg, d_pt, shape, spacing_xyz, vertebrae, return_tissues,
                                           median_filtering, orientation, body_parts_override)
        if return_tissues:
            t = tis.download(shape, np.uint8)
            tis.free()
            return out, t
        return out
    finally:
        for b in (d_ct, d_rg, d_pt):
            b.free()


def bca_measurements_device(ctx: Context, d_ct: DeviceBuffer, d_rg: DeviceBuffer, d_pt: DeviceBuffer, shape, spacing_xyz,
                            vertebrae=None, return_tissues: bool = False, median_filtering: bool = False,
                            orientation="LPS", body_parts_override=None):
    """`bca_measurements` on resident (z,y,x) buffers (int16 CT, uint8 regions / parts) -> (dict, tissues buffer | None)."""
    d_med = None
    try:
        if median_filtering:
            d_med = median_filter_inplane(ctx, d_ct, shape, slice_axis_from_orientation(orientation))
        tis, counts, hu_sums = tissue_aggregate(ctx, d_ct, d_rg, d_pt, shape, want_tissues=return_tissues, ct_rules=d_med)
        present = slice_label_presence(ctx, d_rg, shape)
        out = bca_measurements_from_tables(counts, hu_sums, present, spacing_xyz, vertebrae, body_parts_override)
        return out, tis
    finally:
        if d_med is not None:
            d_med.free()


def create_vertebrae_info(ctx: Context, total_zyx: Optional[np.ndarray], class_map_total: Dict[int, str], parts: Dict[str, bool],
                          d_total: Optional[DeviceBuffer] = None, shape=None) -> Dict[str, Tuple[int, int]]:
    """BCA/commands.py:24-45: slice range (min, max+1) of every vertebra label that is present and whose body part
    (C -> neck, T -> thorax, L -> abdomen) was detected.  One presence pass on the device."""
    own = d_total is None
    if own:
        d_total = ctx.from_numpy(np.ascontiguousarray(total_zyx, dtype=np.uint8))
    try:
        present = slice_label_presence(ctx, d_total, total_zyx.shape if shape is None else shape)
    finally:
        if own:
            d_total.free()
    info = {}
    for label, name in class_map_total.items():
        if not name.startswith("vertebrae_"):
            continue
        vid = name[len("vertebrae_"):]
        idx = np.where(present[:, int(label)])[0]
        if len(idx) == 0:
            continue
        if ("C" in vid and not parts["neck"]) or ("T" in vid and not parts["thorax"]) or \
                ("L" in vid and not parts["abdomen"]):
            continue
        info[vid] = (int(idx.min()), int(idx.max() + 1))
    return info


# ---- connected-component post-processing of the body-region segmentation --------------------------------
def postprocess_region_segmentation(ctx: Context, seg: np.ndarray) -> np.ndarray:
    """BCA/body_regions/postprocess.py:18-40 on the device: for the masks {seg > 0}, {thoracic, mediastinum,
    pericardium}, {pericardium}, {abdominal cavity}: all 26-connected components except the largest -> 255."""
    d_seg = ctx.from_numpy(np.ascontiguousarray(seg, dtype=np.uint8))
    try:
        postprocess_region_segmentation_device(ctx, d_seg, seg.shape)
        return d_seg.download(seg.shape, np.uint8)
    finally:
        d_seg.free()


def _morph_bytes() -> bool:
    import os
    return os.environ.get("BOA_MORPH_BYTES", "") not in ("", "0")


def _lut(values=None, positive=False):
    """256-entry membership table for boa_bits_select: bit 0 set for the label values of the mask."""
    lut = np.zeros(256, np.uint8)
    if positive:
        lut[1:] = 1
    else:
        for v in values:
            lut[int(v)] = 1
    return lut


def postprocess_region_segmentation_device(ctx: Context, d_seg: DeviceBuffer, shape) -> None:
    """In place on a resident uint8 (z,y,x) label buffer.  The four filters of BCA/body_regions/postprocess.py:18-40 depend on each
    other through `seg_data` (a voxel set to 255 leaves the later masks), so they run one after the other -- each on a bit mask
    (csrc/ccl_bits.hip): select (1 B per voxel read), component labelling on tiles (uniform tiles cost nothing), one byte written per
    removed voxel."""
    if _morph_bytes():
        return _postprocess_region_segmentation_device_bytes(ctx, d_seg, shape)
    Z, Y, X = (int(v) for v in shape)
    d_bits = ctx.alloc(int(ctx.lib.boa_bits_words(Z, Y, X)) * 4)

    def run(lut):
        check(ctx.lib.boa_bits_select(ctx.h, d_seg.vp, Z, Y, X, lut.ctypes.data_as(C.c_void_p), 1, d_bits.vp), "boa_bits_select")
        check(ctx.lib.boa_bits_filter_largest(ctx.h, d_bits.vp, Z, Y, X, d_seg.vp, 255), "boa_bits_filter_largest")

    try:
        run(_lut(positive=True))
        run(_lut((REGION["THORACIC_CAVITY"], REGION["MEDIASTINUM"], REGION["PERICARDIUM"])))
        run(_lut((REGION["PERICARDIUM"],)))
        run(_lut((REGION["ABDOMINAL_CAVITY"],)))
    finally:
        d_bits.free()


def _postprocess_region_segmentation_device_bytes(ctx: Context, d_seg: DeviceBuffer, shape) -> None:
    """The byte-mask form (boa_label_select / boa_ccl26 / boa_ccl_filter_largest per mask): the cross-check of the bit-mask path
    ($BOA_MORPH_BYTES=1) and what the z-slab sharded variant below is built from."""
    shape = tuple(int(v) for v in shape)
    n = int(np.prod(shape))
    d_mask = ctx.alloc(n)
    d_roots = ctx.alloc(n * 4)
    d_sizes = ctx.alloc(n * 4)

    def run(mode, vals):
        v = (C.c_int * 3)(*vals)
        check(ctx.lib.boa_label_select(ctx.h, d_seg.vp, n, mode, v, d_mask.vp))
        # (no component count back to the host: boa_ccl_filter_largest is a no-op for <= 1 component, and the chain stays
        #  free of host round trips)
        check(ctx.lib.boa_ccl26(ctx.h, d_mask.vp, shape[0], shape[1], shape[2], d_roots.vp, d_sizes.vp, None))
        check(ctx.lib.boa_ccl_filter_largest(ctx.h, d_roots.vp, d_sizes.vp, n, d_seg.vp, 255))

    try:
        run(1, (0, 0, 0))
        run(2, (REGION["THORACIC_CAVITY"], REGION["MEDIASTINUM"], REGION["PERICARDIUM"]))
        run(0, (REGION["PERICARDIUM"], 0, 0))
        run(0, (REGION["ABDOMINAL_CAVITY"], 0, 0))
    finally:
        for b in (d_mask, d_roots, d_sizes):
            b.free()


def postprocess_region_segmentation_device_sharded(ctx: Context, agg, d_seg: DeviceBuffer, shape) -> None:
    """The same CC filters with the (z,y,x) label volume cut into z-slabs, one per rank (SURVEY 8e aggregation stages;
    boa_hip/agg_shard.py): `agg` = (agg_shard.AggComm, tile_shard.ShardComm).  Every rank passes the same full volume and
    ends with the same cleaned volume (the slabs are exchanged as a sum over disjoint supports)."""
    from . import agg_shard as ag
    from . import tile_shard as ts
    from .device import BufferView
    comm, label_comm = agg
    Z, Y, X = (int(v) for v in shape)
    n = Z * Y * X
    z0, z1 = ag.slab_bounds(Z, comm.world)[comm.rank]
    n_slab = (z1 - z0) * Y * X
    slab = BufferView(d_seg, z0 * Y * X, n_slab)
    d_mask = ctx.alloc(max(n_slab, 1))
    eng = ag.HipAggEngine(ctx, (z1 - z0, Y, X))

    def select(seg, mode, vals):
        if n_slab:
            check(ctx.lib.boa_label_select(ctx.h, seg.vp, n_slab, mode, (C.c_int * 3)(*vals), d_mask.vp))
        return d_mask

    try:
        ag.postprocess_region_segmentation_sharded(comm, eng, select, slab, z0, REGION)
        BufferView(d_seg, 0, z0 * Y * X).zero()
        BufferView(d_seg, z1 * Y * X, n - z1 * Y * X).zero()
        ts.all_reduce_labels(ctx, label_comm, d_seg, n)
    finally:
        eng.close()
        d_mask.free()


def bca_measurements_device_sharded(ctx: Context, agg, d_ct: DeviceBuffer, d_rg: DeviceBuffer, d_pt: DeviceBuffer, shape, spacing_xyz,
                                    vertebrae=None, orientation="LPS", body_parts_override=None):
    """`bca_measurements_device(return_tissues=True)` with the tissue pass and the presence table computed per z-slab and the
    per-slice tables gathered (agg_shard.gather_slice_tables); returns (dict, full tissues buffer), identical on all ranks."""
    from . import agg_shard as ag
    from . import tile_shard as ts
    from .device import BufferView
    comm, label_comm = agg
    Z, Y, X = (int(v) for v in shape)
    n = Z * Y * X
    z0, z1 = ag.slab_bounds(Z, comm.world)[comm.rank]
    Zl, pl = z1 - z0, Y * X
    tis = ctx.zeros(n)
    try:
        if Zl:
            cnt, sums = ctx.alloc(Zl * 16 * 4), ctx.alloc(Zl * 16 * 8)
            check(ctx.lib.boa_tissue_aggregate(ctx.h, BufferView(d_ct, z0 * pl * 2, Zl * pl * 2).vp, None, BufferView(d_rg, z0 * pl, Zl * pl).vp,
                                               BufferView(d_pt, z0 * pl, Zl * pl).vp, BufferView(tis, z0 * pl, Zl * pl).vp, Zl, Y, X,
                                               cnt.vp, sums.vp), "boa_tissue_aggregate")
            counts, hu_sums = cnt.download((Zl, 2, 8), np.uint32), sums.download((Zl, 2, 8), np.int64)
            cnt.free()
            sums.free()
            present = slice_label_presence(ctx, BufferView(d_rg, z0 * pl, Zl * pl), (Zl, Y, X))
        else:
            counts, hu_sums, present = np.zeros((0, 2, 8), np.uint32), np.zeros((0, 2, 8), np.int64), np.zeros((0, 256), bool)
        counts, hu_sums, present = ag.gather_slice_tables(comm, counts, hu_sums, present)
        ts.all_reduce_labels(ctx, label_comm, tis, n)
        return bca_measurements_from_tables(counts, hu_sums, present, spacing_xyz, vertebrae, body_parts_override), tis
    except Exception:
        tis.free()
        raise


def postprocess_part_segmentation(ctx: Context, seg: np.ndarray, threshold: int = 3000) -> np.ndarray:
    """BCA/body_parts/postprocess.py:7-60 (`remove_small_labeled_objects`) on the device.  Per label (ascending):
    slice-wise external-contour fill (boa_fill_holes_2d), remove 26-connected objects with <= threshold-1 voxels,
    remove 26-connected holes with <= threshold-1 voxels, `out[filled] = label`."""
    seg = np.ascontiguousarray(seg, dtype=np.uint8)
    d_seg = ctx.from_numpy(seg)
    try:
        d_out = postprocess_part_segmentation_device(ctx, d_seg, seg.shape, threshold)
        try:
            return d_out.download(seg.shape, np.uint8)
        finally:
            d_out.free()
    finally:
        d_seg.free()


def _fill_holes_2d_cropped(ctx: Context, d_mask: DeviceBuffer, shape, d_i32: DeviceBuffer, d_s1: DeviceBuffer, d_s2: DeviceBuffer,
                           d_out: DeviceBuffer):
    """boa_fill_holes_2d restricted to the mask's bounding box (+ 1 pixel in-plane where the slice is larger): a background
    pixel reaches the slice border through background iff it reaches the border of that box (everything outside the box is
    background, and where the box has no margin its border IS the slice border), so the result is identical while the
    union-find only sees the box -- body-part masks fill a fraction of the volume.  d_s1 / d_s2: uint8 scratch of the
    volume's size, d_i32: int32 scratch; d_mask is left unchanged."""
    from .devarray import DevArray
    Z, Y, X = (int(v) for v in shape)
    bb = (C.c_int * 6)()
    check(ctx.lib.boa_nonzero_bbox(ctx.h, d_mask.vp, 0, (C.c_int * 3)(Z, Y, X), bb), "boa_nonzero_bbox")
    box = [[bb[0], bb[1]], [max(bb[2] - 1, 0), min(bb[3] + 1, Y)], [max(bb[4] - 1, 0), min(bb[5] + 1, X)]]
    cz, cy, cx = (b[1] - b[0] for b in box)
    if cz * cy * cx > 0.7 * Z * Y * X:
        check(ctx.lib.boa_fill_holes_2d(ctx.h, d_mask.vp, Z, Y, X, d_i32.vp, d_s1.vp, d_out.vp), "boa_fill_holes_2d")
        return
    full = DevArray(ctx, d_mask, (Z, Y, X), np.uint8)
    cin = DevArray(ctx, d_s1, (cz, cy, cx), np.uint8)
    full.box(box).copy_to(cin)
    # (d_out doubles as the uint8 scratch of the cropped call; its box is rewritten below, the rest cleared)
    check(ctx.lib.boa_fill_holes_2d(ctx.h, d_s1.vp, cz, cy, cx, d_i32.vp, d_out.vp, d_s2.vp), "boa_fill_holes_2d")
    d_out.zero()
    DevArray(ctx, d_s2, (cz, cy, cx), np.uint8).copy_to(DevArray(ctx, d_out, (Z, Y, X), np.uint8).box(box))


def postprocess_part_segmentation_device(ctx: Context, d_seg: DeviceBuffer, shape, threshold: int = 3000, labels=None) -> DeviceBuffer:
    """Resident uint8 (z,y,x) labels -> new resident buffer with the cleaned labels.

    The per-label passes of remove_small_labeled_objects (BCA/body_parts/postprocess.py:27-50) all start from the ORIGINAL volume
    (`label_mask = mask == label`) and only meet at `out[filled] = label`, so the labels are one batch of bit masks
    (csrc/ccl_bits.hip): one select pass, one slice-wise contour fill launch over (label, slice), the small-object and the
    small-hole filter as two batched component labellings, one assign pass in which the largest label that holds a voxel wins (the
    reference's ascending overwrite order).  `labels`: the label values to process (absent ones are harmless: an empty mask stays
    empty and its complement is one large component); None asks the device which labels occur (one small synchronous read)."""
    shape = tuple(int(v) for v in shape)
    Z, Y, X = shape
    if _morph_bytes() or not ctx.lib.boa_bits_fill_supported(Y, X):
        return _postprocess_part_segmentation_device_bytes(ctx, d_seg, shape, threshold)
    n = Z * Y * X
    # An ABSENT label's inverted pass sees the whole volume as one hole: harmless while the volume has >= `threshold` voxels (the hole
    # stays), but a smaller volume would be filled with a label that does not occur -- the reference only visits np.unique(mask).
    if labels is None or n < threshold:
        labels = np.flatnonzero(slice_label_presence(ctx, d_seg, shape).any(axis=0))
    labels = sorted(int(v) for v in labels if int(v) > 0)
    words = int(ctx.lib.boa_bits_words(Z, Y, X))
    try:
        return _postprocess_part_bits(ctx, d_seg, shape, threshold, labels, words)
    except (MemoryError, ValueError) as e:
        # the bit path holds ~2 B / voxel + 12 KiB / tile per mask for 8 masks at once and indexes voxels / tiles with 31 bits: when it
        # cannot run (BOA_ENOMEM after the allocator's own trim + retry, BOA_EINVAL from a size limit) the byte path does the same
        # filters label by label in 13 B / voxel
        import logging
        logging.getLogger(__name__).warning("bit-mask post-processing unavailable (%s): falling back to the byte-mask path", e)
        ctx.lib.boa_trim(ctx.h)
        return _postprocess_part_segmentation_device_bytes(ctx, d_seg, shape, threshold)


def _postprocess_part_bits(ctx: Context, d_seg: DeviceBuffer, shape, threshold: int, labels, words: int) -> DeviceBuffer:
    Z, Y, X = shape
    n = Z * Y * X
    d_out = ctx.zeros(n)
    try:
        for b0 in range(0, len(labels), 8):          # batches of 8 masks (one byte of membership bits per label value), ascending
            batch = labels[b0:b0 + 8]
            m = len(batch)
            lut = np.zeros(256, np.uint8)
            for j, v in enumerate(batch):
                lut[v] = 1 << j
            d_a, d_b = ctx.alloc(words * 4 * m), ctx.alloc(words * 4 * m)
            try:
                check(ctx.lib.boa_bits_select(ctx.h, d_seg.vp, Z, Y, X, lut.ctypes.data_as(C.c_void_p), m, d_a.vp), "boa_bits_select")
                check(ctx.lib.boa_bits_fill_holes_2d(ctx.h, d_a.vp, Z, Y, X, m, d_b.vp), "boa_bits_fill_holes_2d")
                check(ctx.lib.boa_bits_remove_small(ctx.h, d_b.vp, Z, Y, X, m, threshold - 1, 0), "boa_bits_remove_small")   # small objects
                check(ctx.lib.boa_bits_remove_small(ctx.h, d_b.vp, Z, Y, X, m, threshold - 1, 1), "boa_bits_remove_small")   # small holes
                lab = np.asarray(batch, np.uint8)
                check(ctx.lib.boa_bits_assign_labels(ctx.h, d_b.vp, Z, Y, X, m, lab.ctypes.data_as(C.c_void_p), d_out.vp), "boa_bits_assign_labels")
            finally:
                d_a.free()
                d_b.free()
        return d_out
    except Exception:
        d_out.free()
        raise


def _postprocess_part_segmentation_device_bytes(ctx: Context, d_seg: DeviceBuffer, shape, threshold: int = 3000) -> DeviceBuffer:
    """The byte-mask form, one label after the other ($BOA_MORPH_BYTES=1, and slices too large for the LDS flood of the bit path)."""
    shape = tuple(int(v) for v in shape)
    Z, Y, X = shape
    n = Z * Y * X
    d_out = ctx.zeros(n)
    d_mask, d_fill, d_tmp, d_box = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
    d_roots, d_sizes = ctx.alloc(n * 4), ctx.alloc(n * 4)
    try:
        labels = np.flatnonzero(slice_label_presence(ctx, d_seg, shape).any(axis=0))
        for label in labels[labels > 0]:
            v = (C.c_int * 3)(int(label), 0, 0)
            check(ctx.lib.boa_label_select(ctx.h, d_seg.vp, n, 0, v, d_mask.vp))
            _fill_holes_2d_cropped(ctx, d_mask, shape, d_roots, d_tmp, d_box, d_fill)
            # small foreground objects
            check(ctx.lib.boa_ccl26(ctx.h, d_fill.vp, Z, Y, X, d_roots.vp, d_sizes.vp, None))
            check(ctx.lib.boa_ccl_remove_small(ctx.h, d_roots.vp, d_sizes.vp, n, threshold - 1, d_fill.vp))
            # small holes: the same on the inverted mask
            zero = (C.c_int * 3)(0, 0, 0)
            check(ctx.lib.boa_label_select(ctx.h, d_fill.vp, n, 0, zero, d_mask.vp))       # d_mask = ~filled
            check(ctx.lib.boa_ccl26(ctx.h, d_mask.vp, Z, Y, X, d_roots.vp, d_sizes.vp, None))
            check(ctx.lib.boa_ccl_remove_small(ctx.h, d_roots.vp, d_sizes.vp, n, threshold - 1, d_mask.vp))
            check(ctx.lib.boa_mask_assign(ctx.h, d_mask.vp, n, 1, int(label), d_out.vp))   # out[~d_mask] = label
        return d_out
    except Exception:
        d_out.free()
        raise
    finally:
        for b in (d_mask, d_fill, d_tmp, d_box, d_roots, d_sizes):
            b.free()
