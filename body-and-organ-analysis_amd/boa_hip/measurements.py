"""Per-label HU measurements from ONE device histogram pass (host side).

Mirrors BOA/compute/measurements.py for arrays in SimpleITK order (z, y, x):
  metrics_for_region (:74-123), metrics_for_each_region (:203-241), autochthon_reference (:42-58),
  erode_region (:61-71), compute_lung_measurement / ct_pfav (:126-200), compute_measurements (:244-343, the
  `total` branch incl. the CNR-adjusted regions of CNR_ADJUSTED_REGIONS).
The reference makes ~125 full-volume passes (one boolean mask, one fancy-index gather and seven reductions per
label).  Here the device builds hist[label][HU] over the whole int16 range in a single 3 B/voxel pass
(boa_label_hu_histogram); count / mean / std / min / max / median / percentiles of every label, of label unions
(autochthon left+right, lung lobes) and of HU-window sub-masks (pulmonary fat) follow exactly from the integer
counts on the host.  Only the eroded masks of the CNR adjustment need extra passes (mask -> separable erosion ->
histogram).
Exactness: counts, min, max, median, percentiles (numpy 'linear' interpolation restated operation for operation),
mean (exact integer sum / n) are bit-identical to numpy on the same integer data; std differs by summation order
only (<= 1e-12 relative).
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, Iterable, Optional, Tuple

import numpy as np

from ._lib import check
from .device import Context, DeviceBuffer

ADIPOSE_TISSUE = (-200, -40)  # BOA/compute/measurements.py:22
HU_MIN, NBINS = -32768, 65536
CNR_ADJUSTED_REGIONS = {"total": {"aorta", "autochthon_left", "autochthon_right"},
                        "heartchambers_highres": {"pulmonary_artery"}}
LUNG_MASKS = ["lung_upper_lobe_left", "lung_lower_lobe_left", "lung_upper_lobe_right", "lung_middle_lobe_right",
              "lung_lower_lobe_right"]


# ---- statistics from an integer histogram ---------------------------------------------------------------
def _lerp(a: float, b: float, t: float) -> float:
    """numpy.lib._function_base_impl._lerp for scalars."""
    d = b - a
    r = a + d * t
    if t >= 0.5:
        r = b - d * (1 - t)
    return r


def occupied_window(hist: np.ndarray):
    """[b0, b1): the bin range outside which every row of `hist` [rows, nbins] is empty (one OR-reduction over the rows
    instead of a 65 536-bin scan per label: CT values occupy a few thousand of the int16 bins)."""
    h = np.ascontiguousarray(hist)
    cols = np.bitwise_or.reduce(h.view(np.uint64) if (h.dtype.itemsize == 4 and h.shape[1] % 2 == 0) else h, axis=0)
    nz = np.flatnonzero(cols)
    if nz.size == 0:
        return 0, 0
    k = 2 if cols.shape[0] != h.shape[1] else 1
    return int(nz[0]) * k, (int(nz[-1]) + 1) * k


def stats_from_hist(h: np.ndarray, hu_min: int = HU_MIN) -> Optional[Dict[str, float]]:
    """h: uint32/int64 [nbins] counts of value hu_min + bin.  Returns None for an empty region."""
    nz = np.nonzero(h)[0]
    if nz.size == 0:
        return None
    cnt = h[nz].astype(np.int64)
    vals = nz.astype(np.int64) + hu_min
    n = int(cnt.sum())
    total = int((vals * cnt).sum())
    mean = float(np.float64(total) / np.float64(n))
    dev = vals.astype(np.float64) - mean
    var = float((dev * dev * cnt).sum() / n)
    cum = np.cumsum(cnt)

    def kth(k: int) -> float:
        return float(vals[np.searchsorted(cum, k, side="right")])

    def percentile(p: float) -> float:
        virt = (n - 1) * (p / 100.0)
        prev = int(np.floor(virt))
        nxt = min(prev + 1, n - 1)
        return float(_lerp(kth(prev), kth(nxt), virt - prev))

    if n % 2:
        median = kth(n // 2)
    else:
        median = (kth(n // 2 - 1) + kth(n // 2)) / 2.0
    return {"n": n, "mean": mean, "std": float(np.sqrt(var)), "min": float(vals[0]), "max": float(vals[-1]),
            "median": median, "p25": percentile(25), "p75": percentile(75)}


def _metrics(st: Optional[dict], ml_per_voxel: float, auto_mean, auto_std, cnr_none: bool = False) -> Dict[str, Any]:
    """Dict layout of metrics_for_region (:74-123)."""
    if st is None:
        return {"present": False}
    out: Dict[str, Any] = {"present": True, "volume_ml": st["n"] * ml_per_voxel, "mean_hu": st["mean"], "std_hu": st["std"],
                           "min_hu": st["min"], "median_hu": st["median"], "max_hu": st["max"],
                           "25th_percentile_hu": st["p25"], "75th_percentile_hu": st["p75"]}
    if auto_mean is not None and auto_std is not None and not cnr_none:
        # numpy scalar arithmetic like the reference (`np.mean(hu_region) - autochthon_mean) / autochthon_std`, :117-119): a
        # constant-HU autochthon (std 0) gives inf / nan with a RuntimeWarning there, not a ZeroDivisionError
        with np.errstate(divide="ignore", invalid="ignore"):
            out["cnr"] = (np.float64(st["mean"]) - auto_mean) / np.float64(auto_std)
    else:
        out["cnr"] = None
    return out


# ---- device passes --------------------------------------------------------------------------------------
def _label_hu_histogram_local(ctx: Context, d_ct: DeviceBuffer, d_labels: DeviceBuffer, n: int,
                              d_mask: Optional[DeviceBuffer] = None) -> np.ndarray:
    d_hist = ctx.alloc(256 * NBINS * 4)
    check(ctx.lib.boa_label_hu_histogram(ctx.h, d_ct.vp, d_labels.vp, d_mask.vp if d_mask else None, n, HU_MIN, NBINS,
                                         d_hist.vp), "boa_label_hu_histogram")
    h = d_hist.download((256, NBINS), np.uint32)
    d_hist.free()
    return h


label_hu_histogram = _label_hu_histogram_local


def _lut(labels: Iterable[int]) -> np.ndarray:
    lut = np.zeros(256, dtype=np.uint8)
    for l in labels:
        lut[int(l)] = 1
    return lut


def label_hu_mask(ctx: Context, d_ct, d_labels, labels, mode: int, n: int, out: DeviceBuffer, window=ADIPOSE_TISSUE):
    """mode 0: label in set; 1: and lo <= HU <= hi; 2: and (HU < lo or HU > hi)."""
    lut = _lut(labels)
    check(ctx.lib.boa_label_hu_mask(ctx.h, d_ct.vp, d_labels.vp, lut.ctypes.data_as(C.c_void_p), mode, window[0], window[1],
                                    n, out.vp), "boa_label_hu_mask")


def binary_erode(ctx: Context, d_mask: DeviceBuffer, d_out: DeviceBuffer, d_tmp: DeviceBuffer, shape, kernel_value: int = 6):
    check(ctx.lib.boa_binary_erode(ctx.h, d_mask.vp, d_out.vp, d_tmp.vp, int(shape[0]), int(shape[1]), int(shape[2]),
                                   kernel_value), "boa_binary_erode")


def erode_region(ctx: Context, mask: np.ndarray, kernel_value: int = 6) -> np.ndarray:
    """erode_region (:61-71) for a host boolean array."""
    n = mask.size
    d_m = ctx.from_numpy(np.ascontiguousarray(mask, dtype=np.uint8))
    d_o, d_t = ctx.alloc(n), ctx.alloc(n)
    try:
        binary_erode(ctx, d_m, d_o, d_t, mask.shape, kernel_value)
        return d_o.download(mask.shape, np.uint8).astype(bool)
    finally:
        for b in (d_m, d_o, d_t):
            b.free()


def _masked_stats_local(ctx, d_ct, d_mask, n):
    """stats of ct[mask != 0]: histogram with the mask itself as the (0/1) label volume; only row 1 comes back."""
    from .device import BufferView
    d_hist = ctx.alloc(256 * NBINS * 4)
    try:
        check(ctx.lib.boa_label_hu_histogram(ctx.h, d_ct.vp, d_mask.vp, None, n, HU_MIN, NBINS, d_hist.vp), "boa_label_hu_histogram")
        row = BufferView(d_hist, NBINS * 4, NBINS * 4).download((NBINS,), np.uint32)
    finally:
        d_hist.free()
    return stats_from_hist(row)


_masked_stats = _masked_stats_local


def metrics_for_each_region(ctx: Context, ct: np.ndarray, region_data: np.ndarray, label_map: Dict[str, int],
                            autochthon_mean, autochthon_std, img_spacing) -> Dict[str, Any]:
    """metrics_for_each_region (:203-241), cnr_adjustment=False, from one histogram pass."""
    n = ct.size
    d_ct = ctx.from_numpy(np.ascontiguousarray(ct, dtype=np.int16))
    d_lab = ctx.from_numpy(np.ascontiguousarray(region_data, dtype=np.uint8))
    try:
        hist = label_hu_histogram(ctx, d_ct, d_lab, n)
    finally:
        d_ct.free()
        d_lab.free()
    return _metrics_from_hist(hist, label_map, autochthon_mean, autochthon_std, img_spacing)


def _metrics_from_hist(hist, label_map, am, asd, spacing):
    ml = np.prod(spacing) / 1000.0
    res = {}
    b0, b1 = occupied_window(hist)
    hw, hu0 = hist[:, b0:b1], HU_MIN + b0          # same statistics: bin b of the window is HU hu0 + b
    for region, label in label_map.items():
        res[region] = _metrics(stats_from_hist(hw[label], hu0) if 0 < label < 256 else None, ml, am, asd)
    if "autochthon_left" in label_map and "autochthon_right" in label_map:
        h = hw[label_map["autochthon_left"]].astype(np.int64) + hw[label_map["autochthon_right"]]
        res["autochthon"] = _metrics(stats_from_hist(h, hu0), ml, am, asd)
    return res


def cnr_adjusted_region_metrics(ctx: Context, d_ct: DeviceBuffer, d_lab: DeviceBuffer, shape, label_map: Dict[str, int],
                                regions: Iterable[str], hist: np.ndarray, am, asd, spacing) -> Dict[str, Any]:
    """`metrics_for_each_region(..., cnr_adjustment=True)` for the `regions` of a model other than `total`
    (BOA/compute/measurements.py:318-341; e.g. heartchambers_highres / pulmonary_artery): the region's mask (autochthon names:
    minus fat) is eroded with the 6^3 kernel before the statistics (metrics_for_region, :85-93).  `hist`: the model's
    per-label histogram (tells which labels are present)."""
    shape = tuple(int(v) for v in shape)
    n = int(np.prod(shape))
    ml = np.prod(spacing) / 1000.0
    d_m, d_e, d_t = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
    out: Dict[str, Any] = {}
    try:
        for region in [r for r in label_map if r in set(regions)]:
            is_auto = "autochthon" in region
            label_hu_mask(ctx, d_ct, d_lab, [label_map[region]], 2 if is_auto else 0, n, d_m)
            binary_erode(ctx, d_m, d_e, d_t, shape)
            st = _masked_stats_local(ctx, d_ct, d_e, n) if hist[label_map[region]].any() else None
            out[region] = _metrics(st, ml, am, asd, cnr_none=region.partition("_")[0] == "autochthon")
    finally:
        for b in (d_m, d_e, d_t):
            b.free()
    return out


def total_measurements(ctx: Context, ct: Optional[np.ndarray], total_seg: Optional[np.ndarray], label_map: Dict[str, int],
                       spacing, cnr_adjustment: bool = True, model_name: str = "total", d_ct: Optional[DeviceBuffer] = None,
                       d_lab: Optional[DeviceBuffer] = None, shape=None, mask_on_device: bool = False, shard=None,
                       defer_host: bool = False):
    """compute_measurements (:244-343) for models == ["total"] on (z,y,x) arrays.  Returns (measurements dict, ct_pfav
    mask).  `defer_host=True`: the device passes run now, the ~120 per-label order statistics (pure numpy on the downloaded
    histogram) are returned as a function `finish() -> measurements dict` in place of the dict, so that a caller can run them
    on a worker thread while the device works on the next stage (bench.py: under the BCA nets).  Resident inputs: pass `d_ct` (int16) / `d_lab` (uint8) + `shape` instead of the host arrays (not freed here);
    `mask_on_device=True` returns the mask as a DeviceBuffer (caller frees).

    `shard` = (agg_shard.AggComm, (a, b)): the arrays are this rank's z-slab of the volume INCLUDING a halo of
    agg_shard.ERODE_REACH planes towards its neighbours, and planes [a, b) of them are the ones the rank owns.  Masks and
    erosions run on the whole slab (the halo supplies the neighbours' voxels; its own planes are discarded), histograms only
    over the owned planes and are summed over the ranks -- every rank returns the volume's measurements; the returned
    ct_pfav mask covers the slab."""
    own = d_ct is None
    if own:
        if ct.shape != total_seg.shape:
            raise ValueError("The spacing of the image and of the segmentation should be the same")  # shape contract
        shape = ct.shape
        d_ct = ctx.from_numpy(np.ascontiguousarray(ct, dtype=np.int16))
        d_lab = ctx.from_numpy(np.ascontiguousarray(total_seg, dtype=np.uint8))
    shape = tuple(int(v) for v in shape)
    n = int(np.prod(shape))
    ml = np.prod(spacing) / 1000.0
    meas: Dict[str, Any] = {"segmentations": {}, "info": {}}
    d_m, d_e, d_t = ctx.alloc(max(n, 1)), ctx.alloc(max(n, 1)), ctx.alloc(max(n, 1))
    keep_mask = None
    if shard is not None:
        from .device import BufferView
        comm, (own_a, own_b) = shard
        plane = shape[1] * shape[2]
        n_own = (own_b - own_a) * plane

        def owned(buf, itemsize):
            return BufferView(buf, own_a * plane * itemsize, n_own * itemsize)

        def label_hu_histogram(ctx_, dc, dl, n_, d_mask=None):          # noqa: F811  (slab version: owned planes, all-reduced)
            h = _label_hu_histogram_local(ctx_, owned(dc, 2), owned(dl, 1), n_own, owned(d_mask, 1) if d_mask is not None else None) \
                if n_own else np.zeros((256, NBINS), np.uint32)
            return comm.all_reduce_sum(h)

        def _masked_stats(ctx_, dc, d_mask, n_):                       # noqa: F811
            return stats_from_hist(label_hu_histogram(ctx_, dc, d_mask, n_)[1])
    else:
        label_hu_histogram = _label_hu_histogram_local
        _masked_stats = _masked_stats_local
    try:
        hist = label_hu_histogram(ctx, d_ct, d_lab, n)
        # autochthon reference: (left | right) minus fat, eroded (:42-58)
        auto_labels = [label_map["autochthon_left"], label_map["autochthon_right"]]
        label_hu_mask(ctx, d_ct, d_lab, auto_labels, 2, n, d_m)
        binary_erode(ctx, d_m, d_e, d_t, shape)
        st_auto = _masked_stats(ctx, d_ct, d_e, n)
        am, asd = (st_auto["mean"], st_auto["std"]) if st_auto else (None, None)
        # pulmonary fat (ct_pfav, :151-200): the fat window of a label (union) is a bin range of its histogram rows
        lo, hi = ADIPOSE_TISSUE[0] - HU_MIN, ADIPOSE_TISSUE[1] - HU_MIN

        def fat_metrics(names):
            h = np.zeros(hi + 1 - lo, dtype=np.int64)
            for nm in names:
                h += hist[label_map[nm]][lo:hi + 1]
            return _metrics(stats_from_hist(h, ADIPOSE_TISSUE[0]), ml, am, asd)

        def host_tables():
            seg = _metrics_from_hist(hist, label_map, am, asd, spacing)
            pf = {}
            for nm in LUNG_MASKS:
                pf["ct_pfav_" + nm] = fat_metrics([nm])
            for side in ("left", "right"):
                pf[f"ct_pfav_lobe_{side}"] = fat_metrics([nm for nm in LUNG_MASKS if nm.endswith(side)])
            pf["ct_pfav_lungs"] = fat_metrics(LUNG_MASKS)
            return {**seg, **pf}

        label_hu_mask(ctx, d_ct, d_lab, [label_map[nm] for nm in LUNG_MASKS], 1, n, d_m)
        if mask_on_device:
            keep_mask = ctx.alloc(n)
            check(ctx.lib.boa_copy3(ctx.h, d_m.vp, 0, 0, (C.c_longlong * 3)(0, 0, 1), (C.c_int * 3)(1, 1, n), keep_mask.vp, 0, 0,
                                    (C.c_longlong * 3)(0, 0, 1)), "boa_copy3")
            fat_mask = keep_mask
        else:
            fat_mask = d_m.download(shape, np.uint8)
        if cnr_adjustment and model_name in CNR_ADJUSTED_REGIONS:
            if am is not None and asd is not None:
                adj = {}
                regions = [r for r in label_map if r in CNR_ADJUSTED_REGIONS[model_name]]
                for region in regions:
                    is_auto = "autochthon" in region
                    label_hu_mask(ctx, d_ct, d_lab, [label_map[region]], 2 if is_auto else 0, n, d_m)
                    binary_erode(ctx, d_m, d_e, d_t, shape)
                    st = _masked_stats(ctx, d_ct, d_e, n) if hist[label_map[region]].any() else None
                    adj[region] = _metrics(st, ml, am, asd, cnr_none=region.partition("_")[0] == "autochthon")
                if "autochthon_left" in label_map and "autochthon_right" in label_map and \
                        {"autochthon_left", "autochthon_right"} <= set(regions):
                    # union, minus fat, eroded == the reference mask computed above
                    present = hist[auto_labels[0]].any() or hist[auto_labels[1]].any()
                    adj["autochthon"] = _metrics(st_auto if present else None, ml, am, asd, cnr_none=True)
                meas.setdefault("cnr_adjusted", {}).update(adj)
        meas["info"]["autochthon_mean"] = am
        meas["info"]["autochthon_std"] = asd
        keep_mask = None

        def finish():
            out = {"segmentations": {model_name: host_tables()}, "info": meas["info"]}      # key order of the reference's dict
            if "cnr_adjusted" in meas:
                out["cnr_adjusted"] = meas["cnr_adjusted"]
            return out

        return (finish if defer_host else finish()), fat_mask
    finally:
        for b in ((d_ct, d_lab) if own else ()) + (d_m, d_e, d_t) + ((keep_mask,) if keep_mask is not None else ()):
            b.free()
