"""Aggregation stages of one volume on several GPUs (last row): the (z, y, x) volumes of the measurement /
body-composition path are cut into contiguous z-slabs, one per rank.

  per-slice tables   tissue counts / HU sums (boa_tissue_aggregate) and slice-wise label presence are slab-local: the ranks'
                     rows are concatenated (all-gather of a few KB);
  per-label tables   the HU histogram of every label is a sum over voxels: slab histograms are all-reduced (the 4 MB
                     exchange of the survey -- 16 MB per 65 536-bin row block here);
  erosions           (CNR masks, 6^3 footprint = 3 voxels reach) run on the slab plus a 3-plane halo that each rank cuts from
                     its own copy of the volume; the halo planes of the result are discarded;
  connected components (CC filters of the BCA post-processing): every rank labels its slab (boa_ccl26), the components
                     that touch a slab interface are merged by a union-find over the two boundary planes of each interface
                     (26-connectivity: the 3 x 3 neighbourhood across the interface), which every rank evaluates
                     identically on the gathered planes + component tables -- one small exchange; the merged sizes / the
                     global winner then drive the unchanged slab-local kernels.

Results are bit-identical to the single-GPU stages (integers; the floating-point statistics are computed from the same
integer tables on every rank).  The protocol is written against a small engine interface so that the CPU tests drive it
over gloo with a numpy / scipy engine; `HipAggEngine` is the product engine on the C ABI.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

ERODE_REACH = 3       # erode_region: 6^3 ones kernel padded to 7^3, anchored at its centre (BOA/compute/measurements.py:61-71)
MARK = 0xFFFFFFFF     # `sizes[root]` of the pieces of the globally largest component (boa_ccl_fill_unmarked)


def slab_bounds(Z: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced z ranges [lo, hi) of the ranks (ranks beyond Z get empty slabs)."""
    q, r = divmod(int(Z), int(world))
    out, lo = [], 0
    for i in range(world):
        hi = lo + q + (1 if i < r else 0)
        out.append((lo, hi))
        lo = hi
    return out


class AggComm:
    """What the protocol needs from torch.distributed (gloo on host arrays; the tables are small)."""

    def __init__(self, dist, rank: int, world: int):
        self.dist, self.rank, self.world = dist, int(rank), int(world)

    def all_gather(self, obj) -> list:
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def all_reduce_sum(self, arr: np.ndarray) -> np.ndarray:
        if self.world == 1:
            return arr
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr.astype(np.int64)))      # exact for the uint32 counts of any volume
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.numpy()


# ---------------------------------------------------------------------------------------------------- per-slice / per-label tables
def gather_slice_tables(comm: AggComm, *tables: np.ndarray) -> List[np.ndarray]:
    """Slab-local per-slice tables (first axis = the slab's z) -> the whole volume's tables on every rank."""
    parts = comm.all_gather(tuple(np.ascontiguousarray(t) for t in tables))
    return [np.concatenate([p[i] for p in parts], axis=0) for i in range(len(tables))]


def reduce_histogram(comm: AggComm, hist: np.ndarray) -> np.ndarray:
    """Slab histograms [labels][bins] -> the volume's histogram on every rank (integer sum: exact)."""
    return comm.all_reduce_sum(hist)


# ---------------------------------------------------------------------------------------------------- connected components
class _UF:
    def __init__(self):
        self.p: Dict[int, int] = {}

    def find(self, a: int) -> int:
        p = self.p
        r = a
        while p.get(r, r) != r:
            r = p[r]
        while p.get(a, a) != r:
            p[a], a = r, p[a]
        return r

    def union(self, a: int, b: int):
        ra, rb = self.find(a), self.find(b)
        if ra != rb:
            if ra < rb:
                self.p[rb] = ra
            else:
                self.p[ra] = rb


def merge_components(comm: AggComm, roots_first: Optional[np.ndarray], roots_last: Optional[np.ndarray], comp_roots: np.ndarray,
                     comp_sizes: np.ndarray):
    """Collective.  This rank's slab was labelled on its own: `comp_roots` are the GLOBAL linear indices of the first voxel
    (raster order) of its components, `comp_sizes` their slab-local sizes, `roots_first` / `roots_last` the [Y][X] maps of
    global root ids (-1 = background) of the slab's first / last plane (None for an empty slab).
    Returns (rep_of, size_of, n_merged): for every local component root its merged component's id (= the smallest global
    root of the set, i.e. the component's first voxel in the whole volume: the order in which skimage numbers components)
    and merged size; n_merged = number of components of the whole volume."""
    parts = comm.all_gather((roots_first, roots_last, np.asarray(comp_roots, dtype=np.int64), np.asarray(comp_sizes, dtype=np.int64)))
    uf = _UF()
    live = [p for p in parts if p[0] is not None]          # ranks with a non-empty slab, in z order
    for lower, upper in zip(live[:-1], live[1:]):
        a, b = lower[1], upper[0]                            # last plane below the interface, first plane above it
        Y, X = a.shape
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ya0, ya1 = max(0, -dy), min(Y, Y - dy)
                xa0, xa1 = max(0, -dx), min(X, X - dx)
                va = a[ya0:ya1, xa0:xa1]
                vb = b[ya0 + dy:ya1 + dy, xa0 + dx:xa1 + dx]
                m = (va >= 0) & (vb >= 0)
                if m.any():
                    pairs = np.unique(np.stack([va[m], vb[m]], axis=1), axis=0)
                    for x, y in pairs:
                        uf.union(int(x), int(y))
    all_roots = np.concatenate([p[2] for p in parts]) if parts else np.zeros(0, np.int64)
    all_sizes = np.concatenate([p[3] for p in parts]) if parts else np.zeros(0, np.int64)
    reps = np.array([uf.find(int(r)) for r in all_roots], dtype=np.int64)
    size_by_rep: Dict[int, int] = {}
    for rep, s in zip(reps, all_sizes):
        size_by_rep[int(rep)] = size_by_rep.get(int(rep), 0) + int(s)
    rep_of = {int(r): uf.find(int(r)) for r in comp_roots}
    size_of = {int(r): size_by_rep[rep_of[int(r)]] for r in comp_roots}
    return rep_of, size_of, size_by_rep


def filter_largest_sharded(comm: AggComm, engine, mask, seg, z0: int, fill_value: int = 255):
    """_filter_largest_unique_segment (BCA/body_regions/postprocess.py:8-15) on a z-slab: every component of `mask` except
    the volume's largest (ties: the one numbered first = smallest first-voxel index) gets seg = fill_value."""
    cc = engine.ccl(mask)
    rf, rl, roots, sizes = engine.boundary_and_components(cc, z0)
    rep_of, _, size_by_rep = merge_components(comm, rf, rl, roots, sizes)
    if len(size_by_rep) <= 1:
        return
    best = min(size_by_rep.items(), key=lambda kv: (-kv[1], kv[0]))[0]
    keep = [r for r in roots if rep_of[int(r)] == best]
    engine.fill_all_but(cc, keep, z0, seg, fill_value)


def remove_small_sharded(comm: AggComm, engine, mask, z0: int, max_size: int):
    """skimage.morphology.remove_small_objects(mask, max_size + 1, connectivity=3) on a z-slab, in place."""
    cc = engine.ccl(mask)
    rf, rl, roots, sizes = engine.boundary_and_components(cc, z0)
    _, size_of, _ = merge_components(comm, rf, rl, roots, sizes)
    changed = [(int(r), int(size_of[int(r)])) for r, s in zip(roots, sizes) if size_of[int(r)] != int(s)]
    engine.remove_small(cc, changed, z0, max_size, mask)


# ---------------------------------------------------------------------------------------------------- engines
class NumpyAggEngine:
    """CPU engine for the protocol tests: scipy.ndimage connected components on host arrays (z, y, x)."""

    def ccl(self, mask):
        from scipy import ndimage
        lab, n = ndimage.label(np.asarray(mask) != 0, structure=np.ones((3, 3, 3)))
        idx = np.arange(lab.size, dtype=np.int64).reshape(lab.shape)
        first = np.asarray(ndimage.minimum(idx, lab, index=np.arange(1, n + 1)), dtype=np.int64) if n else np.zeros(0, np.int64)
        sizes = np.bincount(lab.ravel(), minlength=n + 1)[1:]
        roots = np.where(lab > 0, first[np.maximum(lab, 1) - 1], -1)
        return {"roots": roots, "first": first, "sizes": sizes.astype(np.int64), "shape": lab.shape}

    def boundary_and_components(self, cc, z0):
        Zl, Y, X = cc["shape"]
        off = int(z0) * Y * X
        if Zl == 0:
            return None, None, np.zeros(0, np.int64), np.zeros(0, np.int64)
        g = np.where(cc["roots"] >= 0, cc["roots"] + off, -1)
        return g[0].copy(), g[-1].copy(), cc["first"] + off, cc["sizes"]

    def fill_all_but(self, cc, keep_global_roots, z0, seg, fill_value):
        Zl, Y, X = cc["shape"]
        off = int(z0) * Y * X
        keep = np.isin(cc["roots"] + off, np.asarray(list(keep_global_roots), dtype=np.int64)) & (cc["roots"] >= 0)
        seg[(cc["roots"] >= 0) & ~keep] = fill_value

    def remove_small(self, cc, changed, z0, max_size, mask):
        Zl, Y, X = cc["shape"]
        off = int(z0) * Y * X
        size_map = {int(f): int(s) for f, s in zip(cc["first"], cc["sizes"])}
        for groot, s in changed:
            size_map[groot - off] = s
        if Zl == 0:
            return
        lut_keys = np.array(sorted(size_map), dtype=np.int64)
        lut_vals = np.array([size_map[k] for k in lut_keys], dtype=np.int64)
        r = cc["roots"]
        sz = np.zeros(r.shape, dtype=np.int64)
        fg = r >= 0
        sz[fg] = lut_vals[np.searchsorted(lut_keys, r[fg])]
        mask[fg & (sz <= max_size)] = 0


class HipAggEngine:
    """Product engine: slab masks / label volumes are resident uint8 buffers (DeviceBuffer) of shape (Zl, Y, X)."""

    def __init__(self, ctx, shape_slab: Sequence[int]):
        self.ctx, self.lib = ctx, ctx.lib
        self.shape = tuple(int(v) for v in shape_slab)
        self.n = int(np.prod(self.shape))
        self.d_roots = ctx.alloc(max(self.n, 1) * 4)
        self.d_sizes = ctx.alloc(max(self.n, 1) * 4)

    def close(self):
        self.d_roots.free()
        self.d_sizes.free()

    def ccl(self, d_mask):
        from ._lib import check
        Zl, Y, X = self.shape
        ncomp = C.c_int(0)
        if self.n:
            check(self.lib.boa_ccl26(self.ctx.h, d_mask.vp, Zl, Y, X, self.d_roots.vp, self.d_sizes.vp, C.byref(ncomp)), "boa_ccl26")
        return {"n": ncomp.value}

    def boundary_and_components(self, cc, z0):
        from ._lib import check
        Zl, Y, X = self.shape
        if self.n == 0:
            return None, None, np.zeros(0, np.int64), np.zeros(0, np.int64)
        off = int(z0) * Y * X
        roots_all = None
        planes = []
        for z in ((0,) if Zl == 1 else (0, Zl - 1)):
            host = np.empty((Y, X), dtype=np.int32)
            check(self.lib.boa_d2h(self.ctx.h, host.ctypes.data_as(C.c_void_p), C.c_void_p(self.d_roots.ptr + z * Y * X * 4), Y * X * 4),
                  "boa_d2h")
            planes.append(np.where(host >= 0, host.astype(np.int64) + off, -1))
        m = max(cc["n"], 1)
        hr, hs, cnt = np.empty(m, np.int32), np.empty(m, np.uint32), C.c_int(0)
        check(self.lib.boa_ccl_list_components(self.ctx.h, self.d_sizes.vp, self.n, m, hr.ctypes.data_as(C.c_void_p),
                                               hs.ctypes.data_as(C.c_void_p), C.byref(cnt)), "boa_ccl_list_components")
        k = min(cnt.value, m)
        order = np.argsort(hr[:k])
        del roots_all
        return planes[0], planes[-1], hr[:k][order].astype(np.int64) + off, hs[:k][order].astype(np.int64)

    def _scatter(self, idx, val):
        from ._lib import check
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.uint32)
        check(self.lib.boa_scatter_u32(self.ctx.h, self.d_sizes.vp, idx.ctypes.data_as(C.c_void_p), val.ctypes.data_as(C.c_void_p),
                                       len(idx)), "boa_scatter_u32")

    def fill_all_but(self, cc, keep_global_roots, z0, d_seg, fill_value):
        from ._lib import check
        if self.n == 0:
            return
        Zl, Y, X = self.shape
        off = int(z0) * Y * X
        keep = np.array([int(r) - off for r in keep_global_roots], dtype=np.int64)
        self._scatter(keep, np.full(len(keep), MARK, dtype=np.uint32))
        check(self.lib.boa_ccl_fill_unmarked(self.ctx.h, self.d_roots.vp, self.d_sizes.vp, self.n, MARK, d_seg.vp, int(fill_value)),
              "boa_ccl_fill_unmarked")

    def remove_small(self, cc, changed, z0, max_size, d_mask):
        from ._lib import check
        if self.n == 0:
            return
        Zl, Y, X = self.shape
        off = int(z0) * Y * X
        if changed:
            self._scatter([g - off for g, _ in changed], [min(s, 0xFFFFFFFE) for _, s in changed])
        check(self.lib.boa_ccl_remove_small(self.ctx.h, self.d_roots.vp, self.d_sizes.vp, self.n, int(max_size), d_mask.vp),
              "boa_ccl_remove_small")


# ---------------------------------------------------------------------------------------------------- BCA region post-processing on slabs
def postprocess_region_segmentation_sharded(comm: AggComm, engine, select, seg, z0: int, region_ids: Dict[str, int]):
    """postprocess_region_segmentation (BCA/body_regions/postprocess.py:18-40) on this rank's z-slab of the body_regions
    labels, in place: the four CC filters in the reference's order.  `select(seg, mode, vals) -> mask` builds the mask on the
    engine's side (mode 1: seg > 0; 2: seg in vals; 0: seg == vals[0])."""
    R = region_ids
    for mode, vals in ((1, (0, 0, 0)), (2, (R["THORACIC_CAVITY"], R["MEDIASTINUM"], R["PERICARDIUM"])), (0, (R["PERICARDIUM"], 0, 0)),
                       (0, (R["ABDOMINAL_CAVITY"], 0, 0))):
        filter_largest_sharded(comm, engine, select(seg, mode, vals), seg, z0, 255)
