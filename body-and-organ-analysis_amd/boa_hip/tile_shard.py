"""One volume on several GPUs: the tiles of the sliding-window predictor and the Gaussian-weighted aggregation buffer are
partitioned across the ranks of one node, the overlap regions travel over RCCL / xGMI (granularity 3).

Partition.  The tile grid of `_internal_get_sliding_window_slicers` (NN/inference/predict_from_raw_data.py:523-558) is
ordered axis 0 outermost, and the reference adds tile after tile into fp16 buffers (:611-614), so per voxel the rounding
sequence is "ascending tile index".  Ranks therefore own BLOCKS OF TILE ROWS along axis 0: every tile of rank r precedes
every tile of rank r+1, and only the slab where the last row of r and the first row of r+1 overlap (patch - step planes,
32 of 128 at step 0.8) is touched by both.

Two exchange modes:

* ``exact`` (default) -- rank r+1 does not add the slab planes of its first-row tiles at once; it keeps their head input
  (boa_net_predict_sliding_window_deferred), receives rank r's finished partial sums for the slab (one send/recv per
  boundary, all boundaries concurrently on distinct xGMI links), and then adds the kept planes in tile order
  (boa_net_apply_deferred).  Per voxel the fp16 `+=` sequence is the reference's: labels AND logits are bit-identical
  to the single-GPU loop.  No rank waits for more than its direct neighbour: the slab a rank sends never contains
  planes it deferred (plan_rows checks that blocks two apart do not overlap).
* ``allreduce`` -- every rank accumulates all of its tiles from zero and the two partial sums of each slab are added by a
  2-rank RCCL all-reduce in fp16.  One rounding differs from the reference's sequence (P + Q instead of adding Q's tiles
  one by one), so labels can flip where two logits are within an fp16 ulp; the GPU test reports the flip fraction.

After the exchange each rank owns a disjoint range of axis-0 planes: it runs normalise + argmax + remap on those planes
only (boa_finalize_labels_planes), and one uint8 all-reduce (sum over disjoint supports) hands every rank the part's label
volume, which is then merged into the combined volume in part order (TS/nnunet.py:553-556) identically on all ranks.

The protocol (`run_fold_sharded`) is written against a small engine interface so that the CPU tests drive it over gloo
with a numpy engine; `HipShardEngine` is the product engine on top of the C ABI.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ._lib import check, int3


# ---------------------------------------------------------------------------------------------------- plan
@dataclass
class RowPlan:
    rows: List[int]                    # sorted axis-0 tile starts
    blocks: List[Tuple[int, int]]      # per active block: row indices [b0, b1)
    patch0: int
    pv0: int
    row_of_tile: np.ndarray            # [n_tiles] row index of every tile (canonical order)
    ranks: Optional[List[int]] = None  # global rank that owns block i (default: rank i) -- (row x model) units put the blocks of
                                       # one model on a subset of the ranks (plan_units)

    def __post_init__(self):
        if self.ranks is None:
            self.ranks = list(range(len(self.blocks)))
        assert len(self.ranks) == len(self.blocks) and len(set(self.ranks)) == len(self.ranks)

    @property
    def active(self) -> int:
        return len(self.blocks)

    def index(self, rank: int) -> Optional[int]:
        """Block index of global rank `rank`, None when the rank has no tiles of this model."""
        return self.ranks.index(rank) if rank in self.ranks else None

    def tiles(self, rank: int) -> np.ndarray:
        """Indices (canonical order kept) of the tiles of `rank`."""
        i = self.index(rank)
        if i is None:
            return np.zeros(0, dtype=np.int64)
        b0, b1 = self.blocks[i]
        return np.nonzero((self.row_of_tile >= b0) & (self.row_of_tile < b1))[0]

    def boundary(self, k: int) -> Optional[Tuple[int, int]]:
        """Planes [lo, hi) shared by block k (its last row) and block k+1 (its first row); None if they do not overlap."""
        if k < 0 or k + 1 >= self.active:
            return None
        first_upper = self.blocks[k + 1][0]
        lo, hi = self.rows[first_upper], self.rows[first_upper - 1] + self.patch0
        return (lo, hi) if hi > lo else None

    def owned_planes(self, rank: int) -> Tuple[int, int]:
        """Planes whose sums are complete on `rank` after the exchange (a partition of [0, pv0))."""
        i = self.index(rank)
        if i is None:
            return (0, 0)
        lo = 0 if i == 0 else self.rows[self.blocks[i][0]]
        hi = self.pv0 if i == self.active - 1 else self.rows[self.blocks[i + 1][0]]
        return (lo, hi)

    def defer_planes(self, rank: int) -> np.ndarray:
        """Per tile of `rank`: number of leading planes that must wait for the lower rank (exact mode) = the part of
        the tile below the end of the lower rank's last row (normally only the first row of the block reaches it)."""
        t = self.tiles(rank)
        out = np.zeros(len(t), dtype=np.int32)
        i = self.index(rank)
        b = self.boundary(i - 1) if i is not None else None
        if b is not None:
            starts = np.asarray(self.rows)[self.row_of_tile[t]]
            out[:] = np.clip(b[1] - starts, 0, self.patch0)
        return out


def _two_apart_ok(rows, blocks, patch0) -> bool:
    # blocks two apart must not overlap: then the slab a rank sends up never contains planes it deferred itself
    return all(rows[blocks[i][1] - 1] + patch0 <= rows[blocks[i + 2][0]] for i in range(len(blocks) - 2))


def plan_rows(origins: np.ndarray, patch0: int, pv0: int, world: int, assignment: Optional[Sequence[Tuple[int, int]]] = None) -> RowPlan:
    """Contiguous, balanced blocks of tile rows for at most `world` ranks.  The number of active ranks is reduced until
    blocks two apart do not overlap (small volumes, where the actual step can fall to 0.4 * patch).
    `assignment` = [(global rank, number of rows), ...] in row order (plan_units): those blocks on those ranks; when they violate
    the two-apart rule the rows are re-balanced over the same ranks (fewer of them if need be)."""
    origins = np.asarray(origins).reshape(-1, 3)
    rows = sorted(set(int(v) for v in origins[:, 0]))
    row_of_tile = np.searchsorted(np.asarray(rows), origins[:, 0]).astype(np.int64)
    if np.any(np.diff(row_of_tile) < 0):
        raise ValueError("tile origins must be in canonical (axis 0 outermost) order")
    rank_ids = None
    if assignment is not None:
        assignment = [(int(r), int(c)) for r, c in assignment if int(c) > 0]
        if sum(c for _, c in assignment) != len(rows):
            raise ValueError(f"row assignment {assignment} does not cover the {len(rows)} tile rows")
        blocks, lo = [], 0
        for _, c in assignment:
            blocks.append((lo, lo + c))
            lo += c
        rank_ids = [r for r, _ in assignment]
        if _two_apart_ok(rows, blocks, patch0):
            return RowPlan(rows, blocks, int(patch0), int(pv0), row_of_tile, rank_ids)
        world = len(rank_ids)
    for a in range(max(1, min(world, len(rows))), 0, -1):
        q, r = divmod(len(rows), a)
        blocks, lo = [], 0
        for i in range(a):
            hi = lo + q + (1 if i < r else 0)
            blocks.append((lo, hi))
            lo = hi
        if _two_apart_ok(rows, blocks, patch0):
            return RowPlan(rows, blocks, int(patch0), int(pv0), row_of_tile, None if rank_ids is None else rank_ids[:a])
    raise AssertionError("unreachable: one block is always valid")


def plan_units(rows_per_model: Sequence[int], world: int, weights: Optional[Sequence[float]] = None) -> List[List[Tuple[int, int]]]:
    """(row x model) work units of a multi-model task (the five part models of `total` are independent until the label merge,
    TS/nnunet.py:542-556): the units -- tile row j of model m, model-major order -- are cut into `world` contiguous runs of
    near-equal weight (weight of a unit = `weights[m]`, e.g. its tiles per row; default 1).  -> per model the
    [(rank, number of its rows), ...] blocks in row order: plan_rows(..., assignment=that).  A 512^3 `total` volume has
    5 x 5 units: tile rows alone keep 5 of 8 ranks busy, units all 8 (3-4 units each)."""
    rows_per_model = [int(v) for v in rows_per_model]
    w = [1.0] * len(rows_per_model) if weights is None else [float(v) for v in weights]
    total = sum(r * wm for r, wm in zip(rows_per_model, w))
    out: List[List[Tuple[int, int]]] = []
    cum = 0.0
    for m, n_rows in enumerate(rows_per_model):
        blocks: List[Tuple[int, int]] = []
        for _ in range(n_rows):
            # the rank whose share [r, r + 1) * total / world holds the unit's centre: monotone in the unit index -> contiguous runs
            r = min(world - 1, int((cum + 0.5 * w[m]) * world / total)) if total > 0 else 0
            cum += w[m]
            if blocks and blocks[-1][0] == r:
                blocks[-1] = (r, blocks[-1][1] + 1)
            else:
                blocks.append((r, 1))
        out.append(blocks)
    return out


# ---------------------------------------------------------------------------------------------------- transport
class ShardComm:
    """torch.distributed transport for the slabs ("nccl" = RCCL with device tensors, "gloo" with host tensors)."""

    def __init__(self, dist, rank: int, world: int, device: str = "cpu"):
        self.dist, self.rank, self.world, self.device = dist, int(rank), int(world), device
        self._pairs = None

    @property
    def on_device(self) -> bool:
        return str(self.device).startswith("cuda")

    def empty(self, shape, dtype):
        import torch
        return torch.empty(tuple(int(s) for s in shape), dtype=dtype, device=self.device)

    def _done(self):
        if self.on_device:
            import torch
            torch.cuda.synchronize()

    def shift(self, send, dst, recv, src):
        """send -> rank `dst`, recv <- rank `src` (either side may be None); every boundary moves at the same time."""
        ops = []
        if send is not None:
            ops.append(self.dist.P2POp(self.dist.isend, send, int(dst)))
        if recv is not None:
            ops.append(self.dist.P2POp(self.dist.irecv, recv, int(src)))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
            self._done()

    def shift_up(self, send, recv):
        self.shift(send, self.rank + 1, recv, self.rank - 1)

    def all_reduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        self._done()

    def make_room(self, ctx):
        """RCCL / torch allocate from the same HBM as the engine's caching allocator, whose parked blocks they cannot reclaim:
        release them before the first exchange buffers of a volume are created (device transport only)."""
        if self.on_device and ctx.info()["free_mem"] < (16 << 30):   # (only when HBM is actually short: re-filling the pool costs hipMallocs)
            ctx.lib.boa_trim(ctx.h)


# ---------------------------------------------------------------------------------------------------- protocol
def _exchange_via_tensors(engine, comm, dst, upper, src, lower, add: bool):
    """Default slab exchange of an engine without its own `exchange`: pack -> shift -> unpack (exact hand-over) or add."""
    send = engine.pack(*upper) if upper else None
    recv = engine.empty(*lower) if lower else None
    if hasattr(comm, "shift"):
        comm.shift(send, dst, recv, src)
    else:                                    # (rank +- 1 neighbours only)
        comm.shift_up(send, recv)
    if lower:
        if add:      # P + Q with one RTNE rounding per element: what a two-rank fp16 sum all-reduce leaves on the slab's owner
            cur = engine.pack(*lower)
            recv = (cur.float() + recv.float()).to(cur.dtype)
        engine.unpack(lower[0], lower[1], recv)
    return None


def start_fold_sharded(engine, plan: RowPlan, comm, mode: str = "exact"):
    """First half of one fold of one model on this rank: zero the accumulators, run this rank's tiles, QUEUE the slab exchange
    with the neighbouring blocks' ranks.  With the RCCL transport (RcclComm + HipShardEngine) nothing here waits: the exchange runs
    on the communication stream while the caller queues the next model's tiles; `finish_fold_sharded` orders the rest behind it.
    Engine interface:
         begin()                         zero the accumulators
         run(tile_idx, defer) -> stash   forward + accumulate this rank's tiles; `defer[i]` leading planes kept back
         pack(lo, hi) -> tensor          planes [lo, hi) of (acc channels..., n) as one fp16 tensor on comm.device
         empty(lo, hi) -> tensor         receive buffer of the same shape
         unpack(lo, hi, tensor)          overwrite those planes
         apply(stash)                    add the kept planes in tile order
         exchange(comm, dst, upper, src, lower, add) -> token     (optional) the engine's own transport of the slabs
         complete(token)                 (with exchange) make later engine work wait for the exchange; add the received sums"""
    if mode not in ("exact", "allreduce"):
        raise ValueError(f"unknown tile-shard mode {mode!r}")
    engine.begin()
    i = plan.index(comm.rank)
    if i is None:
        return None
    exact = mode == "exact"
    tiles = plan.tiles(comm.rank)
    lower, upper = plan.boundary(i - 1), plan.boundary(i)
    dst = plan.ranks[i + 1] if upper else None
    src = plan.ranks[i - 1] if lower else None
    stash = engine.run(tiles, plan.defer_planes(comm.rank) if exact else np.zeros(len(tiles), dtype=np.int32))
    if hasattr(engine, "exchange"):
        token = engine.exchange(comm, dst, upper, src, lower, not exact)
    else:
        token = _exchange_via_tensors(engine, comm, dst, upper, src, lower, not exact)
    return (engine, plan, comm, exact, stash, token)


def finish_fold_sharded(state) -> Tuple[int, int]:
    """Second half: the received partial sums are in place (exact) / added (allreduce), the kept planes are applied in tile
    order.  Returns the planes this rank owns afterwards."""
    if state is None:
        return (0, 0)
    engine, plan, comm, exact, stash, token = state
    if hasattr(engine, "complete"):
        engine.complete(token)
    if exact:
        engine.apply(stash)
    return plan.owned_planes(comm.rank)


def run_fold_sharded(engine, plan: RowPlan, comm, mode: str = "exact") -> Tuple[int, int]:
    """One fold of one model on this rank, both halves back to back."""
    return finish_fold_sharded(start_fold_sharded(engine, plan, comm, mode))


# ---------------------------------------------------------------------------------------------------- product engine
class _Foreign:
    """Non-owning view of device memory that belongs to a torch tensor (same attributes as DeviceBuffer)."""

    def __init__(self, tensor):
        self._keep = tensor
        self.ptr = tensor.data_ptr()
        self.nbytes = tensor.numel() * tensor.element_size()

    @property
    def vp(self):
        return C.c_void_p(self.ptr)

    def free(self):
        pass


class HipShardEngine:
    """Engine over one HipPredictor fold: acc fp16 [C][PV] and n fp16 [PV] resident on this rank's GPU."""

    def __init__(self, predictor, comm: ShardComm, dvol, V, PV, below, origins, acc, nacc):
        self.p, self.comm = predictor, comm
        self.ctx, self.lib = predictor.ctx, predictor.lib
        self.dvol, self.V, self.PV, self.below = dvol, list(V), list(PV), list(below)
        self.origins = np.ascontiguousarray(origins, dtype=np.int32).reshape(-1, 3)
        self.acc, self.nacc = acc, nacc
        self.C = predictor.geom.num_classes
        self._stage = None

    def begin(self):
        self.acc.zero()
        self.nacc.zero()

    def run(self, tile_idx, defer):
        org = np.ascontiguousarray(self.origins[np.asarray(tile_idx, dtype=np.int64)], dtype=np.int32)
        defer = np.ascontiguousarray(defer, dtype=np.int32)
        g = self.p._gaussian()
        st = C.c_void_p()
        if len(org) == 0:
            return None
        check(self.lib.boa_net_predict_sliding_window_deferred(
            self.p._net, self.dvol.vp, int3(self.V), int3(self.PV), int3(self.below), org.ctypes.data_as(C.POINTER(C.c_int)),
            len(org), g.vp if g else None, self.acc.vp, self.nacc.vp, defer.ctypes.data_as(C.POINTER(C.c_int)), C.byref(st)),
            "boa_net_predict_sliding_window_deferred")
        return st

    def apply(self, stash):
        if stash is None:
            return
        try:
            g = self.p._gaussian()
            check(self.lib.boa_net_apply_deferred(self.p._net, stash, g.vp if g else None, self.acc.vp, self.nacc.vp,
                                                  int3(self.PV)), "boa_net_apply_deferred")
        finally:
            self.lib.boa_stash_destroy(stash)

    # ---- slabs: (C + 1) x planes x PV1*PV2 halves, packed by the strided device copy ---------------------------------
    def _move(self, lo, hi, flat, to_flat: bool):
        pl, plane = hi - lo, self.PV[1] * self.PV[2]
        vv = self.PV[0] * plane
        ll3 = lambda *v: (C.c_longlong * 3)(*v)  # noqa: E731
        for buf, ch, off in ((self.acc, self.C, 0), (self.nacc, 1, self.C * pl * plane)):
            dims = (C.c_int * 3)(ch, pl, plane)
            if to_flat:
                check(self.lib.boa_copy3(self.ctx.h, buf.vp, 1, lo * plane, ll3(vv, plane, 1), dims, flat.vp, 1, off,
                                         ll3(pl * plane, plane, 1)), "boa_copy3")
            else:
                check(self.lib.boa_copy3(self.ctx.h, flat.vp, 1, off, ll3(pl * plane, plane, 1), dims, buf.vp, 1, lo * plane,
                                         ll3(vv, plane, 1)), "boa_copy3")

    def _staging(self, nbytes):
        if self._stage is None or self._stage.nbytes < nbytes:
            if self._stage is not None:
                self._stage.free()
            self._stage = self.ctx.alloc(nbytes)
        return self._stage

    def empty(self, lo, hi):
        import torch
        return self.comm.empty((self.C + 1, hi - lo, self.PV[1], self.PV[2]), torch.float16)

    def pack(self, lo, hi):
        import torch
        t = self.empty(lo, hi)
        shape = tuple(t.shape)
        if self.comm.on_device:
            self._move(lo, hi, _Foreign(t), True)
            self.ctx.sync()
        else:
            st = self._staging(t.numel() * 2)
            self._move(lo, hi, st, True)
            t.view(torch.int16).numpy()[...] = st.download(shape, np.int16)
        return t

    def unpack(self, lo, hi, t):
        import torch
        if self.comm.on_device:
            self._move(lo, hi, _Foreign(t), False)
            self.ctx.sync()
        else:
            st = self._staging(t.numel() * 2)
            st.upload(t.view(torch.int16).numpy())
            self._move(lo, hi, st, False)

    # ---- RCCL transport (boa_hip/rccl.py): the slabs go straight out of / into the accumulator planes, on the C library's
    #      communication stream; nothing here waits, `complete` orders the later kernels behind the exchange ---------------------
    def exchange(self, comm, dst, upper, src, lower, add: bool):
        if not hasattr(comm, "shift_slab"):
            return _exchange_via_tensors(self, comm, dst, upper, src, lower, add)
        if upper is None and lower is None:
            return ("rccl", comm, None, None)
        stage = None
        if add and lower is not None:     # pairwise fp16 sum: the lower block's sums land in a staging buffer and are added in `complete`
            stage = self.ctx.alloc((self.C + 1) * (lower[1] - lower[0]) * self.PV[1] * self.PV[2] * 2)
        comm.shift_slab(dst if upper else None, upper, src if lower else None, lower, self.acc, self.nacc, self.C, self.PV, stage)
        return ("rccl", comm, stage, lower)

    def complete(self, token):
        if not token or token[0] != "rccl":
            return
        _, comm, stage, lower = token
        comm.wait()
        if stage is not None:
            check(self.lib.boa_add_f16_planes(self.ctx.h, self.acc.vp, self.nacc.vp, stage.vp, self.C, int3(self.PV), int(lower[0]), int(lower[1])),
                  "boa_add_f16_planes")
            stage.free()

    def close(self):
        if self._stage is not None:
            self._stage.free()
            self._stage = None


def all_reduce_labels(ctx, comm, buf, n: int):
    """Sum of uint8 label volumes with disjoint supports, in place in the device buffer `buf` (n voxels)."""
    if comm.world == 1:
        return
    if hasattr(comm, "shift_slab"):       # RcclComm: in place, on the communication stream
        comm.all_reduce(buf, n, 0)
        return
    import torch
    if comm.on_device:
        comm.make_room(ctx)
        t = comm.empty((n,), torch.uint8)
        one = (C.c_int * 3)(1, 1, n)
        st = (C.c_longlong * 3)(0, 0, 1)
        f = _Foreign(t)
        check(ctx.lib.boa_copy3(ctx.h, buf.vp, 0, 0, st, one, f.vp, 0, 0, st), "boa_copy3")
        ctx.sync()
        comm.all_reduce_sum(t)
        check(ctx.lib.boa_copy3(ctx.h, f.vp, 0, 0, st, one, buf.vp, 0, 0, st), "boa_copy3")
        ctx.sync()
    else:
        t = torch.from_numpy(buf.download((n,), np.uint8))
        comm.all_reduce_sum(t)
        buf.upload(t.numpy())


def all_reduce_logit_planes(ctx, comm: ShardComm, buf, C_: int, PV, lo: int, hi: int):
    """`buf`: fp16 logits [C][PV0][PV1][PV2] of which this rank holds the planes [lo, hi) of axis 0 (its share of a tile-sharded
    sliding window after the normalisation).  The other planes are cleared and the buffers summed over the ranks (disjoint
    supports: x + 0 is exact in fp16), so every rank ends with the complete logits -- needed when nnU-Net resamples the logits
    before the argmax (export_prediction.py:25-33), which reads across plane ownership."""
    if comm.world == 1:
        return
    plane = int(PV[1]) * int(PV[2])
    vox = int(PV[0]) * plane
    for c in range(C_):
        base = buf.ptr + 2 * c * vox
        if lo > 0:
            check(ctx.lib.boa_memset(ctx.h, C.c_void_p(base), 0, 2 * lo * plane), "boa_memset")
        if hi < PV[0]:
            check(ctx.lib.boa_memset(ctx.h, C.c_void_p(base + 2 * hi * plane), 0, 2 * (int(PV[0]) - hi) * plane), "boa_memset")
    n = C_ * vox
    if hasattr(comm, "shift_slab"):       # RcclComm: fp16 sum in place (x + 0 is exact)
        comm.all_reduce(buf, n, 1)
        return
    import torch
    if comm.on_device:
        # one class plane set at a time: boa_copy3 takes 32-bit extents (25 classes x a 333 x 333 x 1000 volume would wrap), and the
        # staging tensor is one class, not the whole logit volume
        if vox >= 2 ** 31:
            raise ValueError(f"all_reduce_logit_planes: {vox} voxels per class exceed the 32-bit copy extent")
        comm.make_room(ctx)
        t = comm.empty((vox,), torch.float16)
        one = (C.c_int * 3)(1, 1, vox)
        st = (C.c_longlong * 3)(0, 0, 1)
        f = _Foreign(t)
        for c in range(C_):
            check(ctx.lib.boa_copy3(ctx.h, buf.vp, 1, c * vox, st, one, f.vp, 1, 0, st), "boa_copy3")   # dtype 1 = 16-bit words
            ctx.sync()
            comm.all_reduce_sum(t)
            check(ctx.lib.boa_copy3(ctx.h, f.vp, 1, 0, st, one, buf.vp, 1, c * vox, st), "boa_copy3")
            ctx.sync()
    else:
        t = torch.from_numpy(buf.download((n,), np.uint16).view(np.float16).copy())
        comm.all_reduce_sum(t)
        buf.upload(t.numpy().view(np.uint16))


def plane_shares(pv0: int, world: int) -> List[Tuple[int, int]]:
    """The balanced, contiguous plane ranges [P0, P1) of axis 0 on which the ranks finalise (normalise / fold-sum / argmax)."""
    return [((int(pv0) * q) // world, (int(pv0) * (q + 1)) // world) for q in range(world)]


def owner_exchange_lists(owned: List[Tuple[int, int]], shares: List[Tuple[int, int]], rank: int):
    """`owned[p]` = planes rank p holds complete (a partition of [0, pv0), empty ranges allowed), `shares[q]` = planes rank q needs.
    -> (sends, recvs) of `rank` as lists of (peer, lo, hi): what it owns of every other rank's share, what others own of its share."""
    sends, recvs = [], []
    for q, (s0, s1) in enumerate(shares):
        if q != rank:
            lo, hi = max(owned[rank][0], s0), min(owned[rank][1], s1)
            if hi > lo:
                sends.append((q, lo, hi))
    s0, s1 = shares[rank]
    for p, (o0, o1) in enumerate(owned):
        if p != rank:
            lo, hi = max(o0, s0), min(o1, s1)
            if hi > lo:
                recvs.append((p, lo, hi))
    return sends, recvs


def reduce_scatter_logit_planes(ctx, comm, buf, C_: int, PV, owned: List[Tuple[int, int]], shares: Optional[List[Tuple[int, int]]] = None):
    """The reduce-scatter form of all_reduce_logit_planes for consumers that only finalise their own plane share (the fold units of the
    BCA nets, predictor._ShardedJob): `buf` = fp16 logits [C][PV0][PV1][PV2]; rank p holds the planes `owned[p]` complete (every rank
    knows every rank's range: RowPlan.owned_planes is a function of the plan).  Afterwards the planes `shares[rank]` of `buf` are
    complete on this rank; the other planes are unspecified.  Because the supports are disjoint the "sum" is a plane exchange to the
    owner -- every plane crosses one link once, against 2 (N - 1) / N of the whole buffer per link for the ring all-reduce -- and the
    bits arrive untouched (x + 0 would turn -0 into +0)."""
    world = comm.world
    if world == 1:
        return
    shares = shares if shares is not None else plane_shares(int(PV[0]), world)
    sends, recvs = owner_exchange_lists(owned, shares, comm.rank)
    plane = int(PV[1]) * int(PV[2])
    vox = int(PV[0]) * plane
    if hasattr(comm, "planes_to_owner"):          # RcclComm: in place, on the communication stream
        comm.planes_to_owner(buf, C_, PV, sends, recvs)
        return
    import torch
    # gloo / validation transport: one message per (peer, class), posted in the same (class, peer) order on every rank
    dist = comm.dist
    staged = []
    ops = []
    host = None if comm.on_device else buf.download((C_ * vox,), np.uint16).copy()
    for c in range(C_):
        for q, lo, hi in sends:
            n = (hi - lo) * plane
            if comm.on_device:
                t = comm.empty((n,), torch.float16)
                f = _Foreign(t)
                one, st = (C.c_int * 3)(1, 1, n), (C.c_longlong * 3)(0, 0, 1)
                check(ctx.lib.boa_copy3(ctx.h, buf.vp, 1, c * vox + lo * plane, st, one, f.vp, 1, 0, st), "boa_copy3")
            else:
                t = torch.from_numpy(host[c * vox + lo * plane:c * vox + hi * plane].view(np.float16).copy())
            ops.append(dist.P2POp(dist.isend, t, int(q)))
            staged.append(None)
        for p, lo, hi in recvs:
            n = (hi - lo) * plane
            t = comm.empty((n,), torch.float16)
            ops.append(dist.P2POp(dist.irecv, t, int(p)))
            staged.append((c, lo, hi, t))
    if comm.on_device:
        ctx.sync()
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for item in staged:
        if item is None:
            continue
        c, lo, hi, t = item
        n = (hi - lo) * plane
        if comm.on_device:
            f = _Foreign(t)
            one, st = (C.c_int * 3)(1, 1, n), (C.c_longlong * 3)(0, 0, 1)
            check(ctx.lib.boa_copy3(ctx.h, f.vp, 1, 0, st, one, buf.vp, 1, c * vox + lo * plane, st), "boa_copy3")
        else:
            host[c * vox + lo * plane:c * vox + hi * plane] = t.numpy().view(np.uint16)
    if comm.on_device:
        ctx.sync()
    else:
        buf.upload(host)


@dataclass
class TileShard:
    """What `HipPredictor.predict_segmentation_device(..., shard=...)` needs: the transport (ShardComm over torch.distributed, or
    rccl.RcclComm over the C ABI), the exchange mode and -- per model of a multi-model task -- the row assignment of plan_units."""
    comm: object
    mode: str = "exact"
    assignment: Optional[list] = None


def all_reduce_flag(ctx, comm, flag_buf) -> int:
    """Sum of the per-rank inf flags (a device int32) over the ranks -> host int."""
    if comm.world > 1 and hasattr(comm, "shift_slab"):
        comm.all_reduce(flag_buf, 1, 2)
        return int(flag_buf.download((1,), np.int32)[0])
    bad = int(flag_buf.download((1,), np.int32)[0])
    if comm.world == 1:
        return bad
    import torch
    t = torch.tensor([bad], dtype=torch.int32, device=comm.device)
    comm.all_reduce_sum(t)
    return int(t.item())

