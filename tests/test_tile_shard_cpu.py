"""CPU, gloo, world_size 2 and 3: the tile-sharded protocol (boa_hip/tile_shard.py) driven with a numpy engine built from
the oracle's accumulate step.  `exact` mode must reproduce the single-process fp16 accumulators bit for bit (the
reference's per-voxel `+=` order, NN/inference/predict_from_raw_data.py:611-614); `allreduce` mode differs by one
rounding in the overlap slabs only."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [HERE, os.path.join(HERE, "body-and-organ-analysis_amd")]

PATCH = (32, 24, 16)
HEADS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net(patch):
    """toy network: 3 heads from one channel, with values large enough for fp16 rounding to matter"""
    x = patch[0]
    return np.stack([37.0 * x + 0.3, -11.0 * x * x + 5.0, 3.0 * np.abs(x) - 1.7]).astype(np.float32)


def _volume(shape):
    return np.random.default_rng(7).normal(0, 1.5, size=(1, *shape)).astype(np.float32)


class NumpyEngine:
    def __init__(self, data, origins, patch, heads):
        from oracle import sliding_window as osw
        self.osw, self.data, self.origins, self.patch, self.heads = osw, data, origins, patch, heads
        self.g = osw.compute_gaussian(tuple(patch), 1.0 / 8, 10.0)

    def begin(self):
        self.acc = np.zeros((self.heads, *self.data.shape[1:]), dtype=np.float16)
        self.n = np.zeros(self.data.shape[1:], dtype=np.float16)

    def _pred(self, t):
        s, p = self.origins[t], self.patch
        return _net(self.data[:, s[0]:s[0] + p[0], s[1]:s[1] + p[1], s[2]:s[2] + p[2]])

    def run(self, tiles, defer):
        stash = []
        for t, d in zip(tiles, defer):
            pred, s = self._pred(t), self.origins[t]
            if d:
                stash.append((pred[:, :d].copy(), self.g[:d], (s[0], s[1], s[2])))
            if d < self.patch[0]:
                self.osw.accumulate_tile(self.acc, self.n, pred[:, d:], self.g[d:], (s[0] + d, s[1], s[2]))
        return stash

    def apply(self, stash):
        for pred, g, s in stash:
            self.osw.accumulate_tile(self.acc, self.n, pred, g, s)

    def empty(self, lo, hi):
        import torch
        return torch.empty((self.heads + 1, hi - lo, *self.n.shape[1:]), dtype=torch.float16)

    def pack(self, lo, hi):
        import torch
        return torch.from_numpy(np.concatenate([self.acc[:, lo:hi], self.n[None, lo:hi]]).copy())

    def unpack(self, lo, hi, t):
        a = t.numpy()
        self.acc[:, lo:hi] = a[:-1]
        self.n[lo:hi] = a[-1]


def _single(shape, step):
    from oracle import sliding_window as osw
    data = _volume(shape)
    sl = osw.get_sliding_window_slicers(shape, PATCH, step)
    origins = np.array([[s[0], s[1], s[2]] for s in sl])
    eng = NumpyEngine(data, origins, PATCH, HEADS)
    eng.begin()
    eng.run(range(len(origins)), [0] * len(origins))
    return data, origins, eng.acc, eng.n


def _worker(rank, world, port, shape, step, mode, q):
    sys.path[:0] = [HERE, os.path.join(HERE, "body-and-organ-analysis_amd")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from boa_hip import distributed as D
    from boa_hip import tile_shard as ts
    dist = D.init("gloo", rank, world)
    data, origins, _, _ = (None, None, None, None)
    from oracle import sliding_window as osw
    data = _volume(shape)
    origins = np.array([[s[0], s[1], s[2]] for s in osw.get_sliding_window_slicers(shape, PATCH, step)])
    plan = ts.plan_rows(origins, PATCH[0], shape[0], world)
    eng = NumpyEngine(data, origins, PATCH, HEADS)
    lo, hi = ts.run_fold_sharded(eng, plan, ts.ShardComm(dist, rank, world, "cpu"), mode)
    q.put((rank, lo, hi, eng.acc[:, lo:hi].copy(), eng.n[lo:hi].copy(), plan.active))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, shape, step, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, step, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("world,shape,step", [(2, (80, 40, 36), 0.5), (3, (80, 40, 36), 0.5), (2, (75, 24, 30), 0.8)])
def test_exact_mode_is_bit_identical(world, shape, step):
    _, _, acc, n = _single(shape, step)
    got = _run(world, shape, step, "exact")
    covered = 0
    for rank, lo, hi, a, nn, active in got:
        assert active >= 2
        assert lo == covered or hi == 0
        if hi:
            covered = hi
        np.testing.assert_array_equal(a.view(np.uint16), acc[:, lo:hi].view(np.uint16))
        np.testing.assert_array_equal(nn.view(np.uint16), n[lo:hi].view(np.uint16))
    assert covered == shape[0]


def test_allreduce_mode_differs_only_in_the_slabs():
    from boa_hip import tile_shard as ts
    shape, step, world = (80, 40, 36), 0.5, 2
    _, origins, acc, n = _single(shape, step)
    plan = ts.plan_rows(origins, PATCH[0], shape[0], world)
    slab = plan.boundary(0)
    got = _run(world, shape, step, "allreduce")
    full = np.concatenate([g[3] for g in got], axis=1)
    assert full.shape == acc.shape
    diff = full.view(np.uint16) != acc.view(np.uint16)
    outside = np.ones(shape[0], dtype=bool)
    outside[slab[0]:slab[1]] = False
    assert not diff[:, outside].any()                       # untouched planes are exact
    np.testing.assert_allclose(full.astype(np.float32), acc.astype(np.float32), rtol=2e-3, atol=1e-2)  # one fp16 rounding


def test_plan_rows_properties():
    from boa_hip import sliding_window as sw
    from boa_hip import tile_shard as ts
    for V, P, step in [((512, 512, 512), (128, 128, 128), 0.8), ((768, 512, 512), (128, 128, 128), 0.8),
                       ((232, 100, 100), (128, 128, 128), 0.8), ((300, 128, 128), (128, 128, 128), 0.5),
                       ((100, 50, 50), (128, 128, 128), 0.5), ((1024, 512, 512), (128, 128, 128), 0.8)]:
        PV, _ = sw.pad_amounts(list(V), P)
        o = sw.get_sliding_window_origins(PV, P, step)
        for world in (1, 2, 3, 4, 8):
            plan = ts.plan_rows(o, P[0], PV[0], world)
            assert 1 <= plan.active <= world
            tiles = np.concatenate([plan.tiles(r) for r in range(world)])
            assert tiles.tolist() == list(range(len(o)))                    # every tile once, canonical order kept
            planes = [plan.owned_planes(r) for r in range(plan.active)]
            assert planes[0][0] == 0 and planes[-1][1] == PV[0]
            assert all(planes[i][1] == planes[i + 1][0] for i in range(len(planes) - 1))
            for r in range(1, plan.active - 1):                             # deferred and sent planes are disjoint
                lo_b, up_b = plan.boundary(r - 1), plan.boundary(r)
                if lo_b and up_b:
                    assert lo_b[1] <= up_b[0]
            assert all(plan.owned_planes(r) == (0, 0) and len(plan.tiles(r)) == 0 for r in range(plan.active, world))


class _MailboxComm:
    """In-process stand-in for ShardComm (exact mode only): ranks are run one after the other in ascending order, a rank's
    upward slab is parked until the next rank picks it up.  Legal because a rank's send never depends on what it receives."""

    def __init__(self, world):
        self.world, self.rank, self.box = world, 0, {}

    def shift(self, send, dst, recv, src):
        if send is not None:
            self.box[(self.rank, dst)] = send.clone()
        if recv is not None:
            recv.copy_(self.box.pop((src, self.rank)))


def test_exact_protocol_random_geometries():
    """Many random (volume, patch, step, world) combinations through the protocol with the numpy engine, incl. the small
    volumes whose tile rows two apart still overlap (deferral on more than the first row, fewer active ranks)."""
    from boa_hip import tile_shard as ts
    from oracle import sliding_window as osw
    rng = np.random.default_rng(42)
    seen_multi_defer = seen_reduced = 0
    for case in range(40):
        patch = (int(rng.choice([8, 12, 16])), int(rng.choice([6, 8])), int(rng.choice([6, 8])))
        shape = (int(rng.integers(patch[0], 6 * patch[0])), int(rng.integers(patch[1], 2 * patch[1] + 1)),
                 int(rng.integers(patch[2], 2 * patch[2] + 1)))
        step = float(rng.choice([0.5, 0.8, 1.0]))
        world = int(rng.integers(2, 7))
        data = np.random.default_rng(case).normal(0, 1.5, size=(1, *shape)).astype(np.float32)
        origins = np.array([[s[0], s[1], s[2]] for s in osw.get_sliding_window_slicers(shape, patch, step)])
        single = NumpyEngine(data, origins, patch, HEADS)
        single.begin()
        single.run(range(len(origins)), [0] * len(origins))
        plan = ts.plan_rows(origins, patch[0], shape[0], world)
        seen_reduced += plan.active < min(world, len(plan.rows))
        comm = _MailboxComm(world)
        covered = 0
        for r in range(world):
            comm.rank = r
            eng = NumpyEngine(data, origins, patch, HEADS)
            d = plan.defer_planes(r)
            seen_multi_defer += len(set(plan.row_of_tile[plan.tiles(r)][d > 0])) > 1
            lo, hi = ts.run_fold_sharded(eng, plan, comm, "exact")
            if hi:
                assert lo == covered
                covered = hi
                np.testing.assert_array_equal(eng.acc[:, lo:hi].view(np.uint16), single.acc[:, lo:hi].view(np.uint16))
                np.testing.assert_array_equal(eng.n[lo:hi].view(np.uint16), single.n[lo:hi].view(np.uint16))
        assert covered == shape[0] and not comm.box
    assert seen_reduced > 0          # the plan had to give up ranks at least once


# ------------------------------------------------------------------------------------------------ (row x model) units
def test_plan_units_keeps_eight_ranks_busy_at_512():
    """configs[1] / configs[3] geometry: 512^3, patch 128^3, step 0.8 -> 5 tile rows per part model, 5 part models.  Tile rows alone
    give 5 active ranks of 8 (missing #2); (row x model) units give every rank 3-4 units, every unit exactly once,
    contiguous blocks per model in ascending rank order."""
    from boa_hip import sliding_window as sw
    from boa_hip import tile_shard as ts
    PV, _ = sw.pad_amounts([512, 512, 512], [128, 128, 128])
    o = sw.get_sliding_window_origins(PV, [128, 128, 128], 0.8)
    assert ts.plan_rows(o, 128, PV[0], 8).active == 5
    for world in (1, 2, 3, 4, 5, 8, 16):
        units = ts.plan_units([5] * 5, world)
        load = np.zeros(world, int)
        for blocks in units:
            assert sum(c for _, c in blocks) == 5
            rk = [r for r, _ in blocks]
            assert rk == sorted(set(rk))                              # ascending, each rank one block per model
            plan = ts.plan_rows(o, 128, PV[0], world, assignment=blocks)
            assert plan.ranks == rk and plan.active == len(blocks)
            tiles = np.concatenate([plan.tiles(r) for r in range(world)])
            assert sorted(tiles.tolist()) == list(range(len(o)))
            for r, c in blocks:
                load[r] += c
        assert load.sum() == 25
        if world == 8:
            assert (load > 0).all() and load.max() == 4 and load.min() == 3, load
        if world <= 25:
            assert load.max() - load.min() <= 1, (world, load)
    # weights: models with more tiles per row count more
    units = ts.plan_units([2, 8], 4, weights=[49.0, 25.0])
    assert sum(c for _, c in units[0]) == 2 and sum(c for _, c in units[1]) == 8


def test_unit_sharded_protocol_bit_identical_mailbox():
    """Five models on the 5 x 5 x 5 tile grid of a 512^3 volume (scaled 1/8: 64^3, patch 16^3, step 0.8 -> the same tile starts / 8),
    8 ranks, units from plan_units: per model the ranks of its blocks run the exact protocol; accumulators of every model are
    bit-identical to the single-process loop on the planes each rank owns, and all 8 ranks do work."""
    from boa_hip import tile_shard as ts
    from oracle import sliding_window as osw
    shape, patch, step, world, n_models = (64, 64, 64), (16, 16, 16), 0.8, 8, 5
    origins = np.array([[s[0], s[1], s[2]] for s in osw.get_sliding_window_slicers(shape, patch, step)])
    assert len(origins) == 125 and sorted(set(origins[:, 0])) == [0, 12, 24, 36, 48]
    units = ts.plan_units([5] * n_models, world)
    busy = set()
    for m in range(n_models):
        data = np.random.default_rng(100 + m).normal(0, 1.5, size=(1, *shape)).astype(np.float32)
        single = NumpyEngine(data, origins, patch, HEADS)
        single.begin()
        single.run(range(len(origins)), [0] * len(origins))
        plan = ts.plan_rows(origins, patch[0], shape[0], world, assignment=units[m])
        comm = _MailboxComm(world)
        covered = 0
        for r in range(world):
            comm.rank = r
            eng = NumpyEngine(data, origins, patch, HEADS)
            lo, hi = ts.run_fold_sharded(eng, plan, comm, "exact")
            if hi:
                busy.add(r)
                assert lo == covered
                covered = hi
                np.testing.assert_array_equal(eng.acc[:, lo:hi].view(np.uint16), single.acc[:, lo:hi].view(np.uint16))
                np.testing.assert_array_equal(eng.n[lo:hi].view(np.uint16), single.n[lo:hi].view(np.uint16))
            else:
                assert plan.index(r) is None
        assert covered == shape[0] and not comm.box
    assert busy == set(range(world))


def _unit_worker(rank, world, port, q):
    sys.path[:0] = [HERE, os.path.join(HERE, "body-and-organ-analysis_amd")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from boa_hip import distributed as D
    from boa_hip import tile_shard as ts
    from oracle import sliding_window as osw
    dist = D.init("gloo", rank, world)
    shape, patch, step, n_models = (64, 32, 32), (16, 16, 16), 0.8, 5
    origins = np.array([[s[0], s[1], s[2]] for s in osw.get_sliding_window_slicers(shape, patch, step)])
    units = ts.plan_units([5] * n_models, world)
    comm = ts.ShardComm(dist, rank, world, "cpu")
    out = []
    for m in range(n_models):
        data = np.random.default_rng(100 + m).normal(0, 1.5, size=(1, *shape)).astype(np.float32)
        plan = ts.plan_rows(origins, patch[0], shape[0], world, assignment=units[m])
        eng = NumpyEngine(data, origins, patch, HEADS)
        lo, hi = ts.run_fold_sharded(eng, plan, comm, "exact")
        out.append((m, lo, hi, eng.acc[:, lo:hi].copy(), eng.n[lo:hi].copy()))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_unit_sharded_eight_ranks_over_gloo():
    """The same over real processes: 8 gloo ranks, 5 models x 5 tile rows, neighbour-addressed send / recv between the ranks that
    share a model."""
    from oracle import sliding_window as osw
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_unit_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    shape, patch, step = (64, 32, 32), (16, 16, 16), 0.8
    origins = np.array([[s[0], s[1], s[2]] for s in osw.get_sliding_window_slicers(shape, patch, step)])
    active = set()
    for m in range(5):
        data = np.random.default_rng(100 + m).normal(0, 1.5, size=(1, *shape)).astype(np.float32)
        single = NumpyEngine(data, origins, patch, HEADS)
        single.begin()
        single.run(range(len(origins)), [0] * len(origins))
        covered = 0
        for r in range(world):
            _, lo, hi, a, nn = got[r][m]
            if hi:
                active.add(r)
                assert lo == covered
                covered = hi
                np.testing.assert_array_equal(a.view(np.uint16), single.acc[:, lo:hi].view(np.uint16))
                np.testing.assert_array_equal(nn.view(np.uint16), single.n[lo:hi].view(np.uint16))
        assert covered == shape[0]
    assert active == set(range(world))


def _simulate_rendezvous(calls):
    """`calls[rank]` = the rank's communication-stream program: a list of groups, each a list of ("send" | "recv", peer, tag).  RCCL
    semantics of one ncclGroupStart/End: the group is ONE launch on the rank's communication stream; its point-to-point operations
    progress independently, each meeting its counterpart as soon as the peer's stream is EXECUTING the group that holds the
    counterpart (stream order: a rank's groups run one after the other, a group ends when all of its operations are done).
    Returns the number of rounds until every rank has finished, or raises on a deadlock (a round without progress)."""
    pos = [0] * len(calls)
    done = [[set() for _ in c] for c in calls]
    rounds = 0
    while any(p < len(c) for p, c in zip(pos, calls)):
        progress = False
        for r, c in enumerate(calls):
            if pos[r] >= len(c):
                continue
            for j, (kind, peer, tag) in enumerate(c[pos[r]]):
                if j in done[r][pos[r]] or pos[peer] >= len(calls[peer]):
                    continue
                want = ("recv" if kind == "send" else "send", r, tag)
                grp = calls[peer][pos[peer]]
                if want in grp and grp.index(want) not in done[peer][pos[peer]]:
                    done[r][pos[r]].add(j)
                    done[peer][pos[peer]].add(grp.index(want))
                    progress = True
        for r, c in enumerate(calls):
            while pos[r] < len(c) and len(done[r][pos[r]]) == len(c[pos[r]]):
                pos[r] += 1
                progress = True
        if not progress:
            raise AssertionError(f"deadlock: ranks wait at groups {pos}")
        rounds += 1
    return rounds


@pytest.mark.parametrize("group_msgs", [1, 8, 26, 1 << 30])
def test_grouped_slab_exchange_cannot_deadlock_with_both_neighbours(group_msgs):
    """boa_comm_shift_slab (csrc/comm.hip) sends the (C + 1) planes of a boundary slab as (C + 1) ncclSend / ncclRecv, in sub-groups
    of $BOA_COMM_GROUP messages per direction; a rank in the middle of a model's blocks sends up AND receives from below in the same
    call.  The communication-stream programs of all 8 ranks for one 512^3 `total` volume (5 models x 5 tile rows as (model, row)
    units, models in order, every model's boundaries in one call per rank) are played under rendezvous semantics: every sub-group
    size completes -- the pairs of a sub-group are the same piece indices on both sides of every boundary, and a line of neighbour
    exchanges has no cycle -- while a program that pairs the pieces differently on the two sides is caught by the same simulator."""
    from boa_hip import sliding_window as sw
    from boa_hip import tile_shard as ts
    world, C_ = 8, 25

    def program(PV, plans):
        calls = [[] for _ in range(world)]
        n_b = 0
        for m, plan in enumerate(plans):
            for r in range(world):
                i = plan.index(r)
                if i is None:
                    continue
                up = plan.ranks[i + 1] if plan.boundary(i) is not None else None
                dn = plan.ranks[i - 1] if plan.boundary(i - 1) is not None else None
                if up is None and dn is None:
                    continue
                n_b += up is not None
                for k0 in range(0, C_ + 1, min(group_msgs, C_ + 1)):
                    ks = range(k0, min(C_ + 1, k0 + min(group_msgs, C_ + 1)))
                    grp = [("send", up, (m, k)) for k in ks] if up is not None else []
                    grp += [("recv", dn, (m, k)) for k in ks] if dn is not None else []
                    calls[r].append(grp)
        return calls, n_b

    # (a) 512^3, (model, row) units: 25 units in 8 contiguous runs -> 7 cuts, those inside a model are slab boundaries
    PV = [512, 512, 512]
    origins = np.array(sw.get_sliding_window_origins(PV, [128] * 3, 0.8))
    units = ts.plan_units([5] * 5, world)
    calls_a, nb_a = program(PV, [ts.plan_rows(origins, 128, PV[0], world, assignment=units[m]) for m in range(5)])
    assert nb_a >= 4
    _simulate_rendezvous(calls_a)
    # (b) configs[2]'s 768 planes: 8 tile rows -> all 8 ranks in ONE line per model, the six inner ranks send up and receive from
    #     below in the same call; three models one after the other
    PV = [768, 512, 512]
    origins = np.array(sw.get_sliding_window_origins(PV, [128] * 3, 0.8))
    plan = ts.plan_rows(origins, 128, PV[0], world)
    assert plan.active == 8
    calls, n_boundaries = program(PV, [plan] * 3)
    assert n_boundaries == 21 and all(calls[r] for r in range(world))
    both = sum(1 for c in calls for g in c if any(o[0] == "send" for o in g) and any(o[0] == "recv" for o in g))
    assert both > 0
    _simulate_rendezvous(calls)
    # the simulator does catch a broken pairing: one rank walking its sub-groups in the opposite order
    if min(group_msgs, C_ + 1) < C_ + 1:
        bad = [list(c) for c in calls]
        victim = next(r for r in range(world) if len(bad[r]) > 1)
        bad[victim] = bad[victim][::-1]
        with pytest.raises(AssertionError, match="deadlock"):
            _simulate_rendezvous(bad)


def test_fold_units_keep_eight_ranks_busy_on_the_bca_nets():
    """The BCA half of `total+bca` at 512^3: both nets run at (5 mm, 1.5, 1.5) -> a 154 x 512 x 512 grid, patch 128^3, step 0.5: TWO
    tile rows along axis 0 (7 x 7 tiles each) and five folds.  Tile rows alone give two active ranks; the (fold, row) units of
    predictor._ShardedJob (plan_units over [rows] * folds) give every one of 8 ranks work, each fold's rows stay on at most two ranks
    (one slab boundary per fold), and every (fold, row) unit is owned exactly once."""
    from boa_hip import sliding_window as sw
    from boa_hip import tile_shard as ts
    world, folds = 8, 5
    PV, _ = sw.pad_amounts([154, 512, 512], [128, 128, 128])
    origins = np.array(sw.get_sliding_window_origins(PV, [128] * 3, 0.5))
    rows = sorted(set(int(o[0]) for o in origins))
    assert len(rows) == 2 and len(origins) == 2 * 49
    assert ts.plan_rows(origins, 128, PV[0], world).active == 2
    units = ts.plan_units([len(rows)] * folds, world)
    busy = sorted({r for blocks in units for r, _ in blocks})
    assert busy == list(range(world)), busy
    load = {r: 0 for r in range(world)}
    for f, blocks in enumerate(units):
        assert sum(n for _, n in blocks) == len(rows) and len(blocks) <= 2
        plan = ts.plan_rows(origins, 128, PV[0], world, assignment=blocks)
        seen = np.zeros(len(origins), int)
        for r, _ in blocks:
            seen[plan.tiles(r)] += 1
            load[r] += len(plan.tiles(r))
        assert (seen == 1).all()
        lo_hi = [plan.owned_planes(r) for r, _ in blocks]
        assert lo_hi[0][0] == 0 and lo_hi[-1][1] == PV[0] and all(a[1] == b[0] for a, b in zip(lo_hi, lo_hi[1:]))
    assert max(load.values()) <= 2 * 49 and min(load.values()) >= 49     # 490 tile forwards over 8 ranks: 49 .. 98 each (2 ranks alone: 245)


# ---- reduce-scatter of the plane-disjoint fold logits (round 6) ------------------------------------------------------------------
class _HostBuf:
    """numpy stand-in for a DeviceBuffer on the CPU transport (download / upload of uint16 words)."""

    def __init__(self, words):
        self.a = np.ascontiguousarray(words, dtype=np.uint16)

    def download(self, shape, dtype):
        return self.a.reshape(shape).view(dtype)

    def upload(self, arr):
        self.a[...] = np.asarray(arr).view(np.uint16).reshape(self.a.shape)


def _rs_worker(rank, world, port, C_, PV, owned, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from boa_hip import distributed as D
    from boa_hip import tile_shard as ts
    dist = D.init("gloo", rank, world)
    plane = PV[1] * PV[2]
    full = np.random.default_rng(11).integers(0, 0x7BFF, size=(C_, PV[0], plane), dtype=np.uint16)     # finite fp16 bit patterns
    full[0, :, 0] = 0x8000                                   # -0.0: a sum with +0 would lose the sign, the exchange must not
    mine = np.random.default_rng(100 + rank).integers(0, 0xFFFF, size=full.shape, dtype=np.uint16)      # junk wherever this rank owns nothing
    lo, hi = owned[rank]
    mine[:, lo:hi] = full[:, lo:hi]
    buf = _HostBuf(mine.reshape(-1))
    comm = ts.ShardComm(dist, rank, world, "cpu")
    shares = ts.plane_shares(PV[0], world)
    ts.reduce_scatter_logit_planes(None, comm, buf, C_, PV, owned, shares)
    got = buf.a.reshape(full.shape)
    s0, s1 = shares[rank]
    q.put((rank, bool((got[:, s0:s1] == full[:, s0:s1]).all()), s1 - s0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,owned", [
    (4, [(0, 9), (9, 20), (0, 0), (0, 0)]),                  # a fold on two of four ranks (fold units: the other ranks own nothing)
    (4, [(0, 0), (0, 13), (13, 20), (0, 0)]),
    (3, [(0, 7), (7, 14), (14, 20)]),                        # owners == finalisers up to the balanced cut
    (2, [(0, 20), (0, 0)]),                                  # one owner feeds everybody
])
def test_reduce_scatter_of_fold_logits_is_a_bit_exact_plane_exchange(world, owned):
    """predictor._ShardedJob._begin_fold_units: a fold's normalised logits are complete on the planes its member ranks own; every rank
    finalises the balanced share plane_shares()[rank].  After reduce_scatter_logit_planes every rank holds its share bit for bit
    (incl. -0.0, which the all-reduce's x + 0 would flip) and the bytes that moved are each plane once."""
    from boa_hip import tile_shard as ts
    C_, PV = 3, (20, 5, 4)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rs_worker, args=(r, world, port, C_, PV, owned, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in got) and sum(n for _, _, n in got) == PV[0]
    # traffic: every plane that is not already at its finaliser travels once
    shares = ts.plane_shares(PV[0], world)
    moved = 0
    for r in range(world):
        s, rcv = ts.owner_exchange_lists(owned, shares, r)
        moved += sum(hi - lo for _, lo, hi in s)
        assert sum(hi - lo for _, lo, hi in rcv) + max(0, min(owned[r][1], shares[r][1]) - max(owned[r][0], shares[r][0])) == shares[r][1] - shares[r][0]
    stay = sum(max(0, min(owned[r][1], shares[r][1]) - max(owned[r][0], shares[r][0])) for r in range(world))
    assert moved == PV[0] - stay


def test_reduce_scatter_lists_match_across_ranks_at_the_bca_geometry():
    """Send list of p to q == receive list of q from p for the (fold, row) units of the 512^3 BCA nets on 8 ranks, and the per-link bytes
    are at most half of the ring all-reduce's 2 (N - 1) / N of the whole buffer."""
    from boa_hip import sliding_window as sw
    from boa_hip import tile_shard as ts
    world, folds = 8, 5
    PV, _ = sw.pad_amounts([154, 512, 512], [128, 128, 128])
    origins = np.array(sw.get_sliding_window_origins(PV, [128] * 3, 0.5))
    rows = sorted(set(int(o[0]) for o in origins))
    units = ts.plan_units([len(rows)] * folds, world)
    shares = ts.plane_shares(PV[0], world)
    for blocks in units:
        plan = ts.plan_rows(origins, 128, PV[0], world, assignment=blocks)
        owned = [plan.owned_planes(r) for r in range(world)]
        lists = [ts.owner_exchange_lists(owned, shares, r) for r in range(world)]
        for p in range(world):
            for q_, lo, hi in lists[p][0]:
                assert (p, lo, hi) in lists[q_][1]
        assert sum(len(s) for s, _ in lists) == sum(len(r) for _, r in lists)
        out_planes = max(sum(hi - lo for _, lo, hi in s) for s, _ in lists)
        assert out_planes <= PV[0]                                   # a rank sends at most the whole buffer once ...
        ring = 2 * (world - 1) / world * PV[0]
        in_planes = max(sum(hi - lo for _, lo, hi in r) for _, r in lists)
        assert in_planes <= ring / 2 / (world - 1) * world           # ... and receives no more than its own share: 1 / N of it
