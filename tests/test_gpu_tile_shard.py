"""Tile-sharded mode on the device (SURVEY 8e granularity 3): two processes share cuda:0 and one CT; each runs its block of
tile rows, the overlap slab travels through torch.distributed (gloo here -- one GPU box; RCCL when every rank has its own
GPU), each rank finalises its planes and the label volumes are all-reduced.

`exact` mode must give labels bit-identical to the single-process loop (which the other GPU tests pin to the oracle);
`allreduce` mode may flip labels at fp16 near-ties inside the slabs only -- the flip fraction is printed and bounded.

The ranks of a sharded run cut the tile list into different tile batches than one process does; the comparison therefore
also proves that a tile's logits do not depend on the batch it shares a launch with (the InstanceNorm partial sums are
reduced per sample over virtual workgroups whose tile runs are a function of the layer geometry alone, DESIGN.md
"determinism"): the single-process reference runs at tile batch 4, the ranks at tile batch 3.

With two or more GPUs visible the same protocol also runs over RCCL with one GPU per rank (skipped on one-GPU boxes)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = (96, 40, 44)
PARTS = ((801, 6, 1), (802, 4, 2))        # (task id, classes, folds)
RS_SPACING = (1.5, 1.2, 1.3)              # array spacing (z, y, x) that differs from the plans' 1.5 mm: nnU-Net's own resampling runs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


SHAPE_BIG = (256, 224, 320)               # production geometry case: 128^3 patch, six stages (3 x 3 x 4 tiles at step 0.5)


def _models(big=False):
    from boa_hip import plans
    models, luts = [], {}
    for tid, nc, folds in (((801, 25, 1),) if big else PARTS):
        if big:
            pj, dj = plans.synthetic_plans(num_classes=nc, spacing=(1.5, 1.5, 1.5))
        else:
            pj, dj = plans.synthetic_plans(patch=(32, 32, 32), features=(32, 64), num_classes=nc, spacing=(1.5, 1.5, 1.5))
        cfg = plans.model_config_from_plans(pj, dj)
        blobs = [plans.weight_blob_from_state_dict(cfg.geometry, plans.synthetic_state_dict(cfg.geometry, seed=tid + 17 * f))
                 for f in range(folds)]
        models.append((tid, cfg, blobs))
        luts[tid] = np.concatenate([[0], np.arange(1, nc) + 10 * (tid - 800)]).astype(np.uint8)
    return models, luts


def _ct(big=False):
    ct = np.random.default_rng(5).normal(0, 300, size=SHAPE_BIG if big else SHAPE).astype(np.int16)
    ct[ct == 0] = 1
    return ct


def _predict(ctx, shard=None, model_shard=None, max_batch=1, spacing_zyx=None, precision=None, big=False, scatter=False):
    from boa_hip.task import SegmentationTask
    models, luts = _models(big)
    shape = SHAPE_BIG if big else SHAPE
    task = SegmentationTask(ctx, "total", models, resample=1.5, multimodel=True, max_batch=max_batch, part_luts=luts,
                            precision=precision)
    task.model_shard = model_shard
    task.step_size = 0.5
    for _, _, p, _ in task.parts:
        p.tile_step_size = 0.5
    task.shard = shard
    # every path runs the gather head (the unsharded label path, the unsharded resampled path and the tile-sharded path's raw
    # partial sums + deferred planes): the same matrix-core head arithmetic for every tile origin, so "bit-identical" compares
    # like with like also for this geometry's unaligned tile origins (the scatter form falls back to the fp32 VALU head for z
    # origins that are not 8-aligned: tests/test_gpu_gather_head.py)
    if scatter:
        for _, _, p, _ in task.parts:
            p.use_gather_head = False
    d_ct = ctx.from_numpy(_ct(big))
    d_lab = ctx.alloc(int(np.prod(shape)))
    try:
        task.predict_zyx_device(d_ct, shape, d_lab, in_dtype=0, spacing_zyx=spacing_zyx)
        return d_lab.download(shape, np.uint8)
    finally:
        d_ct.free()
        d_lab.free()
        task.close()


def _worker(rank, world, port, mode, q, backend="gloo"):
    sys.path[:0] = [HERE, os.path.join(HERE, "body-and-organ-analysis_amd")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from boa_hip import distributed as D
    from boa_hip import tile_shard as ts
    from boa_hip.device import Context
    if backend == "nccl":           # RCCL: one GPU per rank; the slabs travel through the C ABI's communicator (boa_hip/rccl.py)
        from boa_hip.rccl import RcclComm
        dist = D.init("nccl", rank, world, rank)
        ctx = Context(rank)
        comm = RcclComm(ctx, rank, world)
    else:                           # gloo: the ranks share cuda:0, slabs are staged through the host
        dist = D.init("gloo", rank, world)
        ctx = Context(0)
        comm = ts.ShardComm(dist, rank, world, "cpu")
    sp = RS_SPACING if mode.endswith("+rs") else None    # "+rs": the array is NOT at the plans' spacing (nnU-Net resamples)
    prec = "fp32" if "+f32" in mode else None             # "+f32": the fp32 exact mode of the network
    big = "+big" in mode                                  # "+big": the production geometry (128^3 patch, six stages)
    mode = mode.replace("+rs", "").replace("+f32", "").replace("+big", "")
    if mode == "models":
        lab = _predict(ctx, model_shard=comm, max_batch=4, spacing_zyx=sp, precision=prec, big=big)
    else:
        lab = _predict(ctx, ts.TileShard(comm, mode), max_batch=3, spacing_zyx=sp, precision=prec, big=big)
    q.put((rank, lab))
    if os.environ.get("BOA_TEST_COUNTERS"):     # (second message: which head kernels this rank launched)
        q.put((rank + 1000, ctx.counters()))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


def _run(world, mode, backend="gloo"):
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, mode, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world * (2 if os.environ.get("BOA_TEST_COUNTERS") else 1)))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return got


@pytest.fixture(scope="module")
def single():
    from boa_hip.device import Context
    c = Context(0)
    lab = _predict(c, max_batch=4)
    c.close()
    assert len(np.unique(lab)) > 3
    return lab


@pytest.mark.parametrize("world", [2, 3])
def test_exact_mode_labels_bit_identical(single, world):
    got = _run(world, "exact")
    for r in range(world):
        np.testing.assert_array_equal(got[r], single)       # every rank ends with the full, identical label volume


def test_single_process_labels_do_not_depend_on_the_tile_batch(single):
    from boa_hip.device import Context
    c = Context(0)
    try:
        for mb in (1, 8):
            np.testing.assert_array_equal(_predict(c, max_batch=mb), single)
    finally:
        c.close()


@pytest.mark.parametrize("mode", ["exact+rs", "models+rs"])
def test_sharding_with_plan_spacing_resampling_bit_identical(mode):
    """Tile / model sharding together with nnU-Net's resampling to the plans' spacing (order 3 in, order 1 on the logits back,
    then argmax): the tile-sharded ranks sum their plane-disjoint normalised logits before the fused resize + argmax, the
    model-sharded ranks run their models exactly as one GPU does -> labels bit-identical to the single-process run."""
    from boa_hip.device import Context
    c = Context(0)
    want = _predict(c, max_batch=4, spacing_zyx=RS_SPACING)
    plain = _predict(c, max_batch=4)
    c.close()
    assert (want != plain).mean() > 0.05          # the resampling path really ran
    got = _run(2, mode)
    np.testing.assert_array_equal(got[0], want)
    np.testing.assert_array_equal(got[1], want)


def test_fp32_exact_mode_tile_sharding_bit_identical(single):
    """The network's fp32 exact mode under tile sharding: the deferred planes keep their fp32 channels-last head input."""
    from boa_hip.device import Context
    c = Context(0)
    want = _predict(c, max_batch=4, precision="fp32")
    c.close()
    assert 0 < (want != single).mean() < 0.05      # a different arithmetic, the same segmentation up to near-ties
    got = _run(2, "exact+f32")
    np.testing.assert_array_equal(got[0], want)
    np.testing.assert_array_equal(got[1], want)


@pytest.mark.parametrize("mode", ["exact+big", "allreduce+big"])
def test_tile_sharding_at_the_production_geometry(mode):
    """Two ranks share a 256 x 224 x 320 volume with the 128^3 six-stage part model geometry (36 tiles, the kernels and tile shapes the
    bench runs): the ordered slab hand-over gives the single-process labels bit for bit; the pairwise all-reduce mode may flip labels
    at fp16 near-ties inside the exchanged slabs only."""
    from boa_hip.device import Context
    c = Context(0)
    want = _predict(c, max_batch=4, big=True)
    c.close()
    assert len(np.unique(want)) > 10
    os.environ["BOA_TEST_COUNTERS"] = "1"
    try:
        got = _run(2, mode)
    finally:
        del os.environ["BOA_TEST_COUNTERS"]
    for r in range(2):      # the sharded ranks ran the gather head (raw partial sums + deferred planes), not the per-tile scatter loop
        cnt = got[1000 + r]
        assert cnt["head_gather"] >= (1 + r if mode.startswith("exact") else 1) and cnt["head_mfma"] == 0 and cnt["head_valu"] == 0, cnt
    if mode.startswith("exact"):
        np.testing.assert_array_equal(got[0], want)
        np.testing.assert_array_equal(got[1], want)
    else:
        np.testing.assert_array_equal(got[0], got[1])
        flips = float((got[0] != want).mean())
        print("allreduce mode at the production geometry: label flip fraction", flips)
        assert flips < 2e-3


def _n_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("mode", ["exact", "models"])
def test_rccl_two_gpus_bit_identical(single, mode):
    """One GPU per rank over RCCL / xGMI (the mode bench.py --shard tiles|models uses on a multi-GPU node)."""
    if _n_gpus() < 2:
        pytest.skip("needs >= 2 visible GPUs")
    got = _run(2, mode, backend="nccl")
    np.testing.assert_array_equal(got[0], single)
    np.testing.assert_array_equal(got[1], single)


def test_allreduce_mode_flips_only_near_ties(single):
    got = _run(2, "allreduce")
    np.testing.assert_array_equal(got[0], got[1])
    flips = got[0] != single
    print("allreduce mode: label flips", int(flips.sum()), "of", flips.size)
    from boa_hip import sliding_window as sw
    from boa_hip import tile_shard as ts
    o = sw.get_sliding_window_origins(list(SHAPE), (32, 32, 32), 0.5)
    lo, hi = ts.plan_rows(o, 32, SHAPE[0], 2).boundary(0)
    outside = np.ones(SHAPE[0], dtype=bool)
    outside[lo:hi] = False
    assert not flips[outside].any()
    assert flips.mean() < 2e-3


def test_model_sharding_bit_identical_at_any_batch():
    """SURVEY 8e granularity 2: the part models dealt out to the ranks; each model runs exactly as on one GPU (same tile
    batches), so the merged labels equal the one-GPU result bit for bit also with a tile batch > 1."""
    from boa_hip.device import Context
    c = Context(0)
    want = _predict(c, max_batch=4)
    c.close()
    got = _run(2, "models")
    np.testing.assert_array_equal(got[0], want)
    np.testing.assert_array_equal(got[1], want)


def test_rccl_plumbing_single_rank(single):
    """world_size 1 over the "nccl" (= RCCL) backend: device tensors that alias the engine's buffers, stream hand-over,
    the uint8 label all-reduce.  (Two RCCL ranks cannot share one GPU; the 2-rank protocol is covered over gloo above.)"""
    import torch
    import torch.distributed as dist
    from boa_hip import tile_shard as ts
    from boa_hip.device import Context
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        c = Context(0)
        comm = ts.ShardComm(dist, 0, 1, "cuda:0")
        ref = np.random.default_rng(1).integers(0, 255, size=4096 * 3, dtype=np.uint8)
        buf = c.from_numpy(ref)
        comm.world = 2                      # force the exchange path; a 1-rank sum returns the input
        ts.all_reduce_labels(c, comm, buf, ref.size)
        np.testing.assert_array_equal(buf.download(ref.shape, np.uint8), ref)
        buf.free()
        c.close()
    finally:
        dist.destroy_process_group()


def test_rccl_c_abi_single_rank_self_loop():
    """The C ABI's RCCL transport (csrc/comm.hip) on hardware with one rank: communicator creation from a unique id, the slab
    exchange as a self-loop (planes [6, 9) sent, received into planes [1, 4): in place, and through the staging buffer + the
    fp16 add of the "allreduce" mode), the in-place all-reduce, and the event ordering between the compute stream and the
    communication stream (work queued before / after an exchange sees consistent data without any host synchronisation in
    between).  Two RCCL ranks cannot share one GPU; the multi-rank protocol is covered over gloo and -- when two devices are
    visible -- by test_rccl_two_gpus_bit_identical."""
    import ctypes as C
    from boa_hip._lib import check, int3
    from boa_hip.device import Context
    from boa_hip.rccl import RcclComm
    c = Context(0)
    try:
        comm = RcclComm(c, 0, 1, bcast=lambda b: b)
        assert c.lib.boa_comm_library()
        C_, PV = 3, (10, 6, 8)
        plane = PV[1] * PV[2]
        rng = np.random.default_rng(0)
        acc_h = rng.normal(0, 3, size=(C_, *PV)).astype(np.float16)
        n_h = rng.uniform(0.1, 4, size=PV).astype(np.float16)
        acc, nacc = c.from_numpy(acc_h.view(np.uint16)), c.from_numpy(n_h.view(np.uint16))
        # exact hand-over, in place
        comm.shift_slab(0, (6, 9), 0, (1, 4), acc, nacc, C_, PV)
        comm.wait()
        want_a, want_n = acc_h.copy(), n_h.copy()
        want_a[:, 1:4], want_n[1:4] = acc_h[:, 6:9], n_h[6:9]
        np.testing.assert_array_equal(acc.download((C_, *PV), np.uint16), want_a.view(np.uint16))
        np.testing.assert_array_equal(nacc.download(PV, np.uint16), want_n.view(np.uint16))
        # pairwise sum: stage + add, with compute-stream work queued on both sides of the exchange and no host sync in between
        stage = c.alloc((C_ + 1) * 3 * plane * 2)
        check(c.lib.boa_memset(c.h, C.c_void_p(acc.ptr + 2 * (0 * PV[0] + 7) * plane), 0, 2 * plane))   # class 0, plane 7 := 0 BEFORE the send
        comm.shift_slab(0, (6, 9), 0, (2, 5), acc, nacc, C_, PV, stage)
        comm.wait()
        check(c.lib.boa_add_f16_planes(c.h, acc.vp, nacc.vp, stage.vp, C_, int3(PV), 2, 5))
        src_a, src_n = want_a.copy(), want_n.copy()
        src_a[0, 7] = 0
        exp_a, exp_n = src_a.copy(), src_n.copy()
        exp_a[:, 2:5] = (src_a[:, 2:5].astype(np.float32) + src_a[:, 6:9].astype(np.float32)).astype(np.float16)
        exp_n[2:5] = (src_n[2:5].astype(np.float32) + src_n[6:9].astype(np.float32)).astype(np.float16)
        np.testing.assert_array_equal(acc.download((C_, *PV), np.uint16), exp_a.view(np.uint16))
        np.testing.assert_array_equal(nacc.download(PV, np.uint16), exp_n.view(np.uint16))
        # in-place all-reduce (one rank: identity) on the dtypes the protocol uses
        lab = rng.integers(0, 255, size=5000, dtype=np.uint8)
        d = c.from_numpy(lab)
        comm.all_reduce(d, lab.size, 0)
        np.testing.assert_array_equal(d.download(lab.shape, np.uint8), lab)
        flag = c.from_numpy(np.array([7], np.int32))
        comm.all_reduce(flag, 1, 2)
        assert int(flag.download((1,), np.int32)[0]) == 7
        st = comm.stats()
        assert st["calls"] == 4 and st["bytes_sent_or_reduced"] >= 2 * (C_ + 1) * 3 * plane * 2
        comm.close()
    finally:
        c.close()


@pytest.mark.parametrize("shard", ["tiles", "models"])
def test_bench_shared_volume_modes_run_over_gloo(shard):
    """`bench.py --gpus 2 --shard tiles|models --backend gloo`: the bench's strong-scaling modes end to end (two ranks
    share cuda:0, slabs staged through the host) on a 256^3 volume, `total+bca` incl. the z-slab sharded aggregation: one
    JSON line that counts 2 ranks and the same table summary as the one-rank run."""
    import json
    import subprocess
    common = ["--size", "256", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-parity", "--no-h2h", "--batch", "4"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def run(extra):
        r = subprocess.run([sys.executable, os.path.join(HERE, "bench.py")] + common + extra, env=env, capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    one = run([])
    two = run(["--gpus", "2", "--shard", shard, "--backend", "gloo"])
    assert two["n_gpus"] == 2 and two["config"]["ranks_seen"] == 2 and two["scaling"] == "strong"
    assert one["n_gpus"] == 1 and one["scaling"] == "weak"
    assert two["tables"] == one["tables"]                       # same labels present, same aggregation groups



@pytest.mark.parametrize("shard", ["volumes", "tiles", "models"])
def test_bench_eight_ranks_wiring_over_gloo(shard):
    """`bench.py --gpus 8` end to end with EIGHT ranks (gloo, all sharing cuda:0, one 128^3 volume each / shared): the launcher,
    process-group set-up, barriers, the max-over-ranks reduction, rank counting and the three sharding modes' plumbing at the
    world size of the node the driver's scaling run uses -- so that the first 8-GPU hardware run cannot die on wiring.  (The
    RCCL transport itself needs one GPU per rank: test_rccl_two_gpus_bit_identical.)"""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(HERE, "bench.py"), "--size", "128", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-parity", "--no-h2h",
           "--batch", "2", "--gpus", "8", "--backend", "gloo", "--shard", shard]
    r = subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["config"]["ranks_seen"] == 8
    assert line["scaling"] == ("weak" if shard == "volumes" else "strong")
    assert line["value"] > 0 and line["tables"]["total_labels_present"] > 0
    # whole-job aggregate: 8 volumes per step in the weak mode, one shared volume otherwise
    per_step = line["value"] * line["ms_per_step"] / 1e3
    assert abs(per_step - (8 if shard == "volumes" else 1)) < 1e-6


# ---- (fold, tile row) units: an ensemble on a grid with fewer tile rows than ranks (the BCA nets: 5 folds, 2 rows at 5 mm) --------
FOLD_SHAPE = (40, 44, 48)                 # patch 32, step 0.5 -> 2 tile rows along axis 0


def _fold_task(ctx, folds=3):
    from boa_hip import plans
    from boa_hip.task import SegmentationTask
    pj, dj = plans.synthetic_plans(patch=(32, 32, 32), features=(32, 64), num_classes=7, spacing=(1.5, 1.5, 1.5))
    cfg = plans.model_config_from_plans(pj, dj)
    blobs = [plans.weight_blob_from_state_dict(cfg.geometry, plans.synthetic_state_dict(cfg.geometry, seed=900 + 13 * f)) for f in range(folds)]
    task = SegmentationTask(ctx, "body_parts", [(900, cfg, blobs)], resample=None, multimodel=False, max_batch=3)
    task.step_size = 0.5
    for _, _, p, _ in task.parts:
        p.tile_step_size = 0.5
    return task


def _fold_predict(ctx, shard=None):
    task = _fold_task(ctx)
    task.shard = shard
    ct = np.random.default_rng(6).normal(0, 300, size=FOLD_SHAPE).astype(np.int16)
    ct[ct == 0] = 1
    d_ct = ctx.from_numpy(ct)
    d_lab = ctx.alloc(int(np.prod(FOLD_SHAPE)))
    try:
        task.predict_zyx_device(d_ct, FOLD_SHAPE, d_lab, in_dtype=0)
        return d_lab.download(FOLD_SHAPE, np.uint8)
    finally:
        d_ct.free()
        d_lab.free()
        task.close()


def _fold_worker(rank, world, port, q):
    sys.path[:0] = [HERE, os.path.join(HERE, "body-and-organ-analysis_amd")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from boa_hip import distributed as D
    from boa_hip import tile_shard as ts
    from boa_hip.device import Context
    dist = D.init("gloo", rank, world)
    ctx = Context(0)
    comm = ts.ShardComm(dist, rank, world, "cpu")
    ctx.counters(reset=True)
    lab = _fold_predict(ctx, ts.TileShard(comm, "exact"))
    cnt = ctx.counters()
    q.put((rank, (lab, cnt["conv_ws"] + cnt["first_mfma"])))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 6])
def test_fold_units_keep_every_rank_busy_and_labels_bit_identical(world):
    """Three folds x two tile rows = six (fold, row) units: with tile rows alone two ranks would work, with the units every rank up
    to six runs conv launches; the ordered fp16 fold sum is evaluated on every rank's plane share from the all-reduced (plane-disjoint)
    normalised logits, so the labels equal the one-process run's bit for bit.  Ranks share cuda:0, slabs / logits travel over gloo."""
    import torch.multiprocessing as mp
    from boa_hip import sliding_window as sw
    from boa_hip import tile_shard as ts
    from boa_hip.device import Context
    c = Context(0)
    want = _fold_predict(c)
    c.close()
    assert len(np.unique(want)) > 3
    origins = sw.get_sliding_window_origins(list(FOLD_SHAPE), [32, 32, 32], 0.5)
    assert ts.plan_rows(origins, 32, FOLD_SHAPE[0], world).active == 2          # rows alone: two active ranks
    units = ts.plan_units([2] * 3, world)
    assert sorted({r for blocks in units for r, _ in blocks}) == list(range(world))
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_fold_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(world):
        lab, launches = got[r]
        np.testing.assert_array_equal(lab, want)
        assert launches > 0, f"rank {r} ran no network launch"
