"""CPU, world_size 2, gloo: the N>1 path of the bench / batch driver (volume sharding, barrier, max-reduce of the
timed region, gather of per-rank result tables)."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_volumes, q):
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [here, os.path.join(here, "body-and-organ-analysis_amd")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from boa_hip import distributed as D
    from boa_hip import synthetic
    dist = D.init("gloo", rank, world)
    mine = D.shard_volumes(n_volumes, rank, world)
    # each rank "processes" its own synthetic volumes (seeded per volume index, like bench.py) -> small result table
    table = {v: int(synthetic.ct_phantom((8, 8, 8), seed=20260928 + v).astype(np.int64).sum()) for v in mine}
    dist.barrier()
    elapsed = 0.25 * (rank + 1)
    tmax = D.reduce_max(dist, elapsed)
    allt = D.gather_objects(dist, table, world)
    if rank == 0:
        q.put((tmax, allt))
    dist.barrier()
    dist.destroy_process_group()


def test_volume_sharding_two_ranks_gloo():
    world, n_volumes = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_volumes, q)) for r in range(world)]
    for p in procs:
        p.start()
    tmax, tables = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert tmax == 0.5  # max over ranks
    merged = {}
    for t in tables:
        assert not (set(t) & set(merged))  # disjoint shards
        merged.update(t)
    assert sorted(merged) == list(range(n_volumes))  # every volume exactly once
    from boa_hip import synthetic
    for v, s in merged.items():
        assert s == int(synthetic.ct_phantom((8, 8, 8), seed=20260928 + v).astype(np.int64).sum())


def test_shard_volumes_properties():
    from boa_hip.distributed import shard_volumes
    for n in (0, 1, 7, 8, 9, 64):
        for w in (1, 2, 3, 8):
            parts = [shard_volumes(n, r, w) for r in range(w)]
            flat = [v for p in parts for v in p]
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
