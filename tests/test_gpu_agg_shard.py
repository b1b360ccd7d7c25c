"""z-slab sharded aggregation on the device (last row): two / three processes share cuda:0 (gloo: one-GPU box),
each owning a z-slab of the (z,y,x) volumes.  The sharded CC filters of the body_regions post-processing, the tissue pass +
per-slice tables and the per-label HU statistics (histograms all-reduced, erosions on slab + 3-plane halo) must equal the
single-process device results bit for bit (those are pinned to the oracle in test_gpu_aggregation.py / test_gpu_tasks.py)."""
import json
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = (45, 64, 56)   # (z, y, x)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _volumes():
    rng = np.random.default_rng(21)
    Z, Y, X = SHAPE
    zz, yy, xx = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    r = np.sqrt(((yy - Y / 2) / (Y * 0.42)) ** 2 + ((xx - X / 2) / (X * 0.42)) ** 2)
    regions = np.zeros(SHAPE, np.uint8)
    regions[r < 1.0] = 1                                             # subcutaneous shell
    regions[r < 0.8] = 2                                             # muscle
    regions[(r < 0.55) & (zz < 24)] = 3                              # abdominal cavity
    regions[(r < 0.55) & (zz >= 24)] = 4                             # thoracic cavity
    regions[(r < 0.2) & (zz >= 30)] = 9                              # mediastinum
    regions[(r < 0.1) & (zz >= 34) & (zz < 40)] = 7                  # pericardium
    regions[3:7, 2:6, 2:6] = 3                                       # stray islands -> 255
    regions[40:44, 50:60, 40:50] = 4
    regions[20:23, 5:9, 30:34] = 7
    regions[rng.random(SHAPE) > 0.997] = 9
    parts = np.where(r < 1.0, 1, 0).astype(np.uint8)
    parts[:, :8] = 2
    ct = rng.normal(0, 200, size=SHAPE).astype(np.int16)
    ct[regions == 1] = rng.normal(-100, 20, size=int((regions == 1).sum())).astype(np.int16)
    total = rng.integers(0, 118, size=SHAPE).astype(np.uint8)
    total[r >= 1.0] = 0
    total[10:30, 20:40, 10:20] = 1
    return ct, regions, parts, total


def _compute(ctx, agg, total_shard=None):
    from boa_hip import agg_shard as ag
    from boa_hip import bca, label_maps
    from boa_hip import measurements as M
    ct, regions, parts, total = _volumes()
    Z, Y, X = SHAPE
    d_ct, d_rg, d_pt = ctx.from_numpy(ct), ctx.from_numpy(regions), ctx.from_numpy(parts)
    if agg is None:
        bca.postprocess_region_segmentation_device(ctx, d_rg, SHAPE)
        js, tis = bca.bca_measurements_device(ctx, d_ct, d_rg, d_pt, SHAPE, (1.2, 1.2, 3.0), None, True)
    else:
        bca.postprocess_region_segmentation_device_sharded(ctx, agg, d_rg, SHAPE)
        js, tis = bca.bca_measurements_device_sharded(ctx, agg, d_ct, d_rg, d_pt, SHAPE, (1.2, 1.2, 3.0))
    out = {"regions": d_rg.download(SHAPE, np.uint8), "tissues": tis.download(SHAPE, np.uint8), "js": json.dumps(js, default=float, sort_keys=True)}
    # per-label HU statistics of `total`: every rank cuts its slab + halo out of its copy of the volume
    lm = label_maps.measurement_label_map("total")
    if agg is None:
        d_t = ctx.from_numpy(total)
        meas, mask = M.total_measurements(ctx, None, None, lm, (1.2, 1.2, 3.0), d_ct=d_ct, d_lab=d_t, shape=SHAPE)
        d_t.free()
        out["pfav"] = mask
    else:
        comm = agg[0]
        z0, z1 = ag.slab_bounds(Z, comm.world)[comm.rank]
        h0, h1 = max(0, z0 - ag.ERODE_REACH), min(Z, z1 + ag.ERODE_REACH)
        d_c, d_t = ctx.from_numpy(ct[h0:h1]), ctx.from_numpy(total[h0:h1])
        meas, mask = M.total_measurements(ctx, None, None, lm, (1.2, 1.2, 3.0), d_ct=d_c, d_lab=d_t, shape=(h1 - h0, Y, X),
                                          shard=(comm, (z0 - h0, z1 - h0)))
        d_c.free()
        d_t.free()
        out["pfav"] = (z0, z1, mask[z0 - h0:z1 - h0])
    out["meas"] = json.dumps(meas, default=float, sort_keys=True)
    for b in (d_ct, d_rg, d_pt, tis):
        b.free()
    return out


def _worker(rank, world, port, q):
    sys.path[:0] = [HERE, os.path.join(HERE, "body-and-organ-analysis_amd")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from boa_hip import agg_shard as ag
    from boa_hip import distributed as D
    from boa_hip import tile_shard as ts
    from boa_hip.device import Context
    dist = D.init("gloo", rank, world)
    ctx = Context(0)
    out = _compute(ctx, (ag.AggComm(dist, rank, world), ts.ShardComm(dist, rank, world, "cpu")))
    q.put((rank, out))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_aggregation_equals_single_gpu(world):
    import torch.multiprocessing as mp
    from boa_hip.device import Context
    c = Context(0)
    want = _compute(c, None)
    c.close()
    assert (want["regions"] == 255).sum() > 100 and (want["tissues"] > 0).sum() > 1000
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    pf = np.zeros(SHAPE, np.uint8)
    for r in range(world):
        np.testing.assert_array_equal(got[r]["regions"], want["regions"])
        np.testing.assert_array_equal(got[r]["tissues"], want["tissues"])
        assert got[r]["js"] == want["js"]
        assert got[r]["meas"] == want["meas"]
        z0, z1, m = got[r]["pfav"]
        pf[z0:z1] = m
    np.testing.assert_array_equal(pf, want["pfav"])
