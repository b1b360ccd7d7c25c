#!/usr/bin/env python3
"""bench.py -- CT volumes/sec of the BOA hot path on MI355X (BASELINE.json metric).

Workload (configs[1] of BASELINE.json): one 512x512x512 @1.5 mm synthetic CT per GPU, `total` task = the five
part models Dataset291-295 (PlainConvUNet 3d_fullres, patch 128^3, step 0.8 -> 125 tiles each = 625 tile
forwards), Gaussian-weighted fp16 accumulation, normalise + argmax + part->global merge; the int16 CT is
resident in HBM when the timed region starts and the uint8 label volume stays in HBM (PCIe-inclusive rate: see
DESIGN.md).  One "step" = one whole volume.  Multi-GPU: one process per GPU, every rank segments its own volume
(volume-level sharding, no data-path collective) -> weak scaling; value = all volumes / max-over-ranks time.
`--shard tiles` instead lets the N GPUs share every volume (tile rows split across ranks, overlap slabs of the fp16
accumulators over RCCL, boa_hip/tile_shard.py): strong scaling, the latency mode; not the default because whole volumes
are the cheaper way to fill a node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 512] [--batch 4] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense fp16/bf16


def cpu_baseline(geom_tuple, n_tiles_sample, tiles_per_volume, log):
    """Oracle (torch-CPU fp32 PlainConvUNet + numpy fp16 accumulation = the reference's CPU path restated) timed
    on the host on a bounded sample: `n_tiles_sample` tile forwards of part model 291, extrapolated by tile count."""
    import torch
    from oracle import sliding_window as osw
    from oracle.network import build_from_arch
    pj, dj, sd = geom_tuple
    threads = min(8, os.cpu_count() or 1)  # the reference caps torch at min(8, n) (predict_from_raw_data.py:479-480)
    torch.set_num_threads(threads)
    arch = pj["configurations"]["3d_fullres"]["architecture"]["arch_kwargs"]
    nc = len(dj["labels"])
    net = build_from_arch(arch, 1, nc).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    patch = pj["configurations"]["3d_fullres"]["patch_size"]
    g = osw.compute_gaussian(tuple(patch), 1. / 8, 10)
    x = torch.randn(1, 1, *patch)
    acc = np.zeros((nc, *patch), np.float16)
    n = np.zeros(patch, np.float16)
    with torch.inference_mode():
        net(x)  # warm-up (allocator, mkldnn primitives)
        t0 = time.perf_counter()
        for _ in range(n_tiles_sample):
            y = net(x)[0].numpy()
            osw.accumulate_tile(acc, n, y, g, (0, 0, 0))
        dt = (time.perf_counter() - t0) / n_tiles_sample
    log(f"cpu_baseline: {dt:.3f} s per tile forward+accumulate on {threads} threads")
    return {"value": 1.0 / (dt * tiles_per_volume), "unit": "volumes/s", "cores": threads, "kind": "port",
            "sample": f"{n_tiles_sample} x 128^3 tile forward (torch-CPU fp32 PlainConvUNet, 31M params) + fp16 Gaussian "
                      f"accumulation, extrapolated to {tiles_per_volume} tile forwards per volume; argmax/merge not included",
            "s_per_tile": dt, "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, nargs="+", default=[512])
    ap.add_argument("--batch", type=int, default=8, help="tiles per conv-stack launch")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-tiles", type=int, default=4)
    ap.add_argument("--no-prof", action="store_true", help="no per-launch events (measures their overhead; roofline fields become 0)")
    ap.add_argument("--shard", choices=["volumes", "tiles", "models"], default="volumes",
                    help="N>1: 'volumes' = one volume per GPU, no data-path collective (weak scaling, default); 'tiles' = all "
                         "GPUs share each volume: tile rows split, overlap slabs over RCCL (strong scaling, latency mode); "
                         "'models' = all GPUs share each volume: the five part models dealt out to the ranks, label volumes "
                         "all-reduced and merged in part order (strong scaling, bit-identical at any tile batch)")
    ap.add_argument("--shard-mode", choices=["exact", "allreduce"], default="exact",
                    help="--shard tiles: ordered send/recv hand-over (bit-exact) or pairwise fp16 all-reduce of the slabs")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N>1 (nccl = RCCL; gloo lets several ranks share one GPU for validation)")
    ap.add_argument("--with-bca", dest="with_bca", action="store_true", default=True,
                    help="N=1: also time the BCA half of `total+bca` on the same volume (extra field total_plus_bca; default on)")
    ap.add_argument("--no-bca", dest="with_bca", action="store_false")
    ap.add_argument("--dump", type=str, default=None, help="write per-kernel-class timings to this JSON file")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    n_gpus = args.gpus
    dist = None
    import torch
    if world > 1:
        from boa_hip import distributed as D
        if args.backend == "gloo":
            local_rank %= max(torch.cuda.device_count(), 1)
        dist = D.init(args.backend, rank, world, local_rank)  # backend "nccl" is RCCL on ROCm
    log = (lambda *a: print(*a, file=sys.stderr, flush=True)) if rank == 0 else (lambda *a: None)

    from boa_hip import label_maps, synthetic
    from boa_hip._lib import check
    from boa_hip.device import Context
    from boa_hip.predictor import HipPredictor

    shape = args.size * 3 if len(args.size) == 1 else args.size
    ctx = Context(local_rank)
    log("device:", ctx.info())
    t0 = time.perf_counter()
    models = synthetic.total_part_models()
    predictors = []
    for tid, cfg, blob, _ in models:
        p = HipPredictor(ctx, cfg.geometry, tile_step_size=0.8, max_batch=args.batch)  # TS/nnunet.py:507-514
        p.set_parameters([blob])
        p._ensure_net(0)
        predictors.append((tid, cfg, p))
    tile_shard = None
    if dist is not None and args.shard == "tiles":
        from boa_hip import tile_shard as ts
        tile_shard = ts.TileShard(ts.ShardComm(dist, rank, world, f"cuda:{local_rank}" if args.backend == "nccl" else "cpu"), args.shard_mode)
    model_comm = None
    if dist is not None and args.shard == "models":
        from boa_hip import tile_shard as ts
        model_comm = ts.ShardComm(dist, rank, world, f"cuda:{local_rank}" if args.backend == "nccl" else "cpu")
    shared = tile_shard is not None or model_comm is not None          # all ranks work on the same volume
    ct = synthetic.ct_phantom(shape, seed=20260928 + (0 if shared else rank))
    log(f"setup (synthetic weights + phantom {shape}) {time.perf_counter() - t0:.1f}s")
    nvox = int(np.prod(shape))
    d_ct = ctx.from_numpy(ct)
    d_vol = ctx.alloc(nvox * 4)
    d_lab = ctx.alloc(nvox)
    d_part = ctx.alloc(nvox) if args.shard == "models" else None
    work = {}
    ip = models[0][1].intensity_properties["0"]
    tiles_per_volume = 0
    flops_per_volume = 0.0
    from boa_hip import sliding_window as sw
    for tid, cfg, p in predictors:
        PV, _ = sw.pad_amounts(shape, cfg.geometry.patch_size)
        nt = len(sw.get_sliding_window_origins(PV, cfg.geometry.patch_size, 0.8))
        tiles_per_volume += nt
        flops_per_volume += nt * cfg.geometry.flops_per_tile()

    def step():
        # CTNormalization -> 5 part models (sliding window, fp16 accumulate) -> argmax + merge into `total` labels
        check(ctx.lib.boa_ct_normalize(ctx.h, d_ct.vp, 0, d_vol.vp, nvox, ip["mean"], ip["std"], ip["percentile_00_5"],
                                       ip["percentile_99_5"]))
        d_lab.zero()
        if model_comm is not None:
            from boa_hip import tile_shard as ts
            for k, (tid, cfg, p) in enumerate(predictors):
                check(ctx.lib.boa_memset(ctx.h, d_part.vp, 0, nvox))
                if k % world == rank:
                    p.predict_segmentation_device(d_vol, shape, d_part, lut=label_maps.part_lut(tid), merge=False, work=work)
                ts.all_reduce_labels(ctx, model_comm, d_part, nvox)
                check(ctx.lib.boa_label_overlay(ctx.h, d_part.vp, nvox, d_lab.vp))
            return
        for tid, cfg, p in predictors:
            p.predict_segmentation_device(d_vol, shape, d_lab, lut=label_maps.part_lut(tid), merge=True, work=work,
                                          shard=tile_shard)

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier(device_ids=[local_rank]) if args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize() if torch.cuda.is_available() else None

    for _ in range(args.warmup):
        step()
    ctx.sync()
    ctx.prof_reset()
    ctx.prof_enable(not args.no_prof)
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t_start
    ctx.prof_enable(False)
    prof = ctx.prof_get()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    labels_sum = int(d_lab.download((nvox,), np.uint8).astype(np.int64).sum()) if rank == 0 else 0
    if rank == 0:
        conv = prof["conv_mfma"]
        conv_ms = conv["ms"]
        achieved = conv["flops"] / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        total_ms = sum(v["ms"] for v in prof.values())
        for k, v in prof.items():
            if v["launches"]:
                log(f"  {k:16s} {v['ms']:10.1f} ms  {v['launches']:8d} launches  "
                    f"{v['flops'] / max(v['ms'], 1e-9) / 1e9:9.1f} TFLOP/s  {v['bytes'] / max(v['ms'], 1e-9) / 1e6:9.1f} GB/s")
        log(f"  sum of kernel times {total_ms:.1f} ms of {elapsed * 1e3:.1f} ms wall; label checksum {labels_sum}")
        res = {
            "metric": "CT volumes/sec (512^3 @1.5 mm, total) on MI355X",
            "value": (1 if shared else n_gpus) * args.steps / elapsed, "unit": "volumes/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "strong" if shared else "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"configs[1]: single {shape[0]}x{shape[1]}x{shape[2]} @1.5 mm volume, `total` "
                                   f"(5 part models, {tiles_per_volume} tile forwards of 128^3, step 0.8), " +
                                   (f"each volume tile-sharded over {n_gpus} GPUs ({args.shard_mode} slab exchange)" if tile_shard
                                    else f"the part models of each volume dealt out to {n_gpus} GPUs" if model_comm
                                    else "1 volume per GPU"),
                       "tiles_per_volume": tiles_per_volume, "tile_batch": args.batch,
                       "tflop_per_volume": flops_per_volume / 1e12,
                       "precision": "fp16 activations/weights, fp32 MFMA accumulate, fp16 logit accumulators (reference semantics)"},
            "roofline": {"bound": "mfma", "kernel": "k_conv_ws<R,3,3,3> (all MFMA 3x3x3 conv launches: wave-specialised implicit GEMM, v_mfma_f32_32x32x16_f16)",
                         "achieved": achieved, "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_F16_DENSE_PEAK_TFLOPS, "traffic": None,
                         "launches": conv["launches"], "avg_launch_ms": conv_ms / max(conv["launches"], 1),
                         "flops_per_launch": conv["flops"] / max(conv["launches"], 1),
                         "share_of_kernel_time": conv_ms / max(total_ms, 1e-9)},
            "end_to_end_tflops": flops_per_volume * args.steps / elapsed / 1e12,
            # the HBM-bound stages of the same run (algorithmic bytes / event time, peak 8 TB/s): BASELINE.json's "% HBM roofline"
            "hbm_stages": {k: {"achieved_GBps": prof[k]["bytes"] / max(prof[k]["ms"], 1e-9) / 1e6,
                               "frac": prof[k]["bytes"] / max(prof[k]["ms"], 1e-9) / 1e6 / 8000.0, "ms": prof[k]["ms"],
                               "launches": prof[k]["launches"]}
                           for k in ("head_accum", "finalize_argmax", "convT_mfma", "conv_first") if prof[k]["launches"]},
        }
        # HBM traffic of the dominant kernel: FETCH_SIZE / WRITE_SIZE from the committed rocprofv3 --pmc passes
        # (tools/profile_round.sh: separate passes, gfx950 FETCH correction; they cannot run inside this process), averaged over
        # the k_conv_ws launches; per launch like `achieved` (bytes; compare with bytes_per_launch = algorithmic)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_fetch_write_256.json")))["kernels"]
            rows = [v for k, v in pmc.items() if "k_conv_ws" in k]
            nd = sum(v["dispatches"] for v in rows)
            res["roofline"]["traffic"] = 1e6 * 1.048576 * sum(
                v["dispatches"] * (v["fetch_MB_corrected_per_dispatch"] + v["write_MB_per_dispatch"]) for v in rows) / nd
            res["roofline"]["traffic_unit"] = "HBM bytes per launch (PMC, profiles/r01_pmc_fetch_write_256.json)"
            res["roofline"]["bytes_per_launch"] = conv["bytes"] / max(conv["launches"], 1)
        except Exception:
            pass
        if not args.no_cpu and n_gpus == 1:
            res["cpu_baseline"] = cpu_baseline(models[0][3], args.cpu_tiles, tiles_per_volume, log)
        else:
            res["cpu_baseline"] = None
        if args.with_bca and n_gpus == 1:
            # BASELINE.json's metric names `total+bca`: the BCA half (body_parts + body_regions nets, 5 folds each at 5 mm
            # slices, CC / contour-fill post-processing, tissues, per-slice tables, JSON) on the same volume, host to host
            # (uploads the CT, downloads three label volumes), timed once after one warm-up; not part of `value`.
            try:
                from boa_hip import plans
                from boa_hip.pipeline import BcaPipelineHip
                bm = {}
                for name, nc, seed in (("body_parts", 7, 543), ("body_regions", 12, 542)):
                    pj, dj = plans.synthetic_plans(num_classes=nc, spacing=(5.0, 1.5, 1.5))
                    cfg = plans.model_config_from_plans(pj, dj)
                    bm[name] = (cfg, [plans.weight_blob_from_state_dict(cfg.geometry, plans.synthetic_state_dict(cfg.geometry, seed + f))
                                      for f in range(5)])
                pipe = BcaPipelineHip(ctx, bm["body_parts"], bm["body_regions"], fast_bca=False)
                total_lab = d_lab.download(tuple(shape), np.uint8)
                aff = np.diag([-1.5, -1.5, 1.5, 1.0])
                pipe.run(ct, aff, total_seg=total_lab)
                ctx.sync()
                tb = time.perf_counter()
                out = pipe.run(ct, aff, total_seg=total_lab)
                ctx.sync()
                t_bca = time.perf_counter() - tb
                pipe.close()
                t_total = elapsed / args.steps
                res["total_plus_bca"] = {"bca_s": t_bca, "total_s": t_total, "value": 1.0 / (t_total + t_bca), "unit": "volumes/s",
                                         "note": "bca = 2 nets x 5 folds at 5 mm slices + post-processing + tissues + tables, host to "
                                                 "host, one run after one warm-up; synthetic weights",
                                         "tissue_voxels": int((out["tissues"] > 0).sum())}
                log(f"total+bca: total {t_total:.3f} s + bca {t_bca:.3f} s -> {1.0 / (t_total + t_bca):.3f} volumes/s")
            except Exception as e:  # noqa: BLE001  (the extra must never cost the headline line)
                res["total_plus_bca"] = {"error": f"{type(e).__name__}: {e}"}
        if args.dump:
            os.makedirs(os.path.dirname(os.path.abspath(args.dump)), exist_ok=True)
            with open(args.dump, "w") as f:
                json.dump({"prof": prof, "elapsed_s": elapsed, "result": res}, f, indent=1)
        print(json.dumps(res), flush=True)
    for _, _, p in predictors:
        p.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
