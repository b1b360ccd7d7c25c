#!/usr/bin/env python3
"""bench.py -- CT volumes/sec of the BOA hot path on MI355X (BASELINE.json metric: 512^3 @1.5 mm, `total+bca`).

One "step" = one whole CT through the hot path of `body_organ_analysis --models total+bca`:
    total   : canonicalise -> CTNormalization -> 5 part models (PlainConvUNet 3d_fullres, patch 128^3, step 0.8 = 125 tiles
              each) -> Gaussian-weighted fp16 accumulation -> normalise + argmax + part merge -> `total` labels
    measure : per-label HU statistics of the 117 `total` labels (one histogram pass + eroded / fat-window masks)
    bca     : body_parts + body_regions nets (5 folds each, 5 mm slices), CC / contour-fill post-processing, tissue map,
              per-slice tables, bca-measurements JSON
on a synthetic 512x512x512 @1.5 mm CT with seeded random weights of the documented architectures (no weights offline).
`value` = volumes / (max-over-ranks wall time of the K timed steps), with the int16 CT resident in HBM when the timed
region starts and the label volumes left in HBM (the tables come back to the host); `host_to_host` in the same line is the
PCIe-inclusive rate (CT uploaded, three label volumes + `total` labels downloaded inside the timed region) -- never `value`.

Multi-GPU: one process per GPU over RCCL.  `python bench.py --gpus N` launches the N ranks itself (torch.distributed.run on
127.0.0.1) when it is not already running under a launcher, and refuses to print a line when WORLD_SIZE != N.
Default sharding: whole volumes (every rank segments its own CT, no data-path collective) -> weak scaling.
`--shard tiles|models` lets the N GPUs share every volume (strong scaling; tile rows split with an RCCL exchange of the
overlap slabs of the fp16 accumulators, or the part models dealt out to the ranks).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--models total+bca|total] [--batch 8] [--no-cpu]
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "body-and-organ-analysis_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense fp16/bf16
HBM_PEAK_GBPS = 8000.0               # same guide: 8 TB/s spec (6.3 TB/s measured for a float4 copy)
HBM_COPY_GBPS = 6300.0
# tools/write_bw.hip on the bench's MI355X (round 5): pure 16-byte-per-lane WRITE stream 4.0 - 5.5 TB/s by grid (4.7 at 4 workgroups per
# CU), pure read 6.37, copy 4.6 - 5.3: what the write-dominated kernels (first conv: 94 % of its bytes are stores, transposed conv: 89 %)
# can be priced against besides the 8 TB/s data-sheet figure
HBM_WRITE_GBPS = 4700.0
PMC_PROFILE = os.path.join(ROOT, "profiles", "r06_pmc_fetch_write_512.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, nargs="+", default=[512])
    ap.add_argument("--batch", type=int, default=25,
                    help="tiles per conv-stack launch (25: the 125 tiles of a 512^3 part model in 5 launches; ~1 GB of activations per tile and net)")
    ap.add_argument("--models", choices=["total+bca", "total"], default="total+bca",
                    help="total+bca = BASELINE.json's metric (default); total = the five part models only (configs[1])")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-h2h", action="store_true", help="skip the host-to-host (PCIe-inclusive) extra")
    ap.add_argument("--no-parity", action="store_true", help="skip the fp16-vs-exact-mode label flip sample")
    ap.add_argument("--no-exact", action="store_true", help="skip the measured whole-volume run of the fp32 (split-precision) mode")
    ap.add_argument("--exact-batch", type=int, default=25,
                    help="tile batch of the fp32-mode volume (2 GB of fp32 activations per tile: one 50 GB arena shared by the context's networks)")
    ap.add_argument("--cpu-tiles", type=int, default=8)
    ap.add_argument("--cpu-all-cores", dest="cpu_all_cores", action="store_true", default=True,
                    help="cpu_baseline: also time ONE tile forward on every host core (BASELINE.md 4.3's second setting; ~30 s on an "
                         "oversubscribed host) -> cpu_baseline.all_cores; on by default since round 6")
    ap.add_argument("--no-cpu-all-cores", dest="cpu_all_cores", action="store_false", help="skip the all-cores leg of cpu_baseline")
    ap.add_argument("--lanes", type=int, default=2, choices=[2, 3],
                    help="streams of the overlap extra: 2 = `total` | both BCA nets, 3 = `total` | body_parts | body_regions")
    ap.add_argument("--no-c3", action="store_true", help="skip the configs[2] extra (one 512x512x768 volume, total+bca)")
    ap.add_argument("--no-lanes", action="store_true", help="skip the two-lane extra (`total` and the BCA nets on two streams)")
    ap.add_argument("--no-phantom", action="store_true", help="skip the structured-phantom timings of the aggregation / morphology stages")
    ap.add_argument("--no-prof", action="store_true", help="no per-launch events (measures their overhead; roofline fields become 0)")
    ap.add_argument("--shard", choices=["volumes", "tiles", "models"], default="volumes",
                    help="N>1: 'volumes' = one volume per GPU, no data-path collective (weak scaling, default); 'tiles' = all "
                         "GPUs share each volume: tile rows split, overlap slabs over RCCL (strong scaling, latency mode); "
                         "'models' = the part models of `total` dealt out to the ranks, label volumes all-reduced")
    ap.add_argument("--shard-mode", choices=["exact", "allreduce"], default="exact",
                    help="--shard tiles: ordered send/recv hand-over (bit-exact) or pairwise fp16 all-reduce of the slabs")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N>1 (nccl = RCCL; gloo lets several ranks share one GPU for validation)")
    ap.add_argument("--dump", type=str, default=None, help="write per-kernel-class timings to this JSON file")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks (one per GPU) and relay their exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: launching", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(part_model, n_tiles_sample, work, log, device_tiles=None):
    """The oracle (= the reference's CPU path restated: torch-CPU fp32 PlainConvUNet, numpy fp16 accumulation, numpy argmax /
    merge, numpy tissue + HU statistics) timed on the host on a BOUNDED sample and extrapolated by unit counts:
      network + accumulate : `n_tiles_sample` 128^3 tile forwards              x tile forwards per volume
      normalise + argmax   : one 25-class 128^3 block                           x (sum of classes x voxels) per volume
      part merge           : the reference's 117 compare-and-assign passes on 128^3   x voxels
      aggregation          : tissue map + per-slice tables + per-label HU statistics on 128^3 x voxels
    `device_tiles` = (volume [1,X,Y,Z] fp32, origins, fn(precision, origin) -> fp32 logits [C,128,128,128] of that tile on the
    GPU): the CPU tile outputs are then compared with the device's (both modes) before they are dropped -> `vs_oracle`."""
    import torch
    from oracle import bca as obca
    from oracle import labels as olab
    from oracle import measurements as omeas
    from oracle import sliding_window as osw
    from oracle.network import build_from_arch
    pj, dj, sd = part_model
    threads = min(8, os.cpu_count() or 1)  # the reference caps torch at min(8, n) (predict_from_raw_data.py:479-480)
    torch.set_num_threads(threads)
    arch = pj["configurations"]["3d_fullres"]["architecture"]["arch_kwargs"]
    nc = len(dj["labels"])
    net = build_from_arch(arch, 1, nc).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    patch = pj["configurations"]["3d_fullres"]["patch_size"]
    g = osw.compute_gaussian(tuple(patch), 1. / 8, 10)
    acc = np.zeros((nc, *patch), np.float16)
    n = np.zeros(patch, np.float16)
    if device_tiles is not None:
        vol, origins, dev_fn = device_tiles
        origins = origins[:n_tiles_sample]
        tiles = [torch.from_numpy(np.ascontiguousarray(vol[None, :, o[0]:o[0] + patch[0], o[1]:o[1] + patch[1], o[2]:o[2] + patch[2]]))
                 for o in origins]
    else:
        tiles = [torch.randn(1, 1, *patch) for _ in range(n_tiles_sample)]
    MARGIN_THRESHOLDS = (0.001, 0.003, 0.01)
    vs = {p_: {"max_err_over_range": 0.0, "flips": 0, "max_flip_margin": 0.0, "flips_margin_above": {t: 0 for t in MARGIN_THRESHOLDS}}
          for p_ in ("fp16", "fp32")}
    vs.update({"voxels": 0, "margin_above": {t: 0 for t in MARGIN_THRESHOLDS}})
    t_cpu = 0.0
    with torch.inference_mode():
        net(tiles[0])  # warm-up (allocator, mkldnn primitives)
        for i, x in enumerate(tiles):
            t0 = time.perf_counter()
            y = net(x)[0].numpy()
            osw.accumulate_tile(acc, n, y, g, (0, 0, 0))
            t_cpu += time.perf_counter() - t0
            if device_tiles is not None:      # (untimed) the same tile on the device, both precisions
                rng_ = float(y.max() - y.min())
                lab = y.argmax(0)
                top2 = np.partition(y, -2, axis=0)[-2:]
                margin = (top2[1] - top2[0]) / rng_          # the oracle's winner over its runner-up, in units of the logit range
                for thr in MARGIN_THRESHOLDS:
                    vs["margin_above"][thr] += int((margin > thr).sum())
                for prec in ("fp16", "fp32"):
                    d = dev_fn(prec, origins[i])
                    vs[prec]["max_err_over_range"] = max(vs[prec]["max_err_over_range"], float(np.abs(d - y).max()) / rng_)
                    flipped = d.argmax(0) != lab
                    vs[prec]["flips"] += int(flipped.sum())
                    if flipped.any():
                        vs[prec]["max_flip_margin"] = max(vs[prec]["max_flip_margin"], float(margin[flipped].max()))
                        for thr in MARGIN_THRESHOLDS:
                            vs[prec]["flips_margin_above"][thr] += int((margin[flipped] > thr).sum())
                vs["voxels"] += int(lab.size)
        t_tile = t_cpu / len(tiles)
        # the same forward with every host core (SURVEY 8d asks for both settings): 2 tiles
        allc = os.cpu_count() or 1
        t_tile_all = None
        if allc > threads and work.get("all_cores"):
            torch.set_num_threads(allc)
            t0 = time.perf_counter()
            net(tiles[0])                       # one forward (oversubscribed hosts take ~10x the 8-thread time: keep the leg bounded)
            t_tile_all = time.perf_counter() - t0
            torch.set_num_threads(threads)
    vs_oracle = None
    if device_tiles is not None:
        vs_oracle = {"tiles": len(tiles), "voxels": vs["voxels"],
                     "fp16_max_logit_err_over_range": vs["fp16"]["max_err_over_range"], "fp16_label_flip_fraction": vs["fp16"]["flips"] / vs["voxels"],
                     "exact_max_logit_err_over_range": vs["fp32"]["max_err_over_range"], "exact_label_flip_fraction": vs["fp32"]["flips"] / vs["voxels"],
                     # where the flips are: only voxels whose two best oracle logits are closer than `max_flip_margin` (in units of the
                     # logit range) change label -- the synthetic net's random head leaves most voxels that close; a trained net's
                     # confident voxels (margin above a few 1e-3 of the range) cannot flip at these error levels
                     "fp16_flip_margin": {"max_margin_over_range_of_a_flipped_voxel": vs["fp16"]["max_flip_margin"],
                                          "flips_with_margin_above": {str(t): vs["fp16"]["flips_margin_above"][t] for t in MARGIN_THRESHOLDS},
                                          "voxels_with_margin_above": {str(t): vs["margin_above"][t] for t in MARGIN_THRESHOLDS}},
                     "exact_flip_margin": {"max_margin_over_range_of_a_flipped_voxel": vs["fp32"]["max_flip_margin"]},
                     "sample": f"the {len(tiles)} 128^3 tiles (step 0.8) of a 160x160x192 phantom crop, part model 291: per-tile logits of the device "
                               "(fp16 production mode and fp32 exact mode) against the torch-CPU fp32 oracle outputs of the cpu_baseline leg"}
        log(f"parity vs oracle on {len(tiles)} tiles: fp16 max|err| {vs_oracle['fp16_max_logit_err_over_range']:.3g} of the range, flips "
            f"{vs_oracle['fp16_label_flip_fraction']:.3g}; exact mode max|err| {vs_oracle['exact_max_logit_err_over_range']:.3g}, flips "
            f"{vs_oracle['exact_label_flip_fraction']:.3g}")
    pv = int(np.prod(patch))
    t0 = time.perf_counter()
    logits = osw.finalize_logits(acc, n)
    seg = olab.argmax_labels(logits)
    t_argmax = (time.perf_counter() - t0) / (nc * pv)          # per (class, voxel)
    from boa_hip import label_maps
    t0 = time.perf_counter()
    olab.merge_parts([seg] * 5, [label_maps.CLASS_MAP_PARTS[t] for t in label_maps.PART_TASK_IDS], label_maps.CLASS_MAP_TOTAL_INV)
    t_merge = (time.perf_counter() - t0) / pv                  # per voxel (all five parts)
    # aggregation / post-processing / resampling stages on slabs with the volume's full in-plane extent (a quarter of its slices
    # for the voxel passes, half of the 5 mm grid for the connected-component stages), blocky label phantoms (32^3 blocks) so that
    # the statistics and the labelling see regions, not noise; extrapolated by voxel counts
    from oracle import resample as ores
    rng = np.random.default_rng(0)
    vx, vy, vz = work["shape"]

    def blocky(n, sh, b=32):
        small = rng.integers(0, n, size=[(v + b - 1) // b for v in sh]).astype(np.uint8)
        return np.kron(small, np.ones((b, b, b), np.uint8))[:sh[0], :sh[1], :sh[2]]

    slab = (max(vz // 4, 32), vy, vx)                          # SimpleITK order (z, y, x), a quarter of the slices
    ct = rng.integers(-1000, 1500, size=slab).astype(np.int16)
    regions, parts, total = blocky(12, slab), blocky(7, slab), blocky(118, slab)
    t0 = time.perf_counter()
    tis = obca.subclassify_tissues(ct, regions)
    obca.slicewise_measurements(tis, parts, (1.5, 1.5, 1.5))
    omeas.metrics_for_each_region(ct, total, {f"l{i}": i for i in range(1, 118)}, None, None, (1.5, 1.5, 1.5))
    t_agg = (time.perf_counter() - t0) / float(np.prod(slab))  # per voxel
    s_post = s_res = 0.0
    if work["with_bca"]:
        z5 = int(round(vz * 1.5 / 5.0))
        half5 = (max(z5 // 2, 8), vy, vx)
        t0 = time.perf_counter()
        obca.postprocess_region_segmentation(blocky(12, half5))     # four 26-connected labelings (BCA/body_regions/postprocess.py)
        obca.remove_small_labeled_objects(blocky(7, half5))         # slice-wise contour fill + small objects / holes (body_parts)
        s_post = (time.perf_counter() - t0) * (z5 * vy * vx) / float(np.prod(half5))
        t0 = time.perf_counter()
        small, _ = ores.change_spacing_array(ct, (1.5, 1.5, 1.5), (5.0, 1.5, 1.5), order=3)       # cubic to the 5 mm grid (TS/resampling.py)
        ores.change_spacing_array(regions[:small.shape[0]], (5.0, 1.5, 1.5), target_shape=slab, order=0)  # nearest back
        s_res = (time.perf_counter() - t0) * work["voxels"] / float(np.prod(slab)) * 1.5   # one cubic in, two label volumes back
    s_net = t_tile * work["tile_forwards"]
    s_arg = t_argmax * work["class_voxels"]
    s_merge = t_merge * work["voxels"]
    s_agg = t_agg * work["voxels"] * (1.0 if work["with_bca"] else 0.5)
    total_s = s_net + s_arg + s_merge + s_agg + s_post + s_res
    log(f"cpu_baseline: {t_tile:.3f} s per tile forward+accumulate on {threads} threads"
        + (f" ({t_tile_all:.3f} s per forward on all {allc} cores)" if t_tile_all else "")
        + f"; per volume: nets {s_net:.0f} s, normalise+argmax {s_arg:.0f} s, merge {s_merge:.0f} s, aggregation {s_agg:.0f} s, "
          f"CC / contour-fill post-processing {s_post:.0f} s, resampling {s_res:.0f} s")
    out = {"value": 1.0 / total_s, "unit": "volumes/s", "cores": threads, "kind": "port",
           "sample": f"{len(tiles)} x 128^3 tile forward (torch-CPU fp32 PlainConvUNet, 31M params, {threads} threads = the reference's own cap, "
                     f"predict_from_raw_data.py:479-480) + fp16 Gaussian "
                     f"accumulation; normalise + argmax and the reference's 117-pass part merge on one 128^3 block; tissue map + slice tables "
                     f"+ per-label HU statistics on a {slab[2]}x{slab[1]}x{slab[0]} slab (a quarter of the slices); 26-connected labelings + slice-wise "
                     f"contour fill on half of the 5 mm grid; cubic resampling to 5 mm + nearest back on the slab (numpy / scipy, 1 thread); "
                     f"extrapolated by unit counts to {work['tile_forwards']} tile forwards, {work['class_voxels']:.3g} class-voxels, "
                     f"{work['voxels']:.3g} voxels per volume",
           "s_per_tile": t_tile, "s_per_volume": {"nets": s_net, "argmax": s_arg, "merge": s_merge, "aggregation": s_agg,
                                                  "cc_postprocessing": s_post, "resampling": s_res},
           "host_cpus": os.cpu_count()}
    if t_tile_all:
        s_all = t_tile_all * work["tile_forwards"] + s_arg + s_merge + s_agg + s_post + s_res
        out["all_cores"] = {"cores": allc, "s_per_tile_forward": t_tile_all, "value": 1.0 / s_all,
                            "note": "same extrapolation with the network forward on every host core (1 tile timed); the numpy stages are single-threaded either way"}
    return out, vs_oracle


def parity_sample(ctx, part_model_cfg, blob, batch, log, tile_forwards, step_s):
    """fp16 production mode against the fp32 exact mode (= the reference's CPU arithmetic, tests/test_gpu_exact_mode.py) on a
    bounded sample: label flip fraction of part model 291 on a 160x160x192 crop of the phantom (8 tiles, step 0.8), and the time
    of both modes on that sample (-> an extrapolated volumes/s for the exact mode).  Returns (dict, device_tiles, close):
    device_tiles feeds cpu_baseline's per-tile comparison with the oracle."""
    from boa_hip import sliding_window as sw
    from boa_hip import synthetic
    from boa_hip.predictor import HipPredictor
    ct = synthetic.ct_phantom([160, 160, 192], seed=7).astype(np.float32)
    ip = part_model_cfg.intensity_properties["0"]
    x = ((np.clip(ct, ip["percentile_00_5"], ip["percentile_99_5"]) - ip["mean"]) / max(ip["std"], 1e-8)).astype(np.float32)[None]
    origins = np.array(sw.get_sliding_window_origins([160, 160, 192], part_model_cfg.geometry.patch_size, 0.8), dtype=np.int32)
    labs, preds, t_mode = {}, {}, {}
    for prec in ("fp16", "fp32"):
        p = HipPredictor(ctx, part_model_cfg.geometry, tile_step_size=0.8, max_batch=min(batch, 8), precision=prec)
        p.set_parameters([blob])
        labs[prec] = p.predict_segmentation(x)          # (also the warm-up of this mode)
        ctx.sync()
        t0 = time.perf_counter()
        p.predict_segmentation(x)
        ctx.sync()
        t_mode[prec] = (time.perf_counter() - t0) / len(origins)
        preds[prec] = p
    flips = float((labs["fp16"] != labs["fp32"]).mean())
    log(f"parity sample: fp16 vs exact-mode label flip fraction {flips:.3g} on {labs['fp16'].size} voxels; per tile {t_mode['fp16'] * 1e3:.2f} ms (fp16) / "
        f"{t_mode['fp32'] * 1e3:.2f} ms (exact)")
    out = {"fp16_vs_exact_mode_label_flip_fraction": flips, "voxels": int(labs["fp16"].size),
           "sample": "part model 291 (synthetic weights) on a 160x160x192 phantom crop, 8 tiles, step 0.8; exact mode = fp32 "
                     "weights/activations/accumulation in split precision on the matrix cores (net_x3.hip, k_conv_ws<X3>)",
           "exact_mode_sample": {"ms_per_tile": t_mode["fp32"] * 1e3, "production_ms_per_tile_same_sample": t_mode["fp16"] * 1e3,
                                 "note": "8-tile sample (sliding window incl. head + argmax); the whole-volume number is parity.exact_mode"}}

    def dev_fn(prec, origin):
        return preds[prec].network_forward(x, np.asarray([origin], dtype=np.int32))[0]

    def close():
        for p in preds.values():
            p.close()

    return out, (x, origins, dev_fn), close


def phantom_stages(ctx, shape, ct, log, reps=3):
    """The scan stages of the aggregation half on STRUCTURED label volumes ("a structured phantom (nested ellipsoids ->
    117 labels) so histograms are non-degenerate"; boa_hip/synthetic.py: 117 organs, 6 body parts, 11 nested body regions) at the
    benchmarked size: what these stages cost on labels with the topology of a real segmentation -- the argmax of the random-weight
    nets that the timed region feeds them is noise-like, the worst case of every component filter.  Per stage: median of
    `reps` repetitions of the EVENT-timed kernel time (every launch of the stage bracketed by HIP events on its stream), ALGORITHMIC bytes (what the stage has to read and write once, stated per stage) and their
    fraction of the 8 TB/s HBM peak.  Outside the timed region."""
    from boa_hip import bca, synthetic
    from boa_hip import measurements as M
    X, Y, Z = shape
    zyx = (Z, Y, X)
    n = X * Y * Z
    t0 = time.perf_counter()
    lab = {k: np.ascontiguousarray(f(shape).transpose(2, 1, 0)) for k, f in (("total", synthetic.label_phantom_total),
                                                                           ("parts", synthetic.label_phantom_parts),
                                                                           ("regions", synthetic.label_phantom_regions))}
    d_ct = ctx.from_numpy(np.ascontiguousarray(ct.transpose(2, 1, 0)))      # (z, y, x) int16
    d = {k: ctx.from_numpy(v) for k, v in lab.items()}
    log(f"phantom stages: label phantoms built in {time.perf_counter() - t0:.1f} s "
        f"({len(np.unique(lab['total'])) - 1} total labels, {float((lab['total'] > 0).mean()):.2f} of the volume labelled)")
    out = {}

    def kernel_ms(fn):
        """(event-timed kernel ms, wall ms) of one call: the HIP events of the context's kernel-class timers bracket every launch of
        the call on its stream; the wall time of a sub-millisecond stage is mostly the host's launch + synchronise + table download"""
        ctx.sync()
        ctx.prof_reset()
        ctx.prof_enable(True)
        tb = time.perf_counter()
        r = fn()
        ctx.sync()
        wall = time.perf_counter() - tb
        ctx.prof_enable(False)
        ms = sum(v["ms"] for v in ctx.prof_get().values())
        if hasattr(r, "free"):
            r.free()
        return ms, wall * 1e3

    def timed(name, fn, algo_bytes, what):
        runs = [kernel_ms(fn) for _ in range(reps + 1)][1:]
        ms = float(np.median([r[0] for r in runs]))
        out[name] = {"ms": ms, "wall_ms": float(np.median([r[1] for r in runs])), "algorithmic_bytes": algo_bytes,
                     "achieved_GBps": algo_bytes / ms / 1e6, "frac": algo_bytes / ms / 1e6 / HBM_PEAK_GBPS, "bytes": what}

    def tissue():
        tis, _, _ = bca.tissue_aggregate(ctx, d_ct, d["regions"], d["parts"], zyx)
        return tis

    timed("tissue_aggregate", tissue, 5.0 * n, "5 B per voxel: CT 2 + regions 1 + parts 1 read, tissues 1 written (+ the per-slice tables)")
    timed("label_hu_histogram", lambda: M.label_hu_histogram(ctx, d_ct, d["total"], n) is None, 3.0 * n,
          "3 B per voxel: CT 2 + labels 1 read (117 labels x full int16 range histogram in LDS hash tables)")
    d_o, d_t = ctx.alloc(n), ctx.alloc(n)
    d_m = ctx.alloc(n)
    M.label_hu_mask(ctx, d_ct, d["total"], range(1, 30), 0, n, d_m)
    timed("binary_erode_6", lambda: M.binary_erode(ctx, d_m, d_o, d_t, zyx, 6), 2.0 * n,
          "2 B per voxel: mask read once, eroded mask written once (CNR masks: 6^3 footprint; the three separable passes run on bit masks)")

    def regions():
        dd = ctx.from_numpy(lab["regions"])     # (in place: a fresh copy per repetition; the upload is outside the clock)
        r = kernel_ms(lambda: bca.postprocess_region_segmentation_device(ctx, dd, zyx))
        dd.free()
        return r

    rs = [regions() for _ in range(reps + 1)][1:]
    ms = float(np.median([r[0] for r in rs]))
    out["region_cc_filters"] = {"ms": ms, "wall_ms": float(np.median([r[1] for r in rs])), "algorithmic_bytes": 4.0 * n,
                                "achieved_GBps": 4.0 * n / ms / 1e6, "frac": 4.0 * n / ms / 1e6 / HBM_PEAK_GBPS,
                                "bytes": "4 B per voxel: the four largest-component filters each read the label volume once (26-connected "
                                         "labelling on bit masks, csrc/ccl_bits.hip)"}
    timed("part_fill_and_cc_filters", lambda: bca.postprocess_part_segmentation_device(ctx, d["parts"], zyx, labels=range(1, 7)), 2.0 * n,
          "2 B per voxel: label volume read once, cleaned volume written once (6 labels: slice-wise contour fill + small-object + "
          "small-hole filters on bit masks)")
    for b in list(d.values()) + [d_ct, d_o, d_t, d_m]:
        b.free()
    log("phantom stages: " + ", ".join(f"{k} {v['ms']:.2f} ms ({v['frac']:.3f} of 8 TB/s)" for k, v in out.items()))
    return {"labels": "structured phantoms of boa_hip/synthetic.py at %dx%dx%d: 117 nested organ ellipsoids, 6 body parts, 11 nested body regions" % (X, Y, Z),
            "stages": out, "reps": reps,
            "note": "outside the timed region; the timed region runs the same stages on the synthetic nets' noise-like argmax (its `morphology` / "
                    "`aggregation` kernel classes)"}


# ------------------------------------------------------------------------------------------------- main
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a number for GPUs that are not running",
              file=sys.stderr)
        sys.exit(2)
    import torch
    dist = None
    if world > 1:
        from boa_hip import distributed as D
        if args.backend == "nccl" and torch.cuda.device_count() < world:
            print(f"bench.py: {world} RCCL ranks need {world} GPUs, {torch.cuda.device_count()} visible", file=sys.stderr)
            sys.exit(2)
        if args.backend == "gloo":
            local_rank %= max(torch.cuda.device_count(), 1)
        dist = D.init(args.backend, rank, world, local_rank)  # backend "nccl" is RCCL on ROCm
    log = (lambda *a: print(*a, file=sys.stderr, flush=True)) if rank == 0 else (lambda *a: None)

    from boa_hip import label_maps, plans, synthetic
    from boa_hip import measurements as M
    from boa_hip import sliding_window as sw
    from boa_hip.devarray import DevArray
    from boa_hip.device import Context
    from boa_hip.lanes import TotalBcaRunner
    from boa_hip.pipeline import BcaPipelineHip
    from boa_hip.task import SegmentationTask

    with_bca = args.models == "total+bca"
    shape = args.size * 3 if len(args.size) == 1 else args.size
    ctx = Context(local_rank)
    log("device:", ctx.info())
    t0 = time.perf_counter()
    part_models = synthetic.total_part_models()                         # [(task id, ModelConfig, blob, (plans, dataset, state dict))]
    total_task = SegmentationTask(ctx, "total", [(tid, cfg, [blob]) for tid, cfg, blob, _ in part_models], resample=1.5,
                                  multimodel=True, max_batch=args.batch)
    tasks = [total_task]
    pipe = None
    if with_bca:
        bm = {}
        for name, nc, seed in (("body_parts", 7, 543), ("body_regions", 12, 542)):
            pj, dj = plans.synthetic_plans(num_classes=nc, spacing=(5.0, 1.5, 1.5))
            cfg = plans.model_config_from_plans(pj, dj)
            bm[name] = (cfg, [plans.weight_blob_from_state_dict(cfg.geometry, plans.synthetic_state_dict(cfg.geometry, seed + f))
                              for f in range(5)])
        pipe = BcaPipelineHip(ctx, bm["body_parts"], bm["body_regions"], fast_bca=False, max_batch=args.batch)
        tasks += list(pipe.tasks.values())
    comm = None
    if dist is not None and args.shard != "volumes":
        from boa_hip import tile_shard as ts
        if args.backend == "nccl":     # RCCL through the C ABI: collectives on the engine's communication stream (boa_hip/rccl.py)
            from boa_hip.rccl import RcclComm
            comm = RcclComm(ctx, rank, world)
        else:                          # gloo: host-staged slabs (validation of the protocol with several ranks on one GPU)
            comm = ts.ShardComm(dist, rank, world, "cpu")
        for t in tasks:
            if args.shard == "tiles":
                t.shard = ts.TileShard(comm, args.shard_mode)
            else:
                t.model_shard = comm
        if pipe is not None:      # the aggregation half of a shared volume runs per z-slab (boa_hip/agg_shard.py)
            from boa_hip import agg_shard as ag
            pipe.agg = (ag.AggComm(dist, rank, world), comm)
    shared = comm is not None                                            # all ranks work on the same volume
    ct = synthetic.ct_phantom(shape, seed=20260928 + (0 if shared else rank))   # file array (x, y, z), int16 HU
    affine = np.diag([-1.5, -1.5, 1.5, 1.0])                             # an LPS file @1.5 mm: exercises the canonicalisation
    label_map = label_maps.measurement_label_map("total")
    log(f"setup (synthetic weights + phantom {shape}) {time.perf_counter() - t0:.1f}s")
    nvox = int(np.prod(shape))

    # work per volume (for the roofline / CPU extrapolation): tile forwards and FLOPs of every network
    tile_forwards, flops_per_volume, class_voxels = 0, 0.0, 0.0
    for tid, cfg, _, _ in part_models:
        PV, _ = sw.pad_amounts(shape, cfg.geometry.patch_size)
        nt = len(sw.get_sliding_window_origins(PV, cfg.geometry.patch_size, 0.8))
        tile_forwards += nt
        flops_per_volume += nt * cfg.geometry.flops_per_tile()
        class_voxels += (cfg.geometry.num_classes + 1.0) * nvox
    if with_bca:
        zb = int(round(shape[2] * 1.5 / 5.0))                            # slices at 5 mm (TS/resampling.py:165-181)
        for name in ("body_parts", "body_regions"):
            cfg = bm[name][0]
            vs = [zb, shape[1], shape[0]]                                # nnU-Net array order (z, y, x)
            PV, _ = sw.pad_amounts(vs, cfg.geometry.patch_size)
            nt = len(sw.get_sliding_window_origins(PV, cfg.geometry.patch_size, 0.5)) * 5
            tile_forwards += nt
            flops_per_volume += nt * cfg.geometry.flops_per_tile()
            class_voxels += 5 * (cfg.geometry.num_classes + 1.0) * float(np.prod(vs))

    d_ct = DevArray.from_numpy(ctx, ct)                                  # resident int16 CT, file axis order
    runner = TotalBcaRunner(total_task, pipe, label_map, cnr_adjustment=True)

    stage_log = {}

    def stage(name, t0):
        if os.environ.get("BOA_BENCH_STAGES"):
            ctx.sync()
            stage_log[name] = stage_log.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    def step(d_in, download=False, run=None):
        """One CT through total -> total measurements -> bca (boa_hip/lanes.py).  `download`: also bring the label volumes to
        the host.  `run`: the TotalBcaRunner (default: the one-stream runner of the timed region)."""
        clock = [time.perf_counter()]

        def on_stage(name):
            clock[0] = stage(name, clock[0])

        out = (run or runner).run_resident(d_in, affine, (1.5, 1.5, 1.5), on_stage)
        outs = [out[k] for k in ("total", "body_parts", "body_regions", "tissues") if k in out]
        host = [a.download() for a in outs] if download else None
        for a in outs:
            a.free()
        return out["total_measurements"], out.get("bca_measurements"), host

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier(device_ids=[local_rank]) if args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize() if torch.cuda.is_available() else None

    for _ in range(args.warmup):
        step(d_ct)
    ctx.sync()
    if os.environ.get("BOA_BENCH_CPROFILE") and rank == 0:   # host-side hot spots of one step (development aid)
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        step(d_ct)
        ctx.sync()
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
    ctx.counters(reset=True)
    ctx.prof_reset()
    ctx.prof_enable(not args.no_prof)
    barrier()
    t_start = time.perf_counter()
    step_s = []
    for _ in range(args.steps):
        t_s = time.perf_counter()
        meas, bca_js, _ = step(d_ct)     # (a step ends with the download of its last table: it has completed on the device)
        step_s.append(time.perf_counter() - t_s)
    barrier()
    elapsed = time.perf_counter() - t_start
    ctx.prof_enable(False)
    prof = ctx.prof_get()
    counters = ctx.counters()
    ranks_seen = world
    if dist is not None:
        dev = f"cuda:{local_rank}" if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one)                                             # ranks that actually took part, counted over RCCL
        ranks_seen = int(one.item())
        if ranks_seen != args.gpus:
            print(f"bench.py: {ranks_seen} ranks answered the all-reduce, --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)

    # the same loop without the per-launch HIP events (~33 k event records per step): what the events cost the headline
    no_events = None
    if rank == 0 and args.gpus == 1 and not args.no_prof:
        ctx.sync()
        tb = time.perf_counter()
        n_ne = max(5, args.steps)
        for _ in range(n_ne):
            step(d_ct)
        ctx.sync()
        no_events = {"value": n_ne / (time.perf_counter() - tb), "unit": "volumes/s", "steps": n_ne,
                     "note": "same one-stream loop with event profiling off (the headline keeps the events: `roofline` is measured over the timed region)"}
    # what the matrix cores of this very GPU sustain with no memory traffic at all (outside the timed region, ~50 ms): the part
    # is power-limited under matrix load and the clock it holds depends on the operand bits
    attainable = None
    if rank == 0:
        try:
            attainable = {"pure_mfma_constant_operands_TFLOPs": ctx.mfma_peak(False), "pure_mfma_random_operands_TFLOPs": ctx.mfma_peak(True),
                          "note": "v_mfma_f32_32x32x16_f16 only, one wave per SIMD, 4 accumulator chains, no LDS / memory traffic; "
                                  "synthetic weights and a noise phantom are random-operand data"}
        except Exception as e:  # diagnostics only
            attainable = {"error": f"{type(e).__name__}: {e}"}
    total_only = None
    if rank == 0 and args.models == "total+bca" and args.gpus == 1 and not args.no_h2h:
        # the `total` half alone (configs[1] as round 1 reported it), outside the timed region: 2 volumes
        ctx.sync()
        tb = time.perf_counter()
        for _ in range(2):
            total_task.predict_image(d_ct, affine, return_device=True).free()
        ctx.sync()
        total_only = {"volumes_per_s": 2.0 / (time.perf_counter() - tb), "steps": 2, "tile_forwards_per_volume": 625}
    h2h = None
    if rank == 0 and not args.no_h2h and args.gpus == 1:
        # PCIe-inclusive: upload the CT, download `total` + the three BCA label volumes (tables are host dicts already);
        # median of 5 volumes
        def host_to_host():
            ctx.sync()
            ts_h, chk = [], None
            ct_host = ctx.pinned_empty(ct.shape, ct.dtype)      # the caller's buffer, page-locked (Context.pinned_empty); the label
            ct_host[...] = ct                                    # volumes come back in page-locked arrays too (DeviceBuffer.download)
            for _ in range(5):
                tb = time.perf_counter()
                d_up = DevArray.from_numpy(ctx, ct_host)
                _, _, host = step(d_up, download=True)
                d_up.free()
                ctx.sync()
                ts_h.append(time.perf_counter() - tb)
                chk = int(np.sum(host[0], dtype=np.int64))       # (outside the clock: a checksum of the `total` labels that arrived)
                del host
            out = {"s_per_volume": float(np.median(ts_h)), "s_per_volume_all": ts_h, "steps": len(ts_h), "label_checksum": chk,
                   "host_buffers": "page-locked (boa_host_alloc): 268 MB CT up, 4 x 134 MB label volumes down per volume"}
            out["value"] = 1.0 / out["s_per_volume"]
            return out

        try:
            h2h = host_to_host()
        except Exception as e:  # noqa: BLE001  (an extra must never cost the headline line)
            h2h = {"error": f"{type(e).__name__}: {e}"}

    configs2 = None
    if rank == 0 and with_bca and args.gpus == 1 and not args.no_c3 and list(shape) == [512, 512, 512]:
        # configs[2] of BASELINE.json (the largest single-GPU configuration): one 512x512x768 whole-body volume, `total+bca`, same
        # predictors, same one-stream runner; 1 warm-up + 3 timed volumes, CT resident
        try:
            shape3 = [512, 512, 768]
            d3 = DevArray.from_numpy(ctx, synthetic.ct_phantom(shape3, seed=20260930))
            step(d3)
            ctx.sync()
            t3 = []
            for _ in range(3):
                tb = time.perf_counter()
                m3, b3, _ = step(d3)
                ctx.sync()
                t3.append(time.perf_counter() - tb)
            d3.free()
            configs2 = {"workload": "configs[2]: 512x512x768 @1.5 mm, total+bca, 1 x MI355X", "value": 1.0 / float(np.median(t3)), "unit": "volumes/s",
                        "value_mean": 1.0 / float(np.mean(t3)),
                        "s_per_volume": t3, "steps": len(t3),
                        "total_labels_present": int(sum(1 for v in m3["segmentations"]["total"].values() if v.get("present"))) if m3 else None,
                        "bca_aggregated_groups": len(b3.get("aggregated", {})) if b3 else None}
        except Exception as e:  # noqa: BLE001  (an extra must never cost the headline line)
            configs2 = {"error": f"{type(e).__name__}: {e}"}
        log(f"configs[2]: {configs2}")
    two_lanes = None
    if rank == 0 and with_bca and args.gpus == 1 and not args.no_lanes:
        # the product's two-lane mode (boa_hip/lanes.py): `total` + its measurements on this context's stream, both BCA nets +
        # post-processing + tissues on a second context (own stream, pool, predictors) of the same GPU; same kernels, same
        # results (tests/test_gpu_lanes.py).  Timed outside the headline region because the per-class HIP-event times of two
        # overlapping streams would no longer describe one kernel each: event profiling is off here.
        ctx2 = ctx3 = pipe2 = None
        try:
            check_trim = ctx.lib.boa_trim(ctx.h)                 # parked transient blocks back to the driver: the second context needs room
            log(f"two lanes: free device memory before the second context {ctx.info()['free_mem'] / 2 ** 30:.1f} GiB (trim rc {check_trim})")
            ctx2 = Context(local_rank)
            ctx3 = Context(local_rank) if args.lanes >= 3 else None
            pipe2 = BcaPipelineHip(ctx2, bm["body_parts"], bm["body_regions"], fast_bca=False, max_batch=args.batch, parts_ctx=ctx3)
            run2 = TotalBcaRunner(total_task, pipe2, label_map, cnr_adjustment=True)
            step(d_ct, run=run2)
            ctx.sync()
            ctx2.sync()
            tl = []
            tb = time.perf_counter()
            for _ in range(args.steps):
                t_s = time.perf_counter()
                step(d_ct, run=run2)
                tl.append(time.perf_counter() - t_s)
            ctx.sync()
            ctx2.sync()
            el2 = time.perf_counter() - tb
            two_lanes = {"value": args.steps / el2, "unit": "volumes/s", "ms_per_step": el2 * 1e3 / args.steps, "steps": args.steps,
                         "median_ms_per_step": float(np.median(tl)) * 1e3, "vs_one_stream": (args.steps / el2) / (args.steps / elapsed),
                         "note": "TotalBcaRunner with the BCA half on a second context/stream of the same GPU; event profiling off; "
                                 "labels and tables identical to the one-stream run"}
            two_lanes["lanes"] = args.lanes
        except Exception as e:  # noqa: BLE001  (an extra must never cost the headline line)
            two_lanes = {"error": f"{type(e).__name__}: {e}"}
        finally:
            for obj in (pipe2, ctx2, ctx3):
                try:
                    if obj is not None:
                        obj.close()
                except Exception:  # noqa: BLE001
                    pass
        log(f"two lanes: {two_lanes}")
    phantom = None
    if rank == 0 and args.gpus == 1 and not args.no_phantom:
        try:
            phantom = phantom_stages(ctx, shape, ct, log)
        except Exception as e:  # noqa: BLE001  (an extra must never cost the headline line)
            phantom = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        conv = prof["conv_mfma"]
        conv_ms = conv["ms"]
        achieved = conv["flops"] / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        total_ms = sum(v["ms"] for v in prof.values())
        for k, v in prof.items():
            if v["launches"]:
                log(f"  {k:16s} {v['ms']:10.1f} ms  {v['launches']:8d} launches  "
                    f"{v['flops'] / max(v['ms'], 1e-9) / 1e9:9.1f} TFLOP/s  {v['bytes'] / max(v['ms'], 1e-9) / 1e6:9.1f} GB/s")
        if stage_log:
            log("  stage wall times (s, synchronised, all steps): " + ", ".join(f"{k} {v:.3f}" for k, v in stage_log.items()))
        log(f"  sum of event-timed kernel classes {total_ms:.1f} ms of {elapsed * 1e3:.1f} ms wall; kernel variants {counters}")
        n_vol = (1 if shared else args.gpus) * args.steps
        res = {
            # (BASELINE.json's metric at the default --size 512; other sizes are development runs and say so)
            "metric": "CT volumes/sec (%s @1.5 mm, %s) on MI355X" % ("512^3" if list(shape) == [512, 512, 512] else "x".join(str(v) for v in shape),
                                                                     "total+bca" if with_bca else "total"),
            "value": n_vol / elapsed, "unit": "volumes/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "strong" if shared else "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"configs[1] volume ({shape[0]}x{shape[1]}x{shape[2]} @1.5 mm, LPS file), `{args.models}`: 5 part models "
                                   f"(patch 128^3, step 0.8)" + (" + total measurements + body_parts / body_regions nets (5 folds each at 5 mm "
                                   "slices, step 0.5) + CC / contour-fill post-processing + tissues + tables" if with_bca else
                                   " + total measurements") + f", {tile_forwards} tile forwards per volume; " +
                                   (f"each volume shared by {args.gpus} GPUs (--shard {args.shard}" +
                                    (f", {args.shard_mode} slab exchange)" if args.shard == "tiles" else ")") if shared
                                    else "1 volume per GPU"),
                       "tile_forwards_per_volume": tile_forwards, "tile_batch": args.batch,
                       "tflop_per_volume": flops_per_volume / 1e12, "ranks_seen": ranks_seen, "backend": args.backend if world > 1 else None,
                       "precision": "fp16 activations/weights, fp32 MFMA accumulate, fp16 logit accumulators (reference semantics)",
                       "input": "int16 CT resident in HBM at the start of the timed region; label volumes stay in HBM, tables on the host"},
            "roofline": {"bound": "mfma", "kernel": "k_conv_ws<R,K> + k_conv_ns<S,WN,RM> (all MFMA 3x3x3 / 1x3x3 conv launches of the step: "
                                                    "wave-specialised implicit GEMM, v_mfma_f32_32x32x16_f16; k_conv_ns = the stride-2 layers)",
                         "achieved": achieved, "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_F16_DENSE_PEAK_TFLOPS, "traffic": None,
                         "launches": conv["launches"], "avg_launch_ms": conv_ms / max(conv["launches"], 1),
                         "flops_per_launch": conv["flops"] / max(conv["launches"], 1),
                         "bytes_per_launch": conv["bytes"] / max(conv["launches"], 1),
                         "share_of_kernel_time": conv_ms / max(total_ms, 1e-9),
                         "attainable": attainable,
                         "frac_of_attainable_random_operands": (achieved / attainable["pure_mfma_random_operands_TFLOPs"]
                                                                if attainable and "error" not in attainable else None)},
            "end_to_end_tflops": flops_per_volume * n_vol / elapsed / 1e12,
            # the HBM-bound stages of the same run (algorithmic bytes / event time against 8 TB/s): BASELINE.json's "% HBM roofline"
            "hbm_stages": {k: {"achieved_GBps": prof[k]["bytes"] / max(prof[k]["ms"], 1e-9) / 1e6,
                               "frac": prof[k]["bytes"] / max(prof[k]["ms"], 1e-9) / 1e6 / HBM_PEAK_GBPS,
                               # against what a plain float4 copy reaches on this part (MI355X_MICROARCH.md: 6.3 TB/s measured)
                               "frac_of_measured_copy": prof[k]["bytes"] / max(prof[k]["ms"], 1e-9) / 1e6 / HBM_COPY_GBPS,
                               "frac_of_measured_write_stream": (prof[k]["bytes"] / max(prof[k]["ms"], 1e-9) / 1e6 / HBM_WRITE_GBPS
                                                                 if k in ("convT_mfma", "conv_first") else None),
                               "ms": prof[k]["ms"], "launches": prof[k]["launches"]}
                           for k in ("head_accum", "finalize_argmax", "convT_mfma", "conv_first") if prof[k]["launches"]},
            # (head_accum: since round 3 the gather form of the tile loop -- stash read + labels, no accumulator planes, no separate
            #  finalize pass; its algorithmic bytes are 66 B per voxel and covering tile, a third of the scatter form's)
            # ONE number for BASELINE.json's "% HBM roofline": the time-weighted fraction over every event-timed class that is not the MFMA
            # conv class (sum of their algorithmic bytes / sum of their event times / 8 TB/s); the launch-latency classes (norm_finalize) and
            # the LDS-bound scan stages (morphology) are in it on purpose -- they are part of the step
            "hbm_roofline": (lambda ks: {
                "frac": sum(prof[k]["bytes"] for k in ks) / max(sum(prof[k]["ms"] for k in ks), 1e-9) / 1e6 / HBM_PEAK_GBPS,
                "achieved_GBps": sum(prof[k]["bytes"] for k in ks) / max(sum(prof[k]["ms"] for k in ks), 1e-9) / 1e6,
                "peak_GBps": HBM_PEAK_GBPS, "ms_per_step": sum(prof[k]["ms"] for k in ks) / args.steps,
                "share_of_kernel_time": sum(prof[k]["ms"] for k in ks) / max(total_ms, 1e-9),
                "classes": {k: {"ms_per_step": prof[k]["ms"] / args.steps,
                                "frac": prof[k]["bytes"] / max(prof[k]["ms"], 1e-9) / 1e6 / HBM_PEAK_GBPS} for k in ks},
                "note": "algorithmic bytes (what a stage must read and write once) over HIP-event time, all non-MFMA kernel classes of the timed region"})(
                    [k for k, v in prof.items() if k != "conv_mfma" and v["launches"] and v["bytes"] > 0]),
            "kernel_variants": counters,
            "median_ms_per_step": float(np.median(step_s)) * 1e3,
            # the PCIe-inclusive rate of the same workload, stated at top level next to `value` (host CT in, label volumes + tables out;
            # details in `host_to_host`); `value` itself starts with the CT resident in HBM as the bench contract prescribes
            "value_host_to_host": (h2h or {}).get("value"),
            "value_without_event_profiling": (no_events or {}).get("value"),
            "no_event_profiling": no_events,
            "total_only": total_only,
            "configs2": configs2,
            "two_lanes": two_lanes,
            "phantom_stages": phantom,
            "host_to_host": h2h,
            "tables": {"total_labels_present": int(sum(1 for v in meas["segmentations"]["total"].values() if v.get("present"))) if meas else None,
                       "bca_aggregated_groups": len(bca_js.get("aggregated", {})) if bca_js else None},
        }
        # HBM traffic of the dominant kernel: FETCH_SIZE / WRITE_SIZE from the committed rocprofv3 --pmc passes of THIS workload
        # (tools/profile_round.sh: separate passes at 512^3, gfx950 FETCH correction; counters cannot be collected inside this
        # process), launch-weighted mean over the k_conv_ws launches; per launch like `achieved`
        try:
            pj_ = json.load(open(PMC_PROFILE))
            rows = [v for k, v in pj_["kernels"].items() if "k_conv_ws" in k or "k_conv_ns" in k]
            nd = sum(v["dispatches"] for v in rows)
            res["roofline"]["traffic"] = 1e6 * 1.048576 * sum(
                v["dispatches"] * (v["fetch_MB_corrected_per_dispatch"] + v["write_MB_per_dispatch"]) for v in rows) / nd
            res["roofline"]["traffic_source"] = {"file": os.path.relpath(PMC_PROFILE, ROOT), "command": pj_.get("command"),
                                                 "git": pj_.get("git")}
        except Exception as e:  # noqa: BLE001
            res["roofline"]["traffic_source"] = f"unavailable ({type(e).__name__})"
        if comm is not None and hasattr(comm, "stats"):
            res["comm"] = dict(comm.stats(), note="rank 0's boa_comm_stats over the whole process (warm-up + timed steps): RCCL calls and bytes "
                                                  "sent or reduced; DESIGN.md section 6 tabulates the expected bytes per boundary")
        dev_tiles, close_parity = None, None
        if not args.no_parity and args.gpus == 1:
            try:
                res["parity"], dev_tiles, close_parity = parity_sample(ctx, part_models[0][1], part_models[0][2], min(args.batch, 8), log,
                                                                       tile_forwards, elapsed / args.steps)
            except Exception as e:  # noqa: BLE001  (an extra must never cost the headline line)
                res["parity"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu and args.gpus == 1:
            res["cpu_baseline"], vs_oracle = cpu_baseline(part_models[0][3], args.cpu_tiles,
                                                          {"tile_forwards": tile_forwards, "class_voxels": class_voxels, "voxels": float(nvox),
                                                           "with_bca": with_bca, "shape": list(shape), "all_cores": args.cpu_all_cores}, log,
                                                          device_tiles=dev_tiles)
            if vs_oracle is not None:
                res["parity"]["vs_oracle"] = vs_oracle
        else:
            res["cpu_baseline"] = None
        if close_parity is not None:
            close_parity()
        if not args.no_exact and args.gpus == 1:
            # The label-contract mode, MEASURED: one whole volume of the same workload with every network in the fp32 mode
            # (split precision on the matrix cores, csrc/net_x3.hip; labels of the CPU path: tests/test_gpu_production_geometry.py).
            # The production predictors are closed first (their weight sets and statistics tables are not needed any more; the
            # activation arena of the context is shared by all networks and simply grows to the fp32 mode's 2 GB per tile).
            try:
                for t in tasks:
                    t.close()
                tasks.clear()
                ctx.lib.boa_trim(ctx.h)
                te0 = time.perf_counter()
                x_total = SegmentationTask(ctx, "total", [(tid, cfg, [blob]) for tid, cfg, blob, _ in part_models], resample=1.5,
                                           multimodel=True, max_batch=args.exact_batch, precision="fp32")
                x_pipe = BcaPipelineHip(ctx, bm["body_parts"], bm["body_regions"], fast_bca=False, max_batch=args.exact_batch,
                                        precision="fp32") if with_bca else None
                x_run = TotalBcaRunner(x_total, x_pipe, label_map, cnr_adjustment=True)
                step(d_ct, run=x_run)                           # warm-up: weight split / packing, first touch of the fp32 activations
                ctx.sync()
                t_setup = time.perf_counter() - te0
                ctx.counters(reset=True)
                ctx.prof_reset()
                ctx.prof_enable(True)
                tx = []
                for _ in range(2):
                    tb = time.perf_counter()
                    x_meas, _, _ = step(d_ct, run=x_run)
                    ctx.sync()
                    tx.append(time.perf_counter() - tb)
                ctx.prof_enable(False)
                xprof = ctx.prof_get()
                xc = xprof["conv_mfma"]
                xcnt = ctx.counters()
                res.setdefault("parity", {})
                res["parity"]["exact_mode"] = {
                    "value": 1.0 / float(np.mean(tx)), "unit": "volumes/s", "s_per_volume": tx, "steps": len(tx), "measured": True,
                    "tile_batch": args.exact_batch, "tile_forwards_per_volume": tile_forwards,
                    "ms_per_tile_forward": float(np.mean(tx)) * 1e3 / tile_forwards,
                    "conv_TFLOPs_fp32_equivalent": xc["flops"] / max(xc["ms"], 1e-9) / 1e9,
                    # MFMA flops per fp32 flop: 41 / 13.5 = 3.04 on the tap-paired row-reuse layers (all stride-1 layers 128^3 .. 16^3, round 6),
                    # 4 on the stride-2 (k_conv_ns) and per-tap layers; the issue rate below assumes 3.04 everywhere (a lower bound)
                    "mfma_flops_per_fp32_flop": {"row_reuse_layers": 41.0 / 13.5, "stride2_and_per_tap_layers": 4.0},
                    "conv_mfma_issue_TFLOPs_lower_bound": (41.0 / 13.5) * xc["flops"] / max(xc["ms"], 1e-9) / 1e9,
                    "conv_frac_of_f16_peak_lower_bound": (41.0 / 13.5) * xc["flops"] / max(xc["ms"], 1e-9) / 1e9 / MFMA_F16_DENSE_PEAK_TFLOPS,
                    "kernel_ms_per_volume": {k: v["ms"] / len(tx) for k, v in xprof.items() if v["launches"]},
                    "kernel_variants": xcnt, "setup_and_warmup_s": t_setup,
                    "total_labels_present": int(sum(1 for v in x_meas["segmentations"]["total"].values() if v.get("present"))) if x_meas else None,
                    "note": "the same total+bca volume with every network in the fp32 mode: fp32 weights / activations / accumulation as the "
                            "reference's CPU path (predict_from_raw_data.py:648), every operand split into two fp16 parts on the matrix cores "
                            "(3 MFMAs per 8 input channels and TWO taps in the row-reuse convs: Wh Xh + Wh Xl + Wl Xh with the k-halves holding "
                            "the same part of two taps; 2 MFMAs per tap elsewhere); whole volumes timed, nothing extrapolated"}
                log(f"exact (fp32 split-precision) mode: {tx} s per volume -> {1.0 / float(np.mean(tx)):.4f} volumes/s; conv "
                    f"{xc['flops'] / max(xc['ms'], 1e-9) / 1e9:.0f} TFLOP/s fp32-equivalent; variants {xcnt}")
                x_total.close()
                if x_pipe is not None:
                    x_pipe.close()
            except Exception as e:  # noqa: BLE001  (an extra must never cost the headline line)
                res.setdefault("parity", {})
                res["parity"]["exact_mode"] = {"error": f"{type(e).__name__}: {e}"}
        if args.dump:
            os.makedirs(os.path.dirname(os.path.abspath(args.dump)), exist_ok=True)
            with open(args.dump, "w") as f:
                json.dump({"prof": prof, "elapsed_s": elapsed, "result": res}, f, indent=1)
        print(json.dumps(res), flush=True)
    d_ct.free()
    for t in tasks:
        t.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
