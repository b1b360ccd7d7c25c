"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU).

This module holds the process-group set-up and the VOLUME granularity (SURVEY 8e granularity 1): CT volumes are
independent in the reference (one CT per process, TS/python_api.py:54-72), so ranks take disjoint volumes and no data-path
collective exists; the only communication is the timing barrier / max-reduce of bench.py and the gather of small result
tables.  The two granularities that share ONE volume between the ranks live in `tile_shard.py`: the part models dealt out
to the ranks (label volumes all-reduced), and the tile rows of the sliding window split across the ranks with the overlap
slabs of the fp16 accumulators exchanged over RCCL (`TileShard`, exact hand-over or pairwise sum); the z-slab sharding of the
aggregation stages is in `agg_shard.py`.  The data-path collectives of the shared-volume modes are issued by the C library itself
(`rccl.RcclComm` -> csrc/comm.hip: RCCL on the engine's communication stream); torch.distributed carries the control plane only
-- the bench's barrier and max-over-ranks, the 128-byte RCCL id, pickled host tables -- and the gloo transport
(`tile_shard.ShardComm`) that lets several ranks share one GPU for validation.
"""
from __future__ import annotations

import os
from typing import List, Sequence


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_volumes(n_volumes: int, rank: int, world: int) -> List[int]:
    """Volumes of this rank: contiguous blocks, sizes differ by at most one (all ranks agree without talking)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(n_volumes, world)
    lo = rank * q + min(rank, r)
    return list(range(lo, lo + q + (1 if rank < r else 0)))


def init(backend: str, rank: int, world: int, local_rank: int = 0):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl":
        # device tensors over RCCL, host tensors / pickled objects (the small tables of agg_shard.AggComm, the RCCL id of
        # rccl.RcclComm) over gloo: one process group that dispatches on the tensor's device
        torch.cuda.set_device(local_rank)
        dist.init_process_group("cpu:gloo,cuda:nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def reduce_max(dist, value: float, device="cpu") -> float:
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(dist, obj, world: int) -> Sequence:
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out
