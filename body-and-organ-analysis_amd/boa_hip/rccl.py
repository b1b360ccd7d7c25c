"""RCCL transport of the shared-volume modes on top of the C ABI (csrc/comm.hip): the collectives are queued by libboa_hip.so on
a communication stream of the engine and ordered against the compute stream by events -- no torch tensors, no host
synchronisation in the data path (torch.distributed, where present, only carries the 128-byte RCCL id to the other ranks and the
bench's timing barrier).  Same duck type as tile_shard.ShardComm for the code that only needs rank / world / all_reduce."""
from __future__ import annotations

import ctypes as C
import os
import socket
import struct
import time
from typing import Callable, Optional

from ._lib import check


def tcp_broadcast(payload: Optional[bytes], rank: int, world: int, addr: Optional[str] = None, port: Optional[int] = None,
                  timeout: float = 120.0) -> bytes:
    """Rank 0's `payload` to every rank over a throw-away TCP rendezvous (MASTER_ADDR : $BOA_RDZV_PORT, default MASTER_PORT + 1):
    the out-of-band channel for the RCCL id when no torch.distributed process group exists."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port or os.environ.get("BOA_RDZV_PORT") or (int(os.environ.get("MASTER_PORT", "29511")) + 1))
    if world == 1:
        return payload
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        srv.settimeout(timeout)
        conns = [srv.accept()[0] for _ in range(world - 1)]
        for c in conns:
            c.sendall(struct.pack("<I", len(payload)) + payload)
            c.close()
        srv.close()
        return payload
    t0 = time.time()
    while True:
        try:
            c = socket.create_connection((addr, port), timeout=5.0)
            break
        except OSError:
            if time.time() - t0 > timeout:
                raise
            time.sleep(0.05)
    c.settimeout(timeout)

    def read(n):
        buf = b""
        while len(buf) < n:
            part = c.recv(n - len(buf))
            if not part:
                raise ConnectionError("rendezvous closed early")
            buf += part
        return buf

    n = struct.unpack("<I", read(4))[0]
    out = read(n)
    c.close()
    return out


class RcclComm:
    """One RCCL communicator over all ranks, owned by the C library.  `bcast(bytes or None) -> bytes`: rank 0's bytes on every
    rank (default: torch.distributed's object broadcast when a process group exists, else tcp_broadcast)."""

    on_device = True

    def __init__(self, ctx, rank: int, world: int, bcast: Optional[Callable[[Optional[bytes]], bytes]] = None):
        self.ctx, self.lib, self.rank, self.world = ctx, ctx.lib, int(rank), int(world)
        self.device = f"cuda:{ctx.device}"
        if not self.lib.boa_comm_available():
            raise RuntimeError("librccl.so not found (set BOA_RCCL_LIB)")
        ident = (C.c_ubyte * 128)()
        if self.rank == 0:
            check(self.lib.boa_comm_unique_id(ident), "boa_comm_unique_id")
        raw = (bcast or self._default_bcast)(bytes(ident) if self.rank == 0 else None)
        ident = (C.c_ubyte * 128).from_buffer_copy(raw)
        h = C.c_void_p()
        check(self.lib.boa_comm_create(ctx.h, self.world, self.rank, ident, C.byref(h)), "boa_comm_create")
        self.h = h
        ctx.register(self)

    def _default_bcast(self, payload):
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                box = [payload]
                dist.broadcast_object_list(box, src=0)
                return box[0]
        except ImportError:
            pass
        return tcp_broadcast(payload, self.rank, self.world)

    # ---- collectives (queued on the communication stream; `wait` orders later compute work behind them) -----------------------
    def wait(self):
        check(self.lib.boa_comm_wait(self.h), "boa_comm_wait")

    def all_reduce(self, buf, count: int, dtype: int, wait: bool = True):
        """In-place sum of `count` elements of the device buffer `buf` over all ranks; dtype 0 uint8, 1 fp16, 2 int32, 3 fp32."""
        check(self.lib.boa_comm_all_reduce(self.h, buf.vp, int(count), int(dtype)), "boa_comm_all_reduce")
        if wait:
            self.wait()

    def shift_slab(self, dst, upper, src, lower, acc, nacc, n_classes: int, PV, stage=None):
        from ._lib import int3
        s_lo, s_hi = upper if dst is not None else (0, 0)
        r_lo, r_hi = lower if src is not None else (0, 0)
        check(self.lib.boa_comm_shift_slab(self.h, -1 if dst is None else int(dst), int(s_lo), int(s_hi), -1 if src is None else int(src),
                                           int(r_lo), int(r_hi), acc.vp, nacc.vp, int(n_classes), int3(PV), stage.vp if stage is not None else None),
              "boa_comm_shift_slab")

    def planes_to_owner(self, buf, n_classes: int, PV, sends, recvs, wait: bool = True):
        """Plane ranges of the fp16 logits `buf` [C][PV0][PV1][PV2] to / from several peers, in place: sends / recvs = lists of
        (peer, lo, hi).  See tile_shard.reduce_scatter_logit_planes."""
        from ._lib import int3
        import numpy as np

        def cols(items):
            a = np.asarray(items, dtype=np.int32).reshape(-1, 3)
            return [np.ascontiguousarray(a[:, k]) for k in range(3)]

        sp, sl, sh = cols(sends)
        rp, rl, rh = cols(recvs)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))   # noqa: E731
        check(self.lib.boa_comm_planes_to_owner(self.h, buf.vp, int(n_classes), int3(PV), len(sp), ip(sp), ip(sl), ip(sh), len(rp), ip(rp), ip(rl),
                                                ip(rh)), "boa_comm_planes_to_owner")
        if wait:
            self.wait()

    def stats(self):
        calls, nbytes = C.c_longlong(), C.c_longlong()
        check(self.lib.boa_comm_stats(self.h, C.byref(calls), C.byref(nbytes)), "boa_comm_stats")
        return {"calls": int(calls.value), "bytes_sent_or_reduced": int(nbytes.value)}

    def make_room(self, ctx):
        """RCCL allocates its channel buffers from the same HBM as the engine's caching allocator, whose parked blocks it cannot
        reclaim: release them when HBM is short."""
        if ctx.info()["free_mem"] < (16 << 30):
            ctx.lib.boa_trim(ctx.h)

    def close(self):
        if getattr(self, "h", None) is not None and self.ctx.h is not None:
            self.lib.boa_comm_destroy(self.h)
        self.h = None
