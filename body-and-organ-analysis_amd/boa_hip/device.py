"""Device context and buffers on top of the C ABI (plumbing only: memory, streams, timers)."""
from __future__ import annotations

import atexit
import ctypes as C
import threading
import weakref

import numpy as np

from . import _lib
from ._lib import check


_LIVE = weakref.WeakSet()
_SHUTDOWN = [False]            # set at interpreter exit: page-locked blocks are no longer handed back to a runtime that may be gone


@atexit.register
def _close_all():
    # (atexit runs handlers in reverse registration order: this one is registered first and therefore runs LAST -- after
    #  _pinned_atexit below has given the cached page-locked blocks back while the HIP runtime is still up)
    for c in list(_LIVE):
        try:
            c.close()
        except Exception:
            pass


PINNED_MIN_BYTES = 1 << 20     # downloads of at least this many bytes land in page-locked memory (PCIe at link speed)


# page-locking costs more than the copy it speeds up: released blocks are kept for reuse, up to $BOA_PINNED_CACHE_GB per process
# (default 2: four 512^3 label volumes in flight; page-locked memory cannot be swapped, and a node runs one process per GPU)
PINNED_CACHE_BYTES = int(float(__import__("os").environ.get("BOA_PINNED_CACHE_GB", "2")) * (1 << 30))
_PINNED_FREE: dict = {}        # rounded size -> [host pointers]
_PINNED_CACHED = [0]
_PINNED_LOCK = threading.Lock()


def _pinned_round(nbytes: int) -> int:
    return (int(nbytes) + (2 << 20) - 1) & ~((2 << 20) - 1)


class _PinnedBlock:
    """Page-locked host memory (boa_host_alloc) exposed through the array interface; goes back to the process-wide cache (or
    to the system beyond PINNED_CACHE_BYTES) with the last array viewing it."""

    def __init__(self, lib, ptr: int, nbytes: int, size: int):
        self._lib, self._ptr, self._size = lib, ptr, size
        self.__array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def __del__(self):
        try:
            if _SHUTDOWN[0]:      # interpreter teardown: the HIP runtime may already be gone (the OS reclaims the pages)
                return
            with _PINNED_LOCK:
                if _PINNED_CACHED[0] + self._size <= PINNED_CACHE_BYTES:
                    _PINNED_FREE.setdefault(self._size, []).append(self._ptr)
                    _PINNED_CACHED[0] += self._size
                    return
            self._lib.boa_host_free(None, C.c_void_p(self._ptr))
        except Exception:
            pass


def _pinned_take(size: int):
    with _PINNED_LOCK:
        lst = _PINNED_FREE.get(size)
        if lst:
            _PINNED_CACHED[0] -= size
            return lst.pop()
    return None


def pinned_trim(lib=None):
    """Release every cached page-locked block."""
    lib = lib or _lib.lib()
    with _PINNED_LOCK:
        for lst in _PINNED_FREE.values():
            for ptr in lst:
                lib.boa_host_free(None, C.c_void_p(ptr))
        _PINNED_FREE.clear()
        _PINNED_CACHED[0] = 0


@atexit.register
def _pinned_atexit():
    try:
        pinned_trim()
    except Exception:
        pass
    _SHUTDOWN[0] = True


class DeviceBuffer:
    """A hipMalloc'ed buffer owned by a Context; freed explicitly or when garbage collected."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(ctx.lib.boa_malloc(ctx.h, self.nbytes, C.byref(p)), f"boa_malloc({nbytes})")
        self.ptr = p.value

    def free(self):
        if self.ptr is not None and self.ctx.h is not None:
            self.ctx.lib.boa_free(self.ctx.h, C.c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def zero(self):
        check(self.ctx.lib.boa_memset(self.ctx.h, C.c_void_p(self.ptr), 0, self.nbytes))

    def upload(self, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes, (a.nbytes, self.nbytes)
        check(self.ctx.lib.boa_h2d(self.ctx.h, C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p), a.nbytes))
        return self

    def download(self, shape, dtype) -> np.ndarray:
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        out = self.ctx.pinned_empty(shape, dtype) if nbytes >= PINNED_MIN_BYTES else np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes, (out.nbytes, self.nbytes)
        check(self.ctx.lib.boa_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), out.nbytes))
        return out

    @property
    def vp(self):
        return C.c_void_p(self.ptr)


class FlagRing:
    """Deferred "inf in the predicted array" flags (predict_from_raw_data.py:623-625 raises after every sliding window): each predict
    call of a multi-model task takes its own zeroed int32 slot and the host reads the slots ONCE, when the volume's last model has
    been queued -- a read per model was a stream drain per model (host gaps between the stages)."""
    SLOTS = 64

    def __init__(self, ctx):
        self.ctx = ctx
        self.buf = ctx.alloc(4 * self.SLOTS)
        self.buf.zero()
        self.n = 0

    def next(self):
        if self.n == self.SLOTS:
            self.check()
        v = BufferView(self.buf, 4 * self.n, 4)
        self.n += 1
        return v

    def reset(self):
        """Drop what an aborted volume left behind."""
        if self.n:
            self.buf.zero()
            self.n = 0

    def check(self):
        """Raises if any slot handed out since the last check was set; the slots are zero again afterwards."""
        if self.n == 0:
            return
        used, self.n = self.n, 0
        flags = self.buf.download((self.SLOTS,), np.int32)
        if flags[:used].any():
            self.buf.zero()
            raise RuntimeError("Encountered inf in predicted array. Aborting...")

    def free(self):
        if self.buf is not None:
            self.buf.free()
            self.buf = None


class BufferView:
    """Non-owning window [byte_offset, byte_offset + nbytes) of a DeviceBuffer (z-slabs of a (z, y, x) volume are
    contiguous ranges of its buffer)."""

    def __init__(self, buf, byte_offset: int, nbytes: int):
        assert 0 <= byte_offset and byte_offset + nbytes <= buf.nbytes, (byte_offset, nbytes, buf.nbytes)
        self._keep = buf
        self.ctx = buf.ctx
        self.ptr = buf.ptr + int(byte_offset)
        self.nbytes = int(nbytes)

    @property
    def vp(self):
        return C.c_void_p(self.ptr)

    def free(self):
        pass

    def zero(self):
        if self.nbytes:
            check(self.ctx.lib.boa_memset(self.ctx.h, self.vp, 0, self.nbytes))

    def download(self, shape, dtype) -> np.ndarray:
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        out = self.ctx.pinned_empty(shape, dtype) if nbytes >= PINNED_MIN_BYTES else np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes, (out.nbytes, self.nbytes)
        if out.nbytes:
            check(self.ctx.lib.boa_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.vp, out.nbytes))
        return out


class Context:
    """One per GPU / process (`boa_init`)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = _lib.lib()
        h = C.c_void_p()
        check(self.lib.boa_init(int(device), C.c_void_p(stream) if stream else None, C.byref(h)), "boa_init")
        self.h = h
        self.device = device
        self._children = weakref.WeakSet()  # objects holding device state (predictors) closed before the context
        _LIVE.add(self)

    def mfma_peak(self, random_operands: bool = True, iters: int = 40000) -> float:
        """TFLOP/s a pure v_mfma_f32_32x32x16_f16 loop sustains on this GPU (no memory traffic): the attainable ceiling of the
        matrix cores under power, near-constant or random operand bits (`boa_mfma_peak`)."""
        out = C.c_double()
        check(self.lib.boa_mfma_peak(self.h, 1 if random_operands else 0, int(iters), C.byref(out)), "boa_mfma_peak")
        return float(out.value)

    def register(self, obj):
        self._children.add(obj)

    def close(self):
        if self.h is not None:
            for ch in list(self._children):
                try:
                    ch.close()
                except Exception:
                    pass
            self.lib.boa_destroy(self.h)
            self.h = None

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def pinned_empty(self, shape, dtype) -> np.ndarray:
        """numpy array in page-locked host memory (what `torch.empty(..., pin_memory=True)` is to the reference): uploads from
        it and downloads into it run at PCIe link speed.  Falls back to ordinary memory if the allocation is refused."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape))
        if n == 0:
            return np.empty(shape, dtype=dtype)
        size = _pinned_round(n * dtype.itemsize)
        ptr = _pinned_take(size)
        if ptr is None:
            p = C.c_void_p()
            if self.lib.boa_host_alloc(self.h, size, C.byref(p)) != 0:
                return np.empty(shape, dtype=dtype)
            ptr = p.value
        block = _PinnedBlock(self.lib, ptr, n * dtype.itemsize, size)
        return np.asarray(block).view(dtype).reshape(shape)

    def from_numpy(self, arr: np.ndarray) -> DeviceBuffer:
        a = np.ascontiguousarray(arr)
        return DeviceBuffer(self, max(a.nbytes, 1)).upload(a)

    def zeros(self, nbytes: int) -> DeviceBuffer:
        b = DeviceBuffer(self, nbytes)
        b.zero()
        return b

    def sync(self):
        check(self.lib.boa_sync(self.h), "boa_sync")

    def bind_thread(self):
        """Called by a host thread other than the creating one before it drives this context (boa_bind_thread)."""
        check(self.lib.boa_bind_thread(self.h), "boa_bind_thread")

    def info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int()
        tot, free = C.c_size_t(), C.c_size_t()
        check(self.lib.boa_device_info(self.h, name, 256, C.byref(cu), C.byref(tot), C.byref(free)))
        return {"name": name.value.decode(), "cu_count": cu.value, "total_mem": tot.value, "free_mem": free.value}

    # ---- timing -----------------------------------------------------------------------------------
    def timer_start(self, slot=0):
        check(self.lib.boa_timer_start(self.h, slot))

    def timer_stop(self, slot=0) -> float:
        ms = C.c_float()
        check(self.lib.boa_timer_stop(self.h, slot, C.byref(ms)))
        return ms.value

    def prof_enable(self, on=True):
        check(self.lib.boa_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self):
        check(self.lib.boa_prof_reset(self.h))

    def counters(self, reset: bool = False) -> dict:
        """Launches per kernel variant since creation / the last reset (boa_debug_counter)."""
        return {name: int(self.lib.boa_debug_counter(self.h, k, 1 if reset else 0)) for k, name in enumerate(_lib.CNT_NAMES)}

    def prof_get(self):
        out = {}
        for k, name in enumerate(_lib.K_NAMES):
            ms, n, fl, by = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
            check(self.lib.boa_prof_get(self.h, k, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
            out[name] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
        return out
