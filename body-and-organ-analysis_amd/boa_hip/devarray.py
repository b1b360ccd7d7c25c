"""Strided 3-D views of device buffers (host-side metadata only) + `boa_copy3` to materialise / scatter them.

The index remaps around a task -- reorientation (TS/alignment.py), crops (TS/cropping.py), the (x,y,z) <-> (z,y,x) view
change between nibabel and nnU-Net arrays, z-splits and their recombination (TS/nnunet.py:495-505,583-586) -- are view
operations here (transpose / flip / slice change offset and strides, no data moves); data moves once per pipeline stage
through `boa_copy3`, on the device.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from ._lib import check
from .device import Context, DeviceBuffer

_CODES = {np.dtype(np.uint8): 0, np.dtype(np.int16): 1, np.dtype(np.int32): 2, np.dtype(np.float32): 3,
          np.dtype(np.float64): 4}


def _ll3(v):
    return (C.c_longlong * 3)(*[int(x) for x in v])


def _i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


class DevArray:
    """shape / strides / offset in ELEMENTS of `dtype` on top of a DeviceBuffer (shared between views)."""

    def __init__(self, ctx: Context, buf: DeviceBuffer, shape: Sequence[int], dtype, strides: Optional[Sequence[int]] = None,
                 offset: int = 0):
        self.ctx, self.buf = ctx, buf
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        if self.dtype not in _CODES:
            raise TypeError(f"DevArray: unsupported dtype {self.dtype}")
        if strides is None:
            strides = (self.shape[1] * self.shape[2], self.shape[2], 1)
        self.strides = tuple(int(s) for s in strides)
        self.offset = int(offset)

    # ---- construction --------------------------------------------------------------------------------------
    @classmethod
    def from_numpy(cls, ctx: Context, arr: np.ndarray) -> "DevArray":
        arr = np.ascontiguousarray(arr)
        if arr.ndim != 3:
            raise ValueError("DevArray: 3-D arrays only")
        return cls(ctx, ctx.from_numpy(arr), arr.shape, arr.dtype)

    @classmethod
    def empty(cls, ctx: Context, shape, dtype) -> "DevArray":
        n = int(np.prod(shape))
        return cls(ctx, ctx.alloc(max(n, 1) * np.dtype(dtype).itemsize), shape, dtype)

    @classmethod
    def zeros(cls, ctx: Context, shape, dtype) -> "DevArray":
        a = cls.empty(ctx, shape, dtype)
        a.buf.zero()
        return a

    # ---- views ---------------------------------------------------------------------------------------------
    def _view(self, shape, strides, offset) -> "DevArray":
        return DevArray(self.ctx, self.buf, shape, self.dtype, strides, offset)

    def transpose(self, axes) -> "DevArray":
        axes = [int(a) for a in axes]
        return self._view([self.shape[a] for a in axes], [self.strides[a] for a in axes], self.offset)

    def flip(self, axis: int) -> "DevArray":
        st = list(self.strides)
        off = self.offset + (self.shape[axis] - 1) * st[axis]
        st[axis] = -st[axis]
        return self._view(self.shape, st, off)

    def slice(self, axis: int, start: int, stop: int) -> "DevArray":
        start, stop, _ = slice(start, stop).indices(self.shape[axis])
        sh = list(self.shape)
        sh[axis] = max(stop - start, 0)
        return self._view(sh, self.strides, self.offset + start * self.strides[axis])

    def box(self, bbox) -> "DevArray":
        v = self
        for ax, (a, b) in enumerate(bbox):
            v = v.slice(ax, a, b)
        return v

    def apply_orientation(self, ornt) -> "DevArray":
        """nibabel.orientations.apply_orientation as a view (flips, then transpose)."""
        ornt = np.asarray(ornt)
        v = self
        for ax, fl in enumerate(ornt[:, 1]):
            if fl == -1:
                v = v.flip(ax)
        return v.transpose(np.argsort(ornt[:, 0]))

    @property
    def is_contiguous(self) -> bool:
        return self.offset == 0 and self.strides == (self.shape[1] * self.shape[2], self.shape[2], 1)

    @property
    def size(self) -> int:
        return int(np.prod(self.shape))

    # ---- data movement -------------------------------------------------------------------------------------
    def copy_to(self, dst: "DevArray"):
        """dst[...] = self (same shape; dtype conversion like numpy astype) on the device."""
        if dst.shape != self.shape:
            raise ValueError(f"copy_to: shape {self.shape} -> {dst.shape}")
        check(self.ctx.lib.boa_copy3(self.ctx.h, self.buf.vp, _CODES[self.dtype], self.offset, _ll3(self.strides), _i3(self.shape),
                                     dst.buf.vp, _CODES[dst.dtype], dst.offset, _ll3(dst.strides)), "boa_copy3")

    def contiguous(self, dtype=None, force_copy: bool = False) -> "DevArray":
        dtype = self.dtype if dtype is None else np.dtype(dtype)
        if self.is_contiguous and dtype == self.dtype and not force_copy:
            return self
        out = DevArray.empty(self.ctx, self.shape, dtype)
        self.copy_to(out)
        return out

    def to_context(self, ctx: Context) -> "DevArray":
        """A contiguous copy owned by another Context of the same GPU (its pool, its stream): the copy is issued on the
        DESTINATION's stream, so the source must be complete (its own context synchronised) before this is called."""
        out = DevArray.empty(ctx, self.shape, self.dtype)
        check(ctx.lib.boa_copy3(ctx.h, self.buf.vp, _CODES[self.dtype], self.offset, _ll3(self.strides), _i3(self.shape),
                                out.buf.vp, _CODES[out.dtype], out.offset, _ll3(out.strides)), "boa_copy3")
        return out

    def download(self) -> np.ndarray:
        a = self.contiguous()
        return a.buf.download(a.shape, a.dtype)

    def nonzero_bbox(self):
        """[[lo, hi], ...] of data != 0 (contiguous int16 / int32 / float32 arrays)."""
        if not self.is_contiguous:
            raise ValueError("nonzero_bbox: contiguous array required")
        bb = (C.c_int * 6)()
        check(self.ctx.lib.boa_nonzero_bbox(self.ctx.h, self.buf.vp, _CODES[self.dtype], _i3(self.shape), bb), "boa_nonzero_bbox")
        return [[bb[0], bb[1]], [bb[2], bb[3]], [bb[4], bb[5]]]

    def free(self):
        self.buf.free()
