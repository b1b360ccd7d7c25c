"""Host mirror of the array part of TS/resampling.py (`resample_img` :24-56, `change_spacing` :129-222) on the device
resampler (csrc/resample.hip).  The NIfTI/affine bookkeeping of `change_spacing` stays with the caller; `new_affine`
restates :190-194 for callers that carry an affine."""
from __future__ import annotations

import numpy as np

from ._lib import check, int3
from .device import Context, DeviceBuffer

_IN_DTYPES = {np.dtype(np.int16): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3}


def zoomed_shape(shape, zoom):
    """scipy.ndimage.zoom's output shape: round(n * zoom) per axis (Python round: half to even)."""
    zoom = np.broadcast_to(np.asarray(zoom, dtype=np.float64), (len(shape),))
    return tuple(int(round(n * z)) for n, z in zip(shape, zoom))


def resample_cubic_device(ctx: Context, dev_in: DeviceBuffer, in_dtype, in_shape, out_shape, out_dtype=np.int32):
    """order-3 resample of a device-resident volume; returns a DeviceBuffer of `out_dtype` (int32 or float64)."""
    code = _IN_DTYPES[np.dtype(in_dtype)]
    oc = {np.dtype(np.int32): 0, np.dtype(np.float64): 1}[np.dtype(out_dtype)]
    n = int(np.prod(out_shape))
    out = DeviceBuffer(ctx, n * np.dtype(out_dtype).itemsize)
    check(ctx.lib.boa_resample_cubic(ctx.h, dev_in.vp, code, int3(in_shape), out.vp, oc, int3(out_shape)),
          "boa_resample_cubic")
    return out


def resample_nearest_device(ctx: Context, dev_in: DeviceBuffer, in_shape, out_shape):
    out = DeviceBuffer(ctx, int(np.prod(out_shape)))
    check(ctx.lib.boa_resample_nearest_u8(ctx.h, dev_in.vp, int3(in_shape), out.vp, int3(out_shape)),
          "boa_resample_nearest_u8")
    return out


def resample_img(ctx: Context, img: np.ndarray, zoom=0.5, order: int = 0, out_dtype=None, target_shape=None):
    """`resample_img(img, zoom, order)` for a 3-D array.  order 3 -> float64 (or int32 with the reference's `.astype`
    truncation when out_dtype=np.int32); order 0 -> uint8 labels.  Other orders are not on the reference's path."""
    img = np.asarray(img)
    if img.ndim != 3:
        raise ValueError(f"resample_img: 3-D arrays only, got {img.shape}")
    shape = tuple(int(s) for s in target_shape) if target_shape is not None else zoomed_shape(img.shape, zoom)
    if order == 0:
        if img.dtype != np.uint8:
            if img.min() < 0 or img.max() > 255:
                raise ValueError("resample_img(order=0): label values must fit uint8")
            img = img.astype(np.uint8)
        d = DeviceBuffer(ctx, img.size).upload(img)
        o = resample_nearest_device(ctx, d, img.shape, shape)
        res = o.download(shape, np.uint8)
        d.free(), o.free()
        return res
    if order != 3:
        raise ValueError(f"resample_img: order {order} not supported (the reference uses 0 and 3)")
    if img.dtype not in _IN_DTYPES:
        img = img.astype(np.float64)
    od = np.dtype(np.float64 if out_dtype is None else out_dtype)
    d = DeviceBuffer(ctx, img.nbytes).upload(img)
    o = resample_cubic_device(ctx, d, img.dtype, img.shape, shape, od)
    res = o.download(shape, od)
    d.free(), o.free()
    return res


def change_spacing_array(ctx: Context, data: np.ndarray, img_spacing, new_spacing=1.25, target_shape=None, order=0,
                         dtype=None, remove_negative=False):
    """Array part of `change_spacing` (:165-217): returns (new_data, zoom), or (data, None) when the spacing already
    matches (:179-181).  `img_spacing` as header.get_zooms() gives it (float32)."""
    old_shape = np.array(data.shape)
    img_spacing = np.array(img_spacing, dtype=np.float32)
    if target_shape is not None:
        zoom = np.array(target_shape) / old_shape
        new_spacing = img_spacing / zoom
    else:
        if type(new_spacing) is float:
            new_spacing = [new_spacing] * 3
        new_spacing = np.array(new_spacing)
        zoom = img_spacing / new_spacing
    if np.array_equal(img_spacing, new_spacing):
        return data, None
    want_i32 = dtype is not None and np.dtype(dtype) == np.int32 and not remove_negative and order == 3
    new = resample_img(ctx, data, zoom, order, out_dtype=np.int32 if want_i32 else None)
    if remove_negative:
        new[new < 1e-4] = 0
    if dtype is not None:
        new = new.astype(dtype, copy=False)
    return new, zoom


def new_affine(affine: np.ndarray, zoom) -> np.ndarray:
    """:190-194: scale each column vector by the zoom of its dimension."""
    a = np.copy(affine)
    for i in range(3):
        a[:3, i] = a[:3, i] / zoom[i]
    return a
