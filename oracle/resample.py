"""Oracle: TotalSegmentator resampling (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
from __future__ import annotations

import math

import numpy as np
from scipy import ndimage


def resample_img(img: np.ndarray, zoom, order: int) -> np.ndarray:
    """TS/resampling.py:24-56 for a 3-D array: scipy.ndimage.zoom(img, zoom, mode="nearest", order=order)."""
    return ndimage.zoom(img, zoom, mode="nearest", order=order)


def change_spacing_array(data: np.ndarray, img_spacing, new_spacing=None, target_shape=None, order=0,
                         dtype=None):
    """Array part of TS/resampling.py:129-222 (`change_spacing`): returns (new_data, zoom) or (data, None)
    when spacing already matches (:179-181).  `data` is what nibabel's get_fdata() yields: float64."""
    data = np.asarray(data, dtype=np.float64)
    old_shape = np.array(data.shape)
    img_spacing = np.array(img_spacing, dtype=np.float32)  # header.get_zooms() are float32
    if target_shape is not None:
        zoom = np.array(target_shape) / old_shape
        new_spacing = img_spacing / zoom
    else:
        if isinstance(new_spacing, float):
            new_spacing = [new_spacing] * 3
        new_spacing = np.array(new_spacing)
        zoom = img_spacing / new_spacing
    if np.array_equal(img_spacing, new_spacing):
        return data if dtype is None else data, None
    new = resample_img(data, zoom, order)
    if dtype is not None:
        new = new.astype(dtype)
    return new, zoom


SPLINE3_POLE = float.fromhex("-0x1.126145e9ecd56p-2")


def _filter_axis_reflect(a: np.ndarray, axis: int) -> np.ndarray:
    """Cubic B-spline prefilter along one axis: pole sqrt(3)-2, gain (1-z)(1-1/z), 'reflect' boundary initialisation
    (what scipy's C code uses for mode="nearest"), causal + anticausal recursions.  The pole is the constant the C
    compiler folds `sqrt(3.0) - 2.0` to (correctly rounded, 2 ulp away from the run-time double expression); with it
    this function reproduces scipy.ndimage.spline_filter1d bit for bit (scipy 1.15.3, tests)."""
    z = SPLINE3_POLE
    a = np.moveaxis(a, axis, 0).copy()
    n = a.shape[0]
    a *= (1.0 - z) * (1.0 - 1.0 / z)
    z_n = math.pow(z, n)
    z_i = z
    c0 = a[0].copy()
    a[0] = a[0] + z_n * a[n - 1]
    for i in range(1, n):
        a[0] += z_i * (a[i] + z_n * a[n - 1 - i])
        z_i *= z
    a[0] *= z / (1 - z_n * z_n)
    a[0] += c0
    for i in range(1, n):
        a[i] += z * a[i - 1]
    a[n - 1] *= z / (z - 1.0)
    for i in range(n - 2, -1, -1):
        a[i] = z * (a[i + 1] - a[i])
    return np.moveaxis(a, 0, axis)


def spline_zoom_explicit(data: np.ndarray, out_shape, order: int = 3) -> np.ndarray:
    """Operation-level restatement of scipy.ndimage.zoom(mode="nearest", grid_mode=False): the blueprint of the device
    kernels (csrc/resample.hip).  order 3: edge-pad by 12 (scipy._prepad_for_spline_filter), separable prefilter,
    4x4x4-tap interpolation with in = out * (n_in-1)/(n_out-1) (unclamped, see below), terms accumulated with the
    first axis outermost.  Bit-identical to ndimage.zoom (scipy 1.15.3) on every case of the tests, including the
    `.astype(int32)` truncation of G5.  order 0: index floor(in + 0.5), clamped."""
    if order == 0:
        idx = []
        for n_in, n_out in zip(data.shape, out_shape):
            scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 1.0
            c = np.arange(n_out) * scale
            idx.append(np.clip(np.floor(c + 0.5).astype(np.int64), 0, n_in - 1))
        return data[np.ix_(*idx)]
    assert order == 3
    npad = 12
    c = np.pad(np.asarray(data, dtype=np.float64), npad, mode="edge")
    for ax in range(3):
        c = _filter_axis_reflect(c, ax)
    idxs, ws = [], []
    for n_in, n_out in zip(data.shape, out_shape):
        zf = (n_in - 1) / (n_out - 1) if n_out > 1 else 1.0
        cc = np.arange(n_out) * zf + npad    # (not clipped: an overshoot of one ulp past n_in - 1 is evaluated in the padding)
        fl = np.floor(cc)
        x = cc - fl
        y, zz = x, 1.0 - x
        w1 = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0
        w2 = (zz * zz * (zz - 2.0) * 3.0 + 4.0) / 6.0
        w0 = zz * zz * zz / 6.0
        w3 = 1.0 - w0 - w1 - w2
        idxs.append(fl.astype(int) - 1)
        ws.append(np.stack([w0, w1, w2, w3]))
    out = np.zeros(out_shape)
    for a in range(4):
        for b in range(4):
            for d in range(4):
                cf = c[np.ix_(idxs[0] + a, idxs[1] + b, idxs[2] + d)]
                out += cf * ws[0][a][:, None, None] * ws[1][b][None, :, None] * ws[2][d][None, None, :]
    return out
