"""Oracle: TotalSegmentator resampling (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
from __future__ import annotations

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


def spline_zoom_explicit(data: np.ndarray, out_shape, order: int = 3) -> np.ndarray:
    """Operation-level restatement of scipy.ndimage.zoom(mode="nearest", grid_mode=False) used as the
    blueprint for the device kernel: separable cubic B-spline prefilter (pole z1 = sqrt(3) - 2, 'nearest'
    boundary initialisation as in scipy's ni_splines) then gather with coordinate
    in = out * (n_in - 1) / (n_out - 1), clamped ("nearest") boundary.  Checked against ndimage.zoom in
    tests (<= 1e-9 abs on fp64)."""
    if order == 0:
        idx = []
        for n_in, n_out in zip(data.shape, out_shape):
            scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
            c = np.arange(n_out) * scale
            idx.append(np.clip(np.floor(c + 0.5).astype(np.int64), 0, n_in - 1))
        return data[np.ix_(*idx)]
    assert order == 3
    coeff = ndimage.spline_filter(np.asarray(data, dtype=np.float64), order=3, mode="nearest")
    out = coeff
    for ax, (n_in, n_out) in enumerate(zip(data.shape, out_shape)):
        scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
        c = np.arange(n_out) * scale
        base = np.floor(c).astype(np.int64)
        t = c - base
        w = np.stack([
            (1 - t) ** 3 / 6,
            (3 * t ** 3 - 6 * t ** 2 + 4) / 6,
            (-3 * t ** 3 + 3 * t ** 2 + 3 * t + 1) / 6,
            t ** 3 / 6,
        ], 0)
        acc = 0
        for k in range(4):
            ii = np.clip(base - 1 + k, 0, n_in - 1)
            sl = np.take(out, ii, axis=ax)
            shp = [1] * out.ndim
            shp[ax] = n_out
            acc = acc + sl * w[k].reshape(shp)
        out = acc
    return out
