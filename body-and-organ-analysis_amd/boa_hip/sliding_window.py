"""Host-side geometry of the nnU-Net sliding window: tile starts, Gaussian importance map, padding.

Mirrors (same names, arguments and results)
  NN/inference/sliding_window_prediction.py:10-27   compute_gaussian
  NN/inference/sliding_window_prediction.py:30-54   compute_steps_for_sliding_window
  NN/inference/predict_from_raw_data.py:506-538     _internal_get_sliding_window_slicers (tile order x -> y -> z)
These are O(patch) integer / fp64 computations done once per model; the per-voxel work is on the device.
"""
from __future__ import annotations

from functools import lru_cache

import numpy as np


def compute_steps_for_sliding_window(image_size, tile_size, tile_step_size):
    assert all(i >= j for i, j in zip(image_size, tile_size)), "image size must be as large or larger than patch_size"
    assert 0 < tile_step_size <= 1, "step_size must be larger than 0 and smaller or equal to 1"
    target_step_sizes_in_voxels = [i * tile_step_size for i in tile_size]
    num_steps = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target_step_sizes_in_voxels, tile_size)]
    steps = []
    for dim in range(len(tile_size)):
        max_step_value = image_size[dim] - tile_size[dim]
        actual_step_size = max_step_value / (num_steps[dim] - 1) if num_steps[dim] > 1 else 99999999999
        steps.append([int(np.round(actual_step_size * i)) for i in range(num_steps[dim])])
    return steps


def _gaussian_kernel1d(sigma: float, radius: int) -> np.ndarray:
    """scipy.ndimage._filters._gaussian_kernel1d(order=0): exp(-x^2 / (2 sigma^2)) normalised to sum 1."""
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


@lru_cache(maxsize=4)
def _gaussian_cached(tile_size, sigma_scale, value_scaling_factor):
    # gaussian_filter(delta at the centre, sigma_i = tile_i * sigma_scale, truncate=4, mode="constant") is the
    # outer product of the three truncated 1-D kernels, multiplied axis 0 first (each pass adds only exact zeros).
    prof = []
    for n in tile_size:
        sigma = n * sigma_scale
        radius = int(4.0 * sigma + 0.5)
        w = _gaussian_kernel1d(sigma, radius)
        c = n // 2
        p = np.zeros(n, dtype=np.float64)
        for i in range(n):
            d = i - c
            if -radius <= d <= radius:
                p[i] = w[radius + d]
        prof.append(p)
    g = prof[0][:, None, None] * prof[1][None, :, None]
    g = g * prof[2][None, None, :]
    g = g / (np.max(g) / value_scaling_factor)
    g16 = g.astype(np.float32).astype(np.float16)  # torch: double -> half goes through float
    mask = g16 == 0
    if mask.any():
        g16[mask] = np.min(g16[~mask])
    g16.setflags(write=False)
    return g16


def compute_gaussian(tile_size, sigma_scale: float = 1.0 / 8, value_scaling_factor: float = 1.0) -> np.ndarray:
    """fp16 importance map [tile_size]; zeros are replaced by the smallest non-zero value."""
    return _gaussian_cached(tuple(int(t) for t in tile_size), float(sigma_scale), float(value_scaling_factor))


def pad_amounts(shape, patch_size):
    """acvl_utils.pad_nd_image (call site predict_from_raw_data.py:657): symmetric zero padding up to the patch
    size; returns (padded_shape, below) with below = d // 2."""
    padded, below = [], []
    for s, p in zip(shape, patch_size):
        d = max(p - s, 0)
        padded.append(s + d)
        below.append(d // 2)
    return padded, below


def get_sliding_window_origins(image_size, patch_size, tile_step_size):
    """Tile origins in canonical order (first axis outermost), int32 [n_tiles, 3]."""
    steps = compute_steps_for_sliding_window(image_size, patch_size, tile_step_size)
    out = [(sx, sy, sz) for sx in steps[0] for sy in steps[1] for sz in steps[2]]
    return np.asarray(out, dtype=np.int32).reshape(-1, 3)
