"""Oracle: nnU-Net sliding-window prediction arithmetic (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates, in numpy with explicit fp16/fp32 rounding steps:
  NN/inference/sliding_window_prediction.py:10-27   compute_gaussian
  NN/inference/sliding_window_prediction.py:30-54   compute_steps_for_sliding_window
  NN/inference/predict_from_raw_data.py:506-538     _internal_get_sliding_window_slicers
  NN/inference/predict_from_raw_data.py:560-631     _internal_predict_sliding_window_return_logits
  NN/inference/predict_from_raw_data.py:634-680     predict_sliding_window_return_logits (+ pad_nd_image)
  NN/inference/predict_from_raw_data.py:471-504     predict_logits_from_preprocessed_data (fold mean)
"""
from __future__ import annotations

import numpy as np
from scipy.ndimage import gaussian_filter


def compute_steps_for_sliding_window(image_size, tile_size, tile_step_size):
    """NN/inference/sliding_window_prediction.py:30-54."""
    assert 0 < tile_step_size <= 1
    target = [i * tile_step_size for i in tile_size]
    num_steps = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, tile_size)]
    steps = []
    for dim in range(len(tile_size)):
        max_step_value = image_size[dim] - tile_size[dim]
        if num_steps[dim] > 1:
            actual = max_step_value / (num_steps[dim] - 1)
        else:
            actual = 99999999999
        steps.append([int(np.round(actual * i)) for i in range(num_steps[dim])])
    return steps


def compute_gaussian(tile_size, sigma_scale=1.0 / 8, value_scaling_factor=10.0):
    """NN/inference/sliding_window_prediction.py:10-27 -> np.float16 array.

    fp64 Gaussian (scipy, mode=constant) -> scaled so max == value_scaling_factor -> fp16 ->
    zeros replaced by the smallest non-zero entry.  torch converts double->half through float;
    the same two-step conversion is used here.
    """
    tmp = np.zeros(tile_size)
    center = tuple(i // 2 for i in tile_size)
    sigmas = [i * sigma_scale for i in tile_size]
    tmp[center] = 1
    g = gaussian_filter(tmp, sigmas, 0, mode="constant", cval=0)
    g = g / (np.max(g) / value_scaling_factor)
    g = g.astype(np.float32).astype(np.float16)
    mask = g == 0
    g[mask] = np.min(g[~mask])
    return g


def pad_nd_image(image, new_shape):
    """acvl_utils 0.2.5 pad_nd_image (constant 0), call site NN/inference/predict_from_raw_data.py:657.

    Pads the trailing len(new_shape) axes symmetrically to at least new_shape
    (below = d // 2, above = d // 2 + d % 2).  Returns (padded, slicer_to_revert).
    """
    old = np.array(image.shape)
    n = len(new_shape)
    tgt = list(old[:-n]) + list(new_shape)
    tgt = np.array([max(a, b) for a, b in zip(tgt, old)])
    diff = tgt - old
    below = diff // 2
    above = diff // 2 + diff % 2
    pads = [(int(b), int(a)) for b, a in zip(below, above)]
    res = np.pad(image, pads, "constant", constant_values=0) if diff.any() else image
    slicer = tuple(slice(int(b), int(res.shape[i] - a)) for i, (b, a) in enumerate(pads))
    return res, slicer


def get_sliding_window_slicers(image_size, patch_size, tile_step_size):
    """NN/inference/predict_from_raw_data.py:506-538 (3-D branch): x outer, y, z inner."""
    steps = compute_steps_for_sliding_window(image_size, patch_size, tile_step_size)
    out = []
    for sx in steps[0]:
        for sy in steps[1]:
            for sz in steps[2]:
                out.append((sx, sy, sz))
    return out


def accumulate_tile(acc, n, pred_f32, gauss_f16, start):
    """One iteration of the hot loop, NN/inference/predict_from_raw_data.py:611-614.

    prediction *= gaussian          (fp32 * fp16 -> fp32)
    predicted_logits[sl] += pred    (fp16 += fp32: computed in fp32, rounded RTNE to fp16)
    n_predictions[sl[1:]] += gauss  (fp16 += fp16: computed in fp32, rounded to fp16)
    """
    px, py, pz = pred_f32.shape[1:]
    sx, sy, sz = start
    sl = (slice(None), slice(sx, sx + px), slice(sy, sy + py), slice(sz, sz + pz))
    if gauss_f16 is not None:
        pred = pred_f32.astype(np.float32) * gauss_f16.astype(np.float32)[None]
    else:
        pred = pred_f32.astype(np.float32)
    acc[sl] = (acc[sl].astype(np.float32) + pred).astype(np.float16)
    if gauss_f16 is not None:
        n[sl[1:]] = (n[sl[1:]].astype(np.float32) + gauss_f16.astype(np.float32)).astype(np.float16)
    else:
        n[sl[1:]] = (n[sl[1:]].astype(np.float32) + np.float32(1)).astype(np.float16)


def finalize_logits(acc, n):
    """torch.div(predicted_logits, n_predictions, out=predicted_logits) + inf check,
    NN/inference/predict_from_raw_data.py:620-625.  fp16 / fp16 evaluated in fp32, rounded to fp16."""
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        out = (acc.astype(np.float32) / n.astype(np.float32)[None]).astype(np.float16)
    if np.any(np.isinf(out)):
        raise RuntimeError("Encountered inf in predicted array. Aborting...")
    return out


def predict_sliding_window_return_logits(network_fn, input_image, patch_size, num_heads,
                                         tile_step_size=0.5, use_gaussian=True,
                                         return_aux=False):
    """NN/inference/predict_from_raw_data.py:634-680 on CPU (no autocast).

    network_fn: float32 ndarray [1, C_in, *patch] -> float32 ndarray [1, heads, *patch].
    input_image: float32 [C_in, X, Y, Z].  Returns fp16 [heads, X, Y, Z].
    """
    assert input_image.ndim == 4
    data, revert = pad_nd_image(input_image, patch_size)
    slicers = get_sliding_window_slicers(data.shape[1:], patch_size, tile_step_size)
    acc = np.zeros((num_heads, *data.shape[1:]), dtype=np.float16)
    n = np.zeros(data.shape[1:], dtype=np.float16)
    g = compute_gaussian(tuple(patch_size), 1.0 / 8, 10.0) if use_gaussian else None
    for (sx, sy, sz) in slicers:
        patch = np.ascontiguousarray(
            data[:, sx:sx + patch_size[0], sy:sy + patch_size[1], sz:sz + patch_size[2]])[None]
        pred = np.asarray(network_fn(patch))[0]
        accumulate_tile(acc, n, pred, g, (sx, sy, sz))
    n_before = n.copy()
    out = finalize_logits(acc, n)
    out = out[(slice(None), *revert[1:])]
    if return_aux:
        return out, n_before[revert[1:]], slicers
    return out


def ensemble_folds(fold_logits):
    """NN/inference/predict_from_raw_data.py:483-500: prediction += next (fp16 + fp16),
    prediction /= n_folds (fp16 / python int -> fp16), both evaluated in fp32 and rounded."""
    pred = fold_logits[0].copy()
    for other in fold_logits[1:]:
        pred = (pred.astype(np.float32) + other.astype(np.float32)).astype(np.float16)
    if len(fold_logits) > 1:
        pred = (pred.astype(np.float32) / np.float32(len(fold_logits))).astype(np.float16)
    return pred
