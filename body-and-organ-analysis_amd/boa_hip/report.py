"""Report-side reductions that touch whole volumes, on resident data: only the pixels a figure needs leave
the device.

  create_equidistant_overview   BCA/report/plots/check.py:10-36 (+ overlay.py:7-27): HU window + colour overlay of five slices.
                                The five (y,x) slices of the CT and of every label volume are gathered on the device
                                (`boa_copy3` on strided views) and come back as 5 x Y x X arrays; the blend is the reference's
                                fp64 arithmetic on those 5 slices (pinned by golden G14).
  major_minor_axis / find_axes  BOA/compute/ts_metrics.py:33-66, geometry.py:49-85: slice range of vertebra L3 from the device
                                presence table, the middle slice of the body mask comes back (one slice), then 2-D geometry:
                                scipy ConvexHull + farthest hull points exactly as the reference calls them; the minor-axis end
                                points are cv2 rasterisation there (absent here): restated as a walk along the perpendicular
                                to the last foreground pixel -- unpinned, within ~2 pixels on convex bodies.
The tissue heat-map reductions are `bca.tissue_projections` (boa_tissue_projections).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .devarray import DevArray
from .device import Context

OVERVIEW_NAMES = ["First", "25%", "Central", "75%", "Last"]


def apply_hu_window(image: np.ndarray, hu_min: float = -150.0, hu_max: float = 400.0) -> np.ndarray:
    """overlay.py:20-23: HU -> [0, 255] grey ramp between hu_min and hu_max (same operation order, golden G14)."""
    ramp = np.subtract(image, hu_min)            # int HU -> fp64, float32 HU stays float32 (numpy promotion, as there)
    ramp /= (hu_max - hu_min)
    np.clip(ramp, 0.0, 1.0, out=ramp)
    ramp *= 255.0
    return ramp


def blend_overlay(gray_image: np.ndarray, color_rgb: np.ndarray, mask: np.ndarray, opacity: float) -> np.ndarray:
    """overlay.py:5-13: grey slice [Y][X] + per-pixel colour [Y][X][3]; pixels under `mask` are the opacity mix, the others
    the grey value on all three channels.  Only the masked pixels are mixed (the figure is mostly background)."""
    gray = np.asarray(gray_image)
    out = np.repeat(gray.astype(np.float64)[:, :, None], 3, axis=2)
    sel = np.asarray(mask, dtype=bool)
    if sel.any():
        keep = 1 - opacity
        # `gray * (1 - opacity)` is evaluated in the grey image's own dtype there (a float32 slice stays float32 for this product)
        # before the float64 colour term is added
        out[sel] = (gray[sel] * keep)[:, None] + np.asarray(color_rgb)[sel].astype(np.float64) * opacity
    return out


def overview_locations(num_slices: int) -> List[int]:
    return [0, int(num_slices * 0.25), int(num_slices * 0.5), int(num_slices * 0.75), num_slices - 1]


def _slices(vol, idx: Sequence[int]) -> np.ndarray:
    """The (y,x) slices `idx` of a (z,y,x) volume: host array or resident DevArray (one contiguous slice copy each)."""
    if isinstance(vol, DevArray):
        out = []
        for i in idx:
            s = vol.slice(0, i, i + 1).contiguous(force_copy=True)
            try:
                out.append(s.download()[0])
            finally:
                s.free()
        return np.stack(out)
    return np.stack([np.asarray(vol)[i] for i in idx])


def create_equidistant_overview(image_zyx, segmentations: Sequence[Tuple[object, Sequence[Sequence[int]]]],
                                opacity: float = 0.25) -> List[list]:
    """image_zyx / label volumes: (z,y,x) host arrays or DevArrays.  -> [[name, composed RGB slice (float64 [Y][X][3]) per
    segmentation ...] for the five overview slices], as the reference returns it."""
    idx = overview_locations(int(image_zyx.shape[0]))
    img = _slices(image_zyx, idx)
    segs = [(_slices(s, idx), np.asarray(c)) for s, c in segmentations]
    result = []
    for k, name in enumerate(OVERVIEW_NAMES):
        gray = apply_hu_window(img[k])
        row: list = [name]
        for sl, cmap in segs:
            row.append(blend_overlay(gray, cmap[sl[k]], sl[k] > 0, opacity))
        result.append(row)
    return result


def find_axes(middle_slice: np.ndarray):
    """geometry.py:49-85 on one boolean (y,x) slice -> (major_p1, major_p2, minor_p1, minor_p2) as (x, y) tuples."""
    from scipy import spatial
    points = np.flip(np.transpose(np.where(middle_slice)))
    hull_points = points[spatial.ConvexHull(points).vertices]
    hdist = spatial.distance.cdist(hull_points, hull_points, metric="euclidean")
    i1, i2 = np.unravel_index(hdist.argmax(), hdist.shape)
    p1, p2 = tuple(int(v) for v in hull_points[i1]), tuple(int(v) for v in hull_points[i2])
    mid = ((p1[0] + p2[0]) // 2, (p1[1] + p2[1]) // 2)
    nx, ny = p1[0] - p2[0], p1[1] - p2[1]
    fac = math.sqrt(nx * nx + ny * ny)
    nx, ny = nx / fac, ny / fac
    H, W = middle_slice.shape

    def walk(dx: float, dy: float):
        last = mid
        for t in range(1, H + W):
            x, y = int(round(mid[0] + dx * t)), int(round(mid[1] + dy * t))
            if not (0 <= x < W and 0 <= y < H):
                break
            if middle_slice[y, x]:
                last = (x, y)
        return last

    return p1, p2, walk(-ny, nx), walk(ny, -nx)


def major_minor_axis(ctx: Context, total_zyx, body_parts_zyx, l3_label: int, img_spacing_xy) -> Tuple[Optional[float], Optional[float]]:
    """ts_metrics.major_minor_axis on resident (z,y,x) label volumes: major / minor body axis (mm) on the middle L3 slice."""
    from . import bca
    dev = isinstance(total_zyx, DevArray)
    if dev:
        t = total_zyx.contiguous()
        present = bca.slice_label_presence(ctx, t.buf, t.shape)[:, int(l3_label)]
        if t.buf is not total_zyx.buf:
            t.free()
    else:
        present = (np.asarray(total_zyx) == l3_label).any(axis=(1, 2))
    slices = np.where(present)[0]
    if len(slices) == 0:
        return None, None
    mid = int(np.median(slices))
    middle = _slices(body_parts_zyx, [mid])[0] == 1
    if not middle.any():
        return None, None
    from scipy import spatial
    a1, a2, b1, b2 = find_axes(middle)
    sp = float(np.mean(img_spacing_xy))
    return spatial.distance.euclidean(a1, a2) * sp, spatial.distance.euclidean(b1, b2) * sp
