"""Axis orientation bookkeeping: index remaps only, host side.

TS/alignment.py:8-46 (`as_closest_canonical`, `undo_canonical`) and BCA/io.py:97-113
(`load_nibabel_image_with_axcodes`) delegate to nibabel.orientations (nibabel 5.3, un-vendored, absent here -> PARITY
UNPINNED vs nibabel; the functions below restate its published algorithms `io_orientation`, `axcodes2ornt`,
`ornt_transform`, `apply_orientation`, `inv_ornt_aff`, `aff2axcodes` and are tested for self-consistency and on
hand-checked affines).  An orientation is an (3, 2) array: row i = (output axis of input axis i, flip +1/-1).
"""
from __future__ import annotations

import numpy as np

_LABELS = (("L", "R"), ("P", "A"), ("I", "S"))
RAS_ORNT = np.array([[0, 1], [1, 1], [2, 1]], dtype=np.float64)


def io_orientation(affine: np.ndarray, tol=None) -> np.ndarray:
    """Orientation of input axes in terms of output (world RAS) axes, closest axis-aligned match."""
    affine = np.asarray(affine, dtype=np.float64)
    q, p = affine.shape[0] - 1, affine.shape[1] - 1
    rzs = affine[:q, :p]
    zooms = np.sqrt(np.sum(rzs * rzs, axis=0))
    zooms[zooms == 0] = 1
    rs = rzs / zooms
    P, S, Qs = np.linalg.svd(rs, full_matrices=False)
    if tol is None:
        tol = S.max() * max(rs.shape) * np.finfo(S.dtype).eps
    keep = S > tol
    R = np.dot(P[:, keep], Qs[keep])
    ornt = np.ones((p, 2), dtype=np.int8) * np.nan
    for in_ax in range(p):
        col = R[:, in_ax]
        if not np.allclose(col, 0):
            out_ax = int(np.argmax(np.abs(col)))
            ornt[in_ax, 0] = out_ax
            ornt[in_ax, 1] = -1 if col[out_ax] < 0 else 1
            R[out_ax, :] = 0  # an output axis is used once
    return ornt


def axcodes2ornt(axcodes) -> np.ndarray:
    ornt = np.ones((len(axcodes), 2), dtype=np.int8) * np.nan
    for i, code in enumerate(axcodes):
        for j, codes in enumerate(_LABELS):
            if code in codes:
                ornt[i, :] = [j, -1 if code == codes[0] else 1]
                break
        else:
            raise ValueError(f"unknown axis code {code!r}")
    return ornt


def ornt2axcodes(ornt) -> tuple:
    return tuple(None if np.isnan(ax) else _LABELS[int(ax)][0 if d == -1 else 1] for ax, d in np.asarray(ornt))


def aff2axcodes(affine) -> tuple:
    return ornt2axcodes(io_orientation(affine))


def ornt_transform(start_ornt, end_ornt) -> np.ndarray:
    """Orientation that takes an array in `start_ornt` to `end_ornt`."""
    start_ornt, end_ornt = np.asarray(start_ornt), np.asarray(end_ornt)
    if start_ornt.shape != end_ornt.shape:
        raise ValueError("The orientations must have the same shape")
    result = np.empty_like(start_ornt)
    for end_in, (end_out, end_flip) in enumerate(end_ornt):
        for start_in, (start_out, start_flip) in enumerate(start_ornt):
            if end_out == start_out:
                result[start_in, :] = [end_in, 1 if start_flip == end_flip else -1]
                break
        else:
            raise ValueError(f"Unable to find out axis {end_out} in start_ornt")
    return result


def apply_orientation(arr: np.ndarray, ornt) -> np.ndarray:
    """Flip, then transpose so that input axis i ends up at output axis ornt[i,0] (a view, not a copy)."""
    t = np.asarray(arr)
    ornt = np.asarray(ornt)
    if np.any(np.isnan(ornt[:, 0])):
        raise ValueError("Cannot drop coordinates when applying orientation to data")
    for ax, flip in enumerate(ornt[:, 1]):
        if flip == -1:
            t = np.flip(t, axis=ax)
    full = np.arange(t.ndim)
    full[:ornt.shape[0]] = np.argsort(ornt[:, 0])
    return t.transpose(full)


def inv_ornt_aff(ornt, shape) -> np.ndarray:
    """Affine mapping voxel coordinates of the reoriented array back to those of the original array."""
    ornt = np.asarray(ornt)
    if np.any(np.isnan(ornt)):
        raise ValueError("We cannot invert orientation transform")
    p = ornt.shape[0]
    shape = np.array(shape)[:p]
    axis_transpose = [int(v) for v in ornt[:, 0]]
    undo_reorder = np.eye(p + 1)[axis_transpose + [p], :]
    undo_flip = np.diag(list(ornt[:, 1]) + [1.0])
    center = -(shape - 1) / 2.0
    undo_flip[:p, p] = (ornt[:, 1] * center) - center
    return np.dot(undo_flip, undo_reorder)


def reorient(arr: np.ndarray, affine: np.ndarray, ornt):
    """`SpatialImage.as_reoriented`: (array view, new affine)."""
    if np.array_equal(ornt, RAS_ORNT):
        return arr, affine
    return apply_orientation(arr, ornt), np.asarray(affine, dtype=np.float64).dot(inv_ornt_aff(ornt, arr.shape))


def as_closest_canonical(arr: np.ndarray, affine: np.ndarray):
    """TS/alignment.py:8-12 -> (array in RAS+ order, affine, orientation that was applied)."""
    ornt = io_orientation(affine)
    a, aff = reorient(arr, affine, ornt)
    return a, aff, ornt


def undo_canonical(arr_can: np.ndarray, orig_affine: np.ndarray) -> np.ndarray:
    """TS/alignment.py:24-46: back from RAS+ to the original image's axis order."""
    img_ornt = io_orientation(orig_affine)
    return apply_orientation(arr_can, ornt_transform(RAS_ORNT, img_ornt))


def with_axcodes(arr: np.ndarray, affine: np.ndarray, axcodes="RAS"):
    """BCA/io.py:97-113 `load_nibabel_image_with_axcodes`."""
    cur = "".join(aff2axcodes(affine))
    axcodes = "".join(axcodes)
    if cur == axcodes:
        return arr, affine
    return reorient(arr, affine, ornt_transform(axcodes2ornt(cur), axcodes2ornt(axcodes)))


def zooms_from_affine(affine) -> np.ndarray:
    """Voxel sizes as a NIfTI header reports them (float32 pixdim of the column norms)."""
    a = np.asarray(affine, dtype=np.float64)[:3, :3]
    return np.sqrt(np.sum(a * a, axis=0)).astype(np.float32)
