"""Brute-force flood fills, written from the DEFINITIONS (no scipy, no library labelling): the second anchor of the connected-component
and contour-fill stages, whose reference implementations (skimage.measure.label, cv2.findContours / drawContours) are absent from
this image.  Test infrastructure only; O(voxels) python loops -- small volumes."""
import numpy as np


def components26(mask: np.ndarray):
    """26-connected components of a 3-D boolean mask by explicit flood fill: (roots, sizes) with roots[v] = the smallest linear
    index (C order) of v's component for foreground voxels, -1 for background; sizes = {root: voxel count}.
    Definition: two foreground voxels are neighbours when every coordinate differs by at most 1 (skimage connectivity=3 /
    BCA/body_regions/postprocess.py:9 `measure.label(mask)`: full connectivity is the default)."""
    m = np.asarray(mask, dtype=bool)
    Z, Y, X = m.shape
    roots = np.full(m.shape, -1, dtype=np.int64)
    sizes = {}
    flat = m.ravel()
    for start in range(flat.size):          # ascending linear index: the first voxel met of a component is its smallest index
        if not flat[start] or roots.flat[start] >= 0:
            continue
        stack = [start]
        roots.flat[start] = start
        count = 0
        while stack:
            v = stack.pop()
            count += 1
            z, r = divmod(v, Y * X)
            y, x = divmod(r, X)
            for dz in (-1, 0, 1):
                zz = z + dz
                if zz < 0 or zz >= Z:
                    continue
                for dy in (-1, 0, 1):
                    yy = y + dy
                    if yy < 0 or yy >= Y:
                        continue
                    for dx in (-1, 0, 1):
                        xx = x + dx
                        if xx < 0 or xx >= X:
                            continue
                        w = (zz * Y + yy) * X + xx
                        if flat[w] and roots.flat[w] < 0:
                            roots.flat[w] = start
                            stack.append(w)
        sizes[start] = count
    return roots, sizes


def fill_external_contours(sl: np.ndarray) -> np.ndarray:
    """cv2.findContours(RETR_EXTERNAL) + drawContours(FILLED) on one slice, from the definition: the outer contours bound the
    8-connected foreground objects; filling them sets every pixel that the background cannot reach from outside the image by
    4-connected steps (a diagonal chain of foreground pixels is closed for a 4-connected walker).  = foreground plus enclosed
    background.  Explicit border flood, no library call (BCA/body_parts/postprocess.py:31-38)."""
    m = np.asarray(sl, dtype=bool)
    Y, X = m.shape
    reached = np.zeros((Y + 2, X + 2), dtype=bool)          # a one-pixel frame of outside around the slice
    fg = np.zeros((Y + 2, X + 2), dtype=bool)
    fg[1:-1, 1:-1] = m
    stack = [(0, 0)]
    reached[0, 0] = True
    while stack:
        y, x = stack.pop()
        for yy, xx in ((y - 1, x), (y + 1, x), (y, x - 1), (y, x + 1)):
            if 0 <= yy < Y + 2 and 0 <= xx < X + 2 and not reached[yy, xx] and not fg[yy, xx]:
                reached[yy, xx] = True
                stack.append((yy, xx))
    return ~reached[1:-1, 1:-1]
