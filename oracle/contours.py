"""TEST INFRASTRUCTURE (oracle): a second, independent restatement of the reference's slice-wise contour fill

    contours, _ = cv2.findContours(slice_mask, cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE)
    cv2.drawContours(filled[i], contours, -1, color=1, thickness=cv2.FILLED)          (BCA/body_parts/postprocess.py:31-39)

written from OpenCV's documented algorithms, NOT from connectivity arguments -- cv2 is absent from this image, so this pins nothing
against cv2 itself (PARITY UNPINNED); what it gives is a differential check of the product's formulation ("foreground + all background
that is not 4-connected to the slice border", oracle/bca.py, csrc/ccl_bits.hip) against the border-following + polygon-fill formulation:

  * findContours = Suzuki & Abe, "Topological structural analysis of digitized binary images by border following" (CVGIP 30, 1985),
    Algorithm 1: raster scan, outer borders start at a 1-pixel whose left neighbour is 0, hole borders at a >= 1-pixel whose right
    neighbour is 0; 1-components are 8-connected, 0-components 4-connected; the frame around the image is the background.  RETR_EXTERNAL
    keeps the outer borders whose parent is the frame.  (CHAIN_APPROX_SIMPLE only drops collinear chain points: the polygon is the same.)
  * drawContours(thickness=FILLED) = every contour's polygon edges collected into one edge table, filled by the even-odd scanline rule at
    pixel centres, and the polygon outlines drawn on top, so that every border pixel is set (drawing.cpp: CollectPolyEdges draws the
    edges with Line, FillEdgeCollection fills between pairs of crossings).
Pure-Python loops: small masks only."""
import numpy as np

# 8-neighbourhood in CLOCKWISE order starting at west, as (di, dj) with i = row (down), j = column (right)
_CW = [(0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1)]


def _dir_index(di, dj):
    return _CW.index((di, dj))


def find_contours_external(mask: np.ndarray):
    """-> list of closed pixel chains [(i, j), ...] (outer borders of the outermost 8-connected components, in following order)."""
    m = np.asarray(mask) != 0
    H, W = m.shape
    f = np.zeros((H + 2, W + 2), dtype=np.int64)      # the zero frame
    f[1:-1, 1:-1] = m
    nbd = 1
    is_hole = {1: True}                               # border number -> hole border?  (1 = the frame, a hole border by definition)
    parent = {1: 0}
    chains = {}
    for i in range(1, H + 1):
        lnbd = 1
        for j in range(1, W + 1):
            if f[i, j] == 0:
                continue
            start = None
            if f[i, j] == 1 and f[i, j - 1] == 0:     # (1a) outer border
                nbd += 1
                start = (i, j - 1)
                hole = False
            elif f[i, j] >= 1 and f[i, j + 1] == 0:   # (1b) hole border
                nbd += 1
                start = (i, j + 1)
                hole = True
                if f[i, j] > 1:
                    lnbd = int(f[i, j])
            if start is not None:
                # (2) parent from the border numbered lnbd (table 1 of the paper)
                bprime_hole = is_hole[lnbd]
                if hole:
                    parent[nbd] = parent[lnbd] if bprime_hole else lnbd
                else:
                    parent[nbd] = lnbd if bprime_hole else parent[lnbd]
                is_hole[nbd] = hole
                chains[nbd] = _follow(f, i, j, start, nbd)
            # (4)
            if f[i, j] != 1:
                lnbd = abs(int(f[i, j]))
    out = []
    for b in sorted(chains):
        if not is_hole[b] and parent[b] == 1:
            out.append([(i - 1, j - 1) for (i, j) in chains[b]])
    return out


def _follow(f, i, j, start, nbd):
    """Step (3) of Algorithm 1; marks f in place, returns the chain of border pixels in the order they are visited."""
    # (3.1) clockwise from `start` around (i, j): first non-zero neighbour
    d0 = _dir_index(start[0] - i, start[1] - j)
    i1 = j1 = None
    for k in range(8):
        di, dj = _CW[(d0 + k) % 8]
        if f[i + di, j + dj] != 0:
            i1, j1 = i + di, j + dj
            break
    if i1 is None:
        f[i, j] = -nbd
        return [(i, j)]
    i2, j2 = i1, j1
    i3, j3 = i, j
    chain = []
    while True:
        # (3.3) counter-clockwise from the element after (i2, j2) around (i3, j3): first non-zero neighbour
        d = _dir_index(i2 - i3, j2 - j3)
        east_zero_examined = False
        i4 = j4 = None
        for k in range(1, 9):
            di, dj = _CW[(d - k) % 8]                  # counter-clockwise = backwards through the clockwise table
            if f[i3 + di, j3 + dj] != 0:
                i4, j4 = i3 + di, j3 + dj
                break
            if (di, dj) == (0, 1):
                east_zero_examined = True
        chain.append((i3, j3))
        # (3.4)
        if east_zero_examined:
            f[i3, j3] = -nbd
        elif f[i3, j3] == 1:
            f[i3, j3] = nbd
        # (3.5)
        if (i4, j4) == (i, j) and (i3, j3) == (i1, j1):
            return chain
        i2, j2 = i3, j3
        i3, j3 = i4, j4


def draw_contours_filled(chains, shape) -> np.ndarray:
    """Even-odd scanline fill of all polygons together (vertices = pixel centres) + the outlines themselves."""
    H, W = shape
    cross = np.zeros((H, W + 1), dtype=np.int64)      # cross[y, x]: polygon edges crossing row y at column x (half-open in y)
    out = np.zeros((H, W), dtype=bool)
    for ch in chains:
        n = len(ch)
        for k in range(n):
            (y0, x0), (y1, x1) = ch[k], ch[(k + 1) % n]
            out[y0, x0] = True                         # the outline (unit steps: Line() sets exactly the chain pixels)
            if y0 == y1:
                continue                               # horizontal edges never cross a scanline in the half-open rule
            # a unit step spans the rows [min, max): it is counted on its upper row, at the x of its end point on that row
            ylo, xlo = (y0, x0) if y0 < y1 else (y1, x1)
            cross[ylo, xlo] += 1
    # pixel (y, x) is inside iff the number of crossings at columns > x is odd; boundary pixels are set by the outline anyway
    total = cross.sum(axis=1, keepdims=True)
    upto = np.cumsum(cross, axis=1)[:, :W]             # crossings at columns <= x
    inside = ((total - upto) % 2) == 1
    return out | inside


def fill_external_contours(mask2d: np.ndarray) -> np.ndarray:
    """The reference's two cv2 calls on one slice -> bool mask."""
    m = np.asarray(mask2d)
    return draw_contours_filled(find_contours_external(m), m.shape)
