"""Differential pin of the slice-wise contour fill (BCA/body_parts/postprocess.py:31-39; cv2 absent: PARITY UNPINNED vs cv2 itself).

Two independent formulations are compared on adversarial and random 2-D masks:
  A. border following + polygon fill from OpenCV's documented algorithms (oracle/contours.py: Suzuki-Abe outer borders of the 8-connected
     foreground whose parent is the frame, even-odd scanline fill of all polygons + their outlines);
  B. the formulation the product kernels implement: foreground + background that a 4-connected walker cannot reach from outside the slice
     (tests/floodfill.py explicit flood; scipy.ndimage.binary_fill_holes with the cross element, which the GPU tests compare the kernels to).
Disagreement classes searched for (each has its own generator below): 1-pixel inlets, diagonal necks and diagonal-only seals, blobs touching
the slice border, nested rings (a component inside another's hole), checkerboards / salt-and-pepper, one-pixel-wide spirals and spikes,
single pixels and empty / full slices.  Outcome: NONE of the classes produces a disagreement -- A == B on every mask here; what remains
unpinned is cv2's fixed-point edge rounding in FillEdgeCollection, which cannot matter for polygons whose vertices are pixel centres and
whose edges are unit steps (every crossing is at an integer column)."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import contours

from floodfill import fill_external_contours as flood_fill


def _check(m):
    m = np.asarray(m, dtype=bool)
    a = contours.fill_external_contours(m)
    b = flood_fill(m)
    c = ndimage.binary_fill_holes(m) if m.size else m
    assert (a == b).all(), f"border following != flood on\n{m.astype(int)}\nA\n{a.astype(int)}\nB\n{b.astype(int)}"
    assert (b == c).all()
    assert (a | m == a).all()           # the fill never removes foreground


def _ring(n, t=1):
    m = np.zeros((n, n), bool)
    m[t:n - t, t:n - t] = True
    m[2 * t:n - 2 * t, 2 * t:n - 2 * t] = False      # wall thickness t
    return m


def test_trivial_slices():
    for shape in [(1, 1), (1, 7), (7, 1), (5, 6)]:
        _check(np.zeros(shape, bool))
        _check(np.ones(shape, bool))
    one = np.zeros((5, 5), bool)
    one[2, 2] = True
    _check(one)
    one[0, 0] = one[4, 4] = one[0, 4] = True
    _check(one)


def test_one_pixel_inlet_keeps_the_pocket_open():
    m = _ring(11)
    _check(m)
    assert contours.fill_external_contours(m)[5, 5]
    m[1, 5] = False                      # a one-pixel channel through the wall: the pocket is outside now
    _check(m)
    assert not contours.fill_external_contours(m)[5, 5]
    thick = _ring(15, 2)
    thick[2:4, 7] = False                # channel of width 1 through a wall of thickness 2
    _check(thick)
    assert not contours.fill_external_contours(thick)[7, 7]


def test_diagonal_seal_closes_the_pocket():
    # the wall is only DIAGONALLY connected at one corner: 8-connected foreground = closed for the 4-connected background
    m = _ring(9)
    m[1, 1] = False                      # remove the corner: (1, 2) and (2, 1) stay, touching diagonally
    _check(m)
    assert contours.fill_external_contours(m)[4, 4]
    # a diamond drawn with diagonal steps only
    d = np.zeros((11, 11), bool)
    for k in range(5):
        d[5 - k, k + 1 - 1 + 0] = True
    d[:] = False
    c = 5
    for k in range(-4, 5):
        d[c + k, c + (4 - abs(k))] = True
        d[c + k, c - (4 - abs(k))] = True
    _check(d)
    assert contours.fill_external_contours(d)[5, 5]
    # two diagonal necks in a row, and a neck onto the slice border
    n = np.zeros((8, 8), bool)
    n[0, 3] = n[1, 2] = n[2, 1] = n[3, 0] = True          # a diagonal from border to border cuts off the corner pocket
    _check(n)
    assert contours.fill_external_contours(n)[0, 0] == flood_fill(n)[0, 0]


def test_border_touching_blobs():
    m = np.zeros((10, 12), bool)
    m[0:4, 0:5] = True
    m[1:3, 1:4] = False                  # a hole in a blob that sits in the corner
    m[6:10, 3:9] = True
    m[7:10, 4:8] = False                 # a "U" open towards the slice border: its mouth is outside
    _check(m)
    f = contours.fill_external_contours(m)
    assert f[1, 1] and not f[8, 5]
    full_frame = np.ones((9, 9), bool)
    full_frame[1:-1, 1:-1] = False       # foreground along the whole slice border: everything inside is enclosed
    _check(full_frame)
    assert contours.fill_external_contours(full_frame).all()


def test_nested_rings_and_islands():
    m = _ring(21, 1)
    m[5:16, 5:16] |= _ring(11, 1)        # a ring inside the ring's hole
    m[10, 10] = True                     # an island inside both
    _check(m)
    assert contours.fill_external_contours(m)[1:20, 1:20].all()
    assert len(contours.find_contours_external(m)) == 1     # RETR_EXTERNAL: the inner components are children of a hole


def test_checkerboards_spirals_spikes():
    yy, xx = np.mgrid[0:12, 0:13]
    _check((yy + xx) % 2 == 0)           # every background pixel is diagonally sealed: one 8-connected component, all filled except border cells
    _check((yy + xx) % 2 == 1)
    _check((yy % 2 == 0) & (xx % 2 == 0))
    sp = np.zeros((15, 15), bool)        # a one-pixel-wide square spiral: no enclosed background at all
    y = x = 0
    dy, dx = 0, 1
    n = 15
    for seg in range(14, 0, -2):
        for _ in range(2):
            for _ in range(seg):
                sp[y, x] = True
                y += dy
                x += dx
            dy, dx = dx, -dy
    _check(sp)
    spike = np.zeros((9, 9), bool)
    spike[4, :] = True
    spike[:, 4] = True                   # a cross of one-pixel lines: the border walks out and back along every arm
    _check(spike)
    assert (contours.fill_external_contours(spike) == spike).all()


@pytest.mark.parametrize("density", [0.05, 0.2, 0.4, 0.5, 0.593, 0.7, 0.9])
def test_random_masks(density):
    rng = np.random.default_rng(int(density * 1000))
    for _ in range(120):
        H, W = rng.integers(1, 28, 2)
        _check(rng.random((H, W)) < density)


def test_random_blobs_with_structure():
    rng = np.random.default_rng(7)
    for _ in range(80):
        H, W = rng.integers(8, 40, 2)
        m = ndimage.binary_dilation(rng.random((H, W)) < 0.06, iterations=int(rng.integers(1, 4)))
        m &= ~ndimage.binary_dilation(rng.random((H, W)) < 0.03, iterations=1)      # punch holes and inlets
        m ^= rng.random((H, W)) < 0.02                                               # salt and pepper on top
        _check(m)


def test_external_contours_are_closed_unit_step_chains():
    rng = np.random.default_rng(3)
    for _ in range(50):
        m = rng.random((16, 18)) < 0.55
        lab, n8 = ndimage.label(m, structure=np.ones((3, 3)))
        chains = contours.find_contours_external(m)
        holes_filled = ndimage.binary_fill_holes(m)
        _, n_outer = ndimage.label(holes_filled, structure=np.ones((3, 3)))
        assert len(chains) == n_outer                      # one external contour per outermost 8-connected component
        for ch in chains:
            for k in range(len(ch)):
                (y0, x0), (y1, x1) = ch[k], ch[(k + 1) % len(ch)]
                assert m[y0, x0] and max(abs(y0 - y1), abs(x0 - x1)) <= 1
