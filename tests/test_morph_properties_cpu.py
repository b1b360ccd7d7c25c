"""The scipy restatements of the oracle's connected-component / contour-fill stages (skimage and cv2 are absent: SURVEY 8c "parity
unpinned") against brute-force flood fills written from the definitions (tests/floodfill.py): a second, library-free anchor."""
import numpy as np
from scipy import ndimage

from floodfill import components26, fill_external_contours


def _volume(rng, shape, p):
    sm = ndimage.uniform_filter(rng.random(shape), 3)
    return sm > np.quantile(sm, 1 - p)


def test_scipy_label_26_equals_flood_fill():
    rng = np.random.default_rng(0)
    for shape, p in (((9, 14, 17), 0.25), ((6, 20, 11), 0.5), ((12, 12, 12), 0.08), ((3, 5, 40), 0.6)):
        m = _volume(rng, shape, p)
        m[0, 0, 0] = m[1, 1, 1] = True                       # a purely diagonal contact
        roots, sizes = components26(m)
        lab, k = ndimage.label(m, structure=np.ones((3, 3, 3)))
        assert k == len(sizes)
        # same partition: a bijection between scipy's labels and the flood fill's roots
        pairs = set(zip(lab[m].tolist(), roots[m].tolist()))
        assert len(pairs) == k and len({a for a, _ in pairs}) == k and len({b for _, b in pairs}) == k
        cnt = np.bincount(lab.ravel())
        for a, b in pairs:
            assert cnt[a] == sizes[b]
            assert b == int(np.flatnonzero(lab.ravel() == a)[0])           # root = the component's smallest linear index


def test_oracle_region_filter_equals_flood_fill_rule():
    """filter_largest_unique_segment (BCA/body_regions/postprocess.py:8-15): every component but the largest -> 255."""
    from oracle import bca as obca
    rng = np.random.default_rng(1)
    m = _volume(rng, (10, 16, 18), 0.3)
    seg = np.where(m, 3, 0).astype(np.uint8)
    got = seg.copy()
    obca.filter_largest_unique_segment(got, got > 0)
    roots, sizes = components26(m)
    order = sorted(sizes, key=lambda r: (-sizes[r], r))          # largest first; equal areas keep ascending label order
    want = seg.copy()
    for r in order[1:]:
        want[roots == r] = 255
    np.testing.assert_array_equal(got, want)
    assert (want == 255).any() and (want == 3).any()


def test_binary_fill_holes_cross_equals_border_flood():
    """The oracle restates cv2's per-slice external-contour fill as scipy's binary_fill_holes (cross element); both must equal
    the explicit 4-connected border flood, incl. diagonal rings and objects touching the border."""
    rng = np.random.default_rng(2)
    for shape, p in (((31, 29), 0.45), ((17, 40), 0.3), ((25, 25), 0.6)):
        m = _volume(rng, (1, *shape), p)[0]
        m[5, 5] = m[6, 6] = m[5, 7] = m[4, 6] = True            # diamond ring: (5, 6) is enclosed for a 4-connected walker
        m[5, 6] = False
        m[0, :] |= rng.random(shape[1]) < 0.5                    # foreground on the border
        want = fill_external_contours(m)
        np.testing.assert_array_equal(ndimage.binary_fill_holes(m), want)
        assert want[5, 6]


def test_even_footprint_erosion_side_is_pinned():
    """erode_region (BOA/compute/measurements.py:61-71; pyproject.toml:46 "different results for kernel 6"): the reference pads its 6^3
    footprint AT THE END to 7^3 before skimage sees it, i.e. ones at offsets -3 .. +2 around the centre (7 // 2 = 3) on every axis -- the
    old skimage behaviour it says it preserves (newer skimage pads an even footprint at the START: -2 .. +3).  skimage is absent (parity
    unpinned); this writes the side down: a voxel survives iff the 6 voxels p - 3 .. p + 2 are set on every axis (outside the volume
    counts as set), checked on an asymmetric blob against a brute-force window."""
    from oracle import measurements as om
    m = np.zeros((20, 22, 24), bool)
    m[4:12, 5:13, 6:20] = True            # 8 x 8 x 14 box
    m[4:12, 5:13, 6:9] &= True
    m[8:12, 13:20, 6:20] = True           # an L: second arm 4 thick along axis 0 -> it cannot survive a 6-wide window
    got = om.erode_region(m)
    pad = np.ones(tuple(s + 6 for s in m.shape), bool)     # border_value True: the outside does not erode
    pad[3:-3, 3:-3, 3:-3] = m
    want = np.zeros_like(m)
    for p in np.argwhere(m):
        z, y, x = p + 3
        want[tuple(p)] = pad[z - 3:z + 3, y - 3:y + 3, x - 3:x + 3].all()
    assert (got == want).all()
    # the 8-wide box leaves 3 survivors per short axis, at box start + 3 .. + 5: shifted half a voxel towards HIGHER indices
    zs, ys, xs = np.nonzero(got)
    assert (zs.min(), zs.max()) == (7, 9) and (ys.min(), ys.max()) == (8, 10) and (xs.min(), xs.max()) == (9, 17)
    # the mirrored convention (-2 .. +3) would give 6..8 / 7..9 / 8..16: it must NOT be what the oracle computes
    k = np.zeros((7, 7, 7), bool)
    k[1:, 1:, 1:] = True
    other = ndimage.binary_erosion(m, structure=k, border_value=1)
    assert not (other == got).all() and np.nonzero(other)[0].min() == 6
