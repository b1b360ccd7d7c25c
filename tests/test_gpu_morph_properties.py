"""Device connected components / contour fill against brute-force flood fills written from the definitions (tests/floodfill.py) --
no scipy in the loop: the library-free anchor of boa_ccl26 / boa_fill_holes_2d / the BCA post-processing."""
import ctypes as C

import numpy as np
import pytest

from floodfill import components26, fill_external_contours

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from boa_hip.device import Context
    c = Context(0)
    yield c
    c.close()


def _volume(rng, shape, p):
    a = rng.random(shape)
    sm = (a + np.roll(a, 1, 0) + np.roll(a, 1, 1) + np.roll(a, 1, 2) + np.roll(a, -1, 2)) / 5.0
    return sm > np.quantile(sm, 1 - p)


@pytest.mark.parametrize("shape,p", [((20, 24, 40), 0.3), ((7, 33, 35), 0.5), ((34, 18, 33), 0.12), ((3, 5, 70), 0.6)])
def test_ccl26_roots_and_sizes_equal_flood_fill(ctx, shape, p):
    from boa_hip._lib import check
    rng = np.random.default_rng(sum(shape))
    m = _volume(rng, shape, p)
    m[0, 0, 0] = m[1, 1, 1] = True                                      # a purely diagonal contact
    roots_want, sizes_want = components26(m)
    n = m.size
    d_m, d_roots, d_sizes = ctx.from_numpy(m.astype(np.uint8)), ctx.alloc(n * 4), ctx.alloc(n * 4)
    ncomp = C.c_int()
    check(ctx.lib.boa_ccl26(ctx.h, d_m.vp, shape[0], shape[1], shape[2], d_roots.vp, d_sizes.vp, C.byref(ncomp)), "boa_ccl26")
    roots = d_roots.download(shape, np.int32).astype(np.int64)
    sizes = d_sizes.download((n,), np.uint32)
    for d in (d_m, d_roots, d_sizes):
        d.free()
    assert ncomp.value == len(sizes_want)
    np.testing.assert_array_equal(roots, roots_want)                      # root = smallest linear index of the component
    for r, cnt in sizes_want.items():
        assert int(sizes[r]) == cnt


def test_fill_holes_2d_equals_border_flood(ctx):
    from boa_hip._lib import check
    rng = np.random.default_rng(5)
    shape = (6, 37, 45)
    m = np.stack([_volume(rng, (1, *shape[1:]), p)[0] for p in (0.2, 0.35, 0.5, 0.65, 0.45, 0.3)])
    m[0, 5, 5] = m[0, 6, 6] = m[0, 5, 7] = m[0, 4, 6] = True             # diamond ring
    m[0, 5, 6] = False
    m[1, 0, :] = True                                                    # a wall on the border
    want = np.stack([fill_external_contours(s) for s in m])
    n = m.size
    d_m = ctx.from_numpy(m.astype(np.uint8))
    d_i, d_t, d_o = ctx.alloc(n * 4), ctx.alloc(n), ctx.alloc(n)
    check(ctx.lib.boa_fill_holes_2d(ctx.h, d_m.vp, shape[0], shape[1], shape[2], d_i.vp, d_t.vp, d_o.vp))
    got = d_o.download(shape, np.uint8).astype(bool)
    for d in (d_m, d_i, d_t, d_o):
        d.free()
    np.testing.assert_array_equal(got, want)
    assert got[0, 5, 6] and want.sum() > m.sum()


def test_region_postprocess_equals_flood_fill_rules(ctx):
    """BCA/body_regions/postprocess.py:18-40 from the definition: for the four masks in turn, every 26-connected component but
    the largest becomes 255 (ties: the component met first in scan order stays)."""
    from boa_hip import bca
    REG = bca.REGION if hasattr(bca, "REGION") else None
    from oracle.bca import REGION
    rng = np.random.default_rng(9)
    shape = (12, 20, 24)
    seg = np.zeros(shape, np.uint8)
    vals = [REGION["THORACIC_CAVITY"], REGION["MEDIASTINUM"], REGION["PERICARDIUM"], REGION["ABDOMINAL_CAVITY"], 1]
    for v in vals:
        seg[_volume(rng, shape, 0.12)] = v
    want = seg.copy()

    def filt(mask):
        roots, sizes = components26(mask)
        order = sorted(sizes, key=lambda r: (-sizes[r], r))
        for r in order[1:]:
            want[roots == r] = 255

    filt(want > 0)
    filt((want == REGION["THORACIC_CAVITY"]) | (want == REGION["MEDIASTINUM"]) | (want == REGION["PERICARDIUM"]))
    filt(want == REGION["PERICARDIUM"])
    filt(want == REGION["ABDOMINAL_CAVITY"])
    got = bca.postprocess_region_segmentation(ctx, seg)
    np.testing.assert_array_equal(got, want)
    assert (want == 255).any()
