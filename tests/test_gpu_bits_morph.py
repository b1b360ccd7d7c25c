"""GPU parity tests of the bit-mask morphology (csrc/ccl_bits.hip, through the C ABI): select / unpack / contour fill / batched
small-object and small-hole removal / largest-component filter / label assign against scipy.ndimage (the restatement of the
skimage / cv2 calls of BCA/body_parts/postprocess.py:7-52 and BCA/body_regions/postprocess.py:8-40) and against the
byte-mask kernels (boa_ccl26 path) they replace.  Everything here is integer work: bit-exact."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

S26 = np.ones((3, 3, 3), bool)


@pytest.fixture(scope="module")
def ctx():
    from boa_hip.device import Context
    c = Context(0)
    yield c
    c.close()


def _blobs(rng, shape, n_blobs, rmax):
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    m = np.zeros(shape, bool)
    for _ in range(n_blobs):
        c = [rng.integers(0, s) for s in shape]
        r = rng.integers(1, rmax)
        ball = (zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2 <= r * r
        if rng.random() < 0.35:
            m &= ~ball
        else:
            m |= ball
    return m


def _pack(ctx, masks):
    """list of bool [Z][Y][X] -> device bit masks via boa_bits_select on a label volume (mask j <-> bit j of the label value)"""
    from boa_hip._lib import check
    Z, Y, X = masks[0].shape
    seg = np.zeros((Z, Y, X), np.uint8)
    for j, m in enumerate(masks):
        seg |= (m.astype(np.uint8) << j)
    lut = np.arange(256, dtype=np.uint8)          # label value v belongs to mask j iff bit j of v
    d_seg = ctx.from_numpy(seg)
    words = int(ctx.lib.boa_bits_words(Z, Y, X))
    d_bits = ctx.alloc(words * 4 * len(masks))
    check(ctx.lib.boa_bits_select(ctx.h, d_seg.vp, Z, Y, X, lut.ctypes.data_as(C.c_void_p), len(masks), d_bits.vp), "boa_bits_select")
    d_seg.free()
    return d_bits, words


def _unpack(ctx, d_bits, words, shape, j):
    from boa_hip._lib import check
    from boa_hip.device import BufferView
    Z, Y, X = shape
    d_o = ctx.alloc(Z * Y * X)
    check(ctx.lib.boa_bits_unpack(ctx.h, BufferView(d_bits, j * words * 4, words * 4).vp, Z, Y, X, d_o.vp), "boa_bits_unpack")
    out = d_o.download(shape, np.uint8).astype(bool)
    d_o.free()
    return out


def _remove_small_ref(m, max_size):
    from scipy import ndimage
    lab, k = ndimage.label(m, structure=S26)
    if k == 0:
        return m.copy()
    cnt = np.bincount(lab.ravel())
    small = cnt <= max_size
    small[0] = False
    return m & ~small[lab]


SHAPES = [(33, 35, 70), (16, 16, 32), (17, 49, 33), (5, 3, 100), (48, 32, 64), (1, 1, 7), (20, 37, 31)]


@pytest.mark.parametrize("shape", SHAPES)
def test_select_unpack_roundtrip_and_word_layout(ctx, shape):
    rng = np.random.default_rng(sum(shape) + 1)
    masks = [rng.random(shape) < p for p in (0.1, 0.5, 0.9)] + [np.ones(shape, bool), np.zeros(shape, bool)]
    d_bits, words = _pack(ctx, masks)
    assert words == shape[0] * shape[1] * ((shape[2] + 31) // 32)
    raw = d_bits.download((len(masks), shape[0], shape[1], (shape[2] + 31) // 32), np.uint32)
    for j, m in enumerate(masks):
        np.testing.assert_array_equal(_unpack(ctx, d_bits, words, shape, j), m)
        # bit i of word w <-> x = 32 w + i; padding bits are zero
        x = np.arange(((shape[2] + 31) // 32) * 32)
        bits = ((raw[j][..., x // 32] >> (x % 32).astype(np.uint32)) & 1).astype(bool)
        np.testing.assert_array_equal(bits[..., :shape[2]], m)
        assert not bits[..., shape[2]:].any()
    d_bits.free()


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("invert", [0, 1])
def test_remove_small_batched_vs_scipy(ctx, shape, invert):
    """remove_small_objects(max_size, connectivity 3) on a batch of masks of very different structure (noise of five densities: the
    union-find's worst case; blobs; full and empty: the uniform-tile fast paths), on shapes that cut the 32 x 16 x 16 tiles raggedly;
    invert: the same on the complements (small holes are filled)."""
    from boa_hip._lib import check
    rng = np.random.default_rng(sum(shape) + 7 * invert)
    masks = [rng.random(shape) < p for p in (0.03, 0.15, 0.4, 0.75, 0.97)] + [_blobs(rng, shape, 25, max(2, min(shape) // 2 + 2)),
                                                                            np.ones(shape, bool), np.zeros(shape, bool)]
    for max_size in (1, 9, 200):
        d_bits, words = _pack(ctx, masks)
        check(ctx.lib.boa_bits_remove_small(ctx.h, d_bits.vp, shape[0], shape[1], shape[2], len(masks), max_size, invert), "boa_bits_remove_small")
        for j, m in enumerate(masks):
            got = _unpack(ctx, d_bits, words, shape, j)
            want = ~_remove_small_ref(~m, max_size) if invert else _remove_small_ref(m, max_size)
            np.testing.assert_array_equal(got, want, err_msg=f"mask {j} max_size {max_size} invert {invert}")
        d_bits.free()


def test_remove_small_components_spanning_many_tiles(ctx):
    """Components that only connect through other tiles (a thin 26-connected diagonal staircase through 4 x 3 x 3 tiles, a hollow box
    whose inside is a hole) and the size threshold right at / one below the component sizes."""
    from boa_hip._lib import check
    shape = (40, 50, 130)
    m = np.zeros(shape, bool)
    for i in range(38):
        m[1 + i, 2 + i, 3 + 3 * i:6 + 3 * i] = True   # staircase of 3-voxel runs, consecutive runs touch diagonally (dz = dy = dx = 1):
                                                        # one component of 114 voxels through 3 x 3 x 4 tiles
    m[5:30, 20:45, 10:50] = True
    m[8:27, 23:42, 13:47] = False                   # hollow box: a closed hole of 19 * 19 * 34 voxels
    m[0, 49, 129] = True                            # isolated corner voxel
    from scipy import ndimage
    lab, k = ndimage.label(m, structure=S26)
    sizes = sorted(np.bincount(lab.ravel())[1:])
    assert sizes[0] == 1 and 114 in sizes
    for max_size in (0, 1, 113, 114, 10 ** 6):
        d_bits, words = _pack(ctx, [m])
        check(ctx.lib.boa_bits_remove_small(ctx.h, d_bits.vp, *shape, 1, max_size, 0))
        np.testing.assert_array_equal(_unpack(ctx, d_bits, words, shape, 0), _remove_small_ref(m, max_size))
        d_bits.free()
    hole = 19 * 19 * 34
    for max_size in (hole - 1, hole):
        d_bits, words = _pack(ctx, [m])
        check(ctx.lib.boa_bits_remove_small(ctx.h, d_bits.vp, *shape, 1, max_size, 1))
        got = _unpack(ctx, d_bits, words, shape, 0)
        np.testing.assert_array_equal(got, ~_remove_small_ref(~m, max_size))
        assert got[10, 30, 20] == (max_size >= hole)
        d_bits.free()


@pytest.mark.parametrize("shape", [(9, 61, 53), (3, 16, 32), (2, 5, 100), (4, 40, 64)])
def test_fill_holes_2d_bits_vs_scipy(ctx, shape):
    from scipy import ndimage
    from boa_hip._lib import check
    rng = np.random.default_rng(11 + shape[2])
    a = _blobs(rng, shape, 60, 12)
    a[0] = False
    if shape[1] > 12 and shape[2] > 13:
        a[0, 10, 10] = a[0, 11, 11] = a[0, 10, 12] = a[0, 9, 11] = True       # diamond ring: closed for the (8-connected contour) fill
    a[1] = rng.random(shape[1:]) < 0.45
    b = rng.random(shape) < 0.6
    c = np.ones(shape, bool)
    c[:, 1:-1, 1:-1] = False                                                    # frame: everything inside is filled
    masks = [a, b, c, np.zeros(shape, bool)]
    d_in, words = _pack(ctx, masks)
    d_out = ctx.alloc(words * 4 * len(masks))
    assert ctx.lib.boa_bits_fill_supported(shape[1], shape[2]) == 1
    check(ctx.lib.boa_bits_fill_holes_2d(ctx.h, d_in.vp, *shape, len(masks), d_out.vp), "boa_bits_fill_holes_2d")
    for j, m in enumerate(masks):
        ref = np.stack([ndimage.binary_fill_holes(m[i]) for i in range(shape[0])])
        np.testing.assert_array_equal(_unpack(ctx, d_out, words, shape, j), ref)
    d_in.free()
    d_out.free()


def test_filter_largest_vs_scipy_with_ties(ctx):
    """All components but the largest -> fill value; equal sizes: the component whose first voxel comes first in raster order stays
    (the reference sorts regionprops by area with a stable sort: lowest label first)."""
    from scipy import ndimage
    from boa_hip._lib import check
    rng = np.random.default_rng(5)
    shape = (34, 40, 70)
    cases = []
    m = np.zeros(shape, bool)
    m[2:6, 2:6, 2:6] = True
    m[20:24, 30:34, 60:64] = True           # two cubes of 64 voxels in different tiles: the first one stays
    m[10, 10, 40:45] = True
    cases.append(m)
    cases.append(_blobs(rng, shape, 30, 9))
    cases.append(rng.random(shape) < 0.2)
    cases.append(np.zeros(shape, bool))
    one = np.zeros(shape, bool)
    one[3:30, 3:30, 3:60] = True             # a single component: nothing changes
    cases.append(one)
    for m in cases:
        seg = (m.astype(np.uint8) * 3)
        seg[~m] = (rng.random(shape)[~m] < 0.1) * 9          # other labels around it stay untouched
        d_seg = ctx.from_numpy(seg)
        d_bits, words = _pack(ctx, [m])
        check(ctx.lib.boa_bits_filter_largest(ctx.h, d_bits.vp, *shape, d_seg.vp, 255), "boa_bits_filter_largest")
        got = d_seg.download(shape, np.uint8)
        lab, k = ndimage.label(m, structure=S26)
        want = seg.copy()
        if k > 1:
            cnt = np.bincount(lab.ravel())[1:]
            keep = int(np.argmax(cnt)) + 1                     # argmax: first maximum = lowest label
            want[m & (lab != keep)] = 255
        np.testing.assert_array_equal(got, want)
        d_seg.free()
        d_bits.free()


def test_assign_labels_overlay_order(ctx):
    from boa_hip._lib import check
    rng = np.random.default_rng(9)
    shape = (6, 20, 75)
    masks = [rng.random(shape) < 0.3 for _ in range(5)]
    labels = np.array([1, 2, 4, 5, 9], np.uint8)
    d_bits, words = _pack(ctx, masks)
    out0 = (rng.random(shape) < 0.2).astype(np.uint8) * 77     # earlier content survives where no mask has the voxel
    d_out = ctx.from_numpy(out0)
    check(ctx.lib.boa_bits_assign_labels(ctx.h, d_bits.vp, *shape, len(masks), labels.ctypes.data_as(C.c_void_p), d_out.vp))
    want = out0.copy()
    for m, v in zip(masks, labels):
        want[m] = v
    np.testing.assert_array_equal(d_out.download(shape, np.uint8), want)
    d_bits.free()
    d_out.free()


def test_part_and_region_postprocess_bits_equal_bytes_and_oracle(ctx, monkeypatch):
    """The product functions on the bit path, on the byte path ($BOA_MORPH_BYTES=1) and the oracle give the same label volumes;
    13 labels exercise two batches of the part post-processing."""
    from boa_hip import bca
    from oracle import bca as obca
    rng = np.random.default_rng(21)
    shape = (26, 72, 70)
    seg = np.zeros(shape, np.uint8)
    for label in range(1, 14):
        seg[_blobs(rng, shape, 6, 12)] = label
    seg[rng.random(shape) < 0.01] = 4
    for thr in (3000, 300):
        ref = obca.remove_small_labeled_objects(seg, threshold=thr)
        monkeypatch.delenv("BOA_MORPH_BYTES", raising=False)
        np.testing.assert_array_equal(bca.postprocess_part_segmentation(ctx, seg, threshold=thr), ref)
        monkeypatch.setenv("BOA_MORPH_BYTES", "1")
        np.testing.assert_array_equal(bca.postprocess_part_segmentation(ctx, seg, threshold=thr), ref)
    rseg = (seg % 12).astype(np.uint8)
    want = obca.postprocess_region_segmentation(rseg)
    monkeypatch.delenv("BOA_MORPH_BYTES", raising=False)
    np.testing.assert_array_equal(bca.postprocess_region_segmentation(ctx, rseg), want)
    monkeypatch.setenv("BOA_MORPH_BYTES", "1")
    np.testing.assert_array_equal(bca.postprocess_region_segmentation(ctx, rseg), want)
    assert (want == 255).any()
