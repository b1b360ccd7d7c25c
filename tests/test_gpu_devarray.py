"""Device views + boa_copy3 / boa_nonzero_bbox vs numpy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from boa_hip.device import Context
    c = Context(0)
    yield c
    c.close()


def test_views_match_numpy(ctx):
    from boa_hip.devarray import DevArray
    rng = np.random.default_rng(0)
    a = rng.integers(-1024, 3000, size=(7, 11, 13)).astype(np.int16)
    d = DevArray.from_numpy(ctx, a)
    np.testing.assert_array_equal(d.download(), a)
    np.testing.assert_array_equal(d.transpose((2, 1, 0)).download(), a.transpose(2, 1, 0))
    np.testing.assert_array_equal(d.flip(0).flip(2).transpose((1, 2, 0)).download(), a[::-1, :, ::-1].transpose(1, 2, 0))
    np.testing.assert_array_equal(d.slice(1, 2, 9).slice(2, 3, -2).flip(1).download(), a[:, 2:9, 3:-2][:, ::-1])
    np.testing.assert_array_equal(d.box([[1, 6], [0, 11], [4, 5]]).contiguous(np.float32).download(),
                                  a[1:6, :, 4:5].astype(np.float32))
    # scatter into a sub-box of a larger array with conversion
    full = DevArray.zeros(ctx, (9, 12, 15), np.int32)
    d.transpose((0, 1, 2)).copy_to(full.box([[1, 8], [1, 12], [2, 15]]))
    ref = np.zeros((9, 12, 15), np.int32)
    ref[1:8, 1:12, 2:15] = a
    np.testing.assert_array_equal(full.download(), ref)
    # float -> int truncation like numpy astype
    f = (rng.standard_normal((4, 5, 6)) * 100).astype(np.float64)
    np.testing.assert_array_equal(DevArray.from_numpy(ctx, f).contiguous(np.int32).download(), f.astype(np.int32))


def test_apply_orientation_view_matches_host(ctx):
    import itertools
    from boa_hip import orientation as o
    from boa_hip.devarray import DevArray
    a = np.arange(4 * 5 * 6, dtype=np.int32).reshape(4, 5, 6)
    d = DevArray.from_numpy(ctx, a)
    for perm in itertools.permutations(range(3)):
        for flips in itertools.product([1, -1], repeat=3):
            ornt = np.array([[perm[i], flips[i]] for i in range(3)], dtype=float)
            np.testing.assert_array_equal(d.apply_orientation(ornt).download(), o.apply_orientation(a, ornt))


@pytest.mark.parametrize("dtype", [np.int16, np.int32, np.float32])
def test_nonzero_bbox(ctx, dtype):
    from boa_hip.devarray import DevArray
    from boa_hip.task import nonzero_bbox
    rng = np.random.default_rng(1)
    a = np.zeros((20, 33, 17), dtype)
    assert DevArray.from_numpy(ctx, a).nonzero_bbox() == [[0, 20], [0, 33], [0, 17]]
    a[3:15, 7:30, 2:9] = rng.integers(1, 5, size=(12, 23, 7))
    a[16, 31, 12] = -3
    assert DevArray.from_numpy(ctx, a).nonzero_bbox() == nonzero_bbox(a)


def test_nonzero_bbox_g12(ctx):
    """boa_nonzero_bbox against the masks nnU-Net's own create_nonzero_mask produced (golden G12)."""
    import os
    from conftest import GOLDEN
    from boa_hip.devarray import DevArray
    z = np.load(os.path.join(GOLDEN, "g12_cropping.npz"))
    for j in range(int(z["n_cases"][1])):
        d, mask = z[f"n{j}_data"][0], z[f"n{j}_mask"].astype(bool)
        want = []
        for ax in range(3):
            nz = np.flatnonzero(mask.any(axis=tuple(a for a in range(3) if a != ax)))
            want.append([0, mask.shape[ax]] if nz.size == 0 else [int(nz[0]), int(nz[-1]) + 1])
        assert DevArray.from_numpy(ctx, d).nonzero_bbox() == want, j
