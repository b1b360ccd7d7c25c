"""Device resampler vs the reference's scipy.ndimage.zoom outputs (golden G5) and vs scipy itself."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    from boa_hip.device import Context
    c = Context(0)
    yield c
    c.close()


CASES = ["half", "twothirds", "thick", "up2", "aniso"]


@pytest.mark.parametrize("k", CASES)
def test_g5_order0_exact(ctx, k):
    from boa_hip import resample
    z = np.load(os.path.join(GOLD, "g5_resample.npz"))
    out = resample.resample_img(ctx, z["lab"], z[f"zoom_{k}"], 0)
    np.testing.assert_array_equal(out, z[f"lab0_{k}"])


@pytest.mark.parametrize("k", CASES)
def test_g5_order3_int32_exact(ctx, k):
    """`.astype(np.int32)` of the cubic zoom, bit for bit against the reference's scipy output (golden G5)."""
    from boa_hip import resample
    z = np.load(os.path.join(GOLD, "g5_resample.npz"))
    out = resample.resample_img(ctx, z["ct"], z[f"zoom_{k}"], 3, out_dtype=np.int32)
    np.testing.assert_array_equal(out, z[f"ct3_{k}"])


@pytest.mark.parametrize("in_dtype", [np.int16, np.float32, np.float64, np.int32])
def test_order3_fp64_bits_vs_scipy(ctx, in_dtype):
    from scipy import ndimage
    from boa_hip import resample
    rng = np.random.default_rng(5)
    x = (rng.normal(size=(37, 45, 29)) * 400).astype(in_dtype)
    for zoom in [(0.5, 0.5, 0.5), (1.0, 1.0, 0.3), (1.31, 0.77, 2.0)]:
        ref = ndimage.zoom(x.astype(np.float64), zoom, order=3, mode="nearest")
        out = resample.resample_img(ctx, x, zoom, 3)
        np.testing.assert_array_equal(out.view(np.uint64), ref.view(np.uint64))
    # last coordinate one ulp past the input extent (47 -> 43 samples), see tests/test_oracle_golden.py
    x = (rng.normal(size=(10, 47, 14)) * 400).astype(in_dtype)
    zoom = (0.589242864991995, 0.9166715392459119, 1.27427627251354)
    ref = ndimage.zoom(x.astype(np.float64), zoom, order=3, mode="nearest")
    np.testing.assert_array_equal(resample.resample_img(ctx, x, zoom, 3).view(np.uint64), ref.view(np.uint64))


def test_change_spacing_matches_oracle(ctx):
    from oracle import resample as oresample
    from boa_hip import resample
    from boa_hip.synthetic import ct_phantom
    ct = ct_phantom((64, 64, 40), seed=3)
    new, zoom = resample.change_spacing_array(ctx, ct.astype(np.float64), (0.9, 0.9, 3.0), 1.5, order=3, dtype=np.int32)
    ref, rzoom = oresample.change_spacing_array(ct, (0.9, 0.9, 3.0), 1.5, order=3, dtype=np.int32)
    np.testing.assert_array_equal(zoom, rzoom)
    assert new.shape == ref.shape and new.dtype == np.int32
    np.testing.assert_array_equal(new, ref)   # constant -1024 background: exact-integer values everywhere
    same, z0 = resample.change_spacing_array(ctx, ct, (1.5, 1.5, 1.5), 1.5)
    assert z0 is None and same is ct
    # target_shape path (labels back to the original grid, TS/nnunet.py:685-687)
    lab = (ct > 0).astype(np.uint8)
    up, _ = resample.change_spacing_array(ctx, lab, (1.5, 1.5, 1.5), target_shape=(96, 96, 60), order=0)
    rup, _ = oresample.change_spacing_array(lab, (1.5, 1.5, 1.5), target_shape=(96, 96, 60), order=0, dtype=np.uint8)
    np.testing.assert_array_equal(up, rup)
