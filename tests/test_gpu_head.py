"""The PRODUCTION head kernel (`k_head_mfma`: 1x1x1 head on the matrix cores + x Gaussian + fp16 read-modify-write of the
class planes -- what boa_net_predict_sliding_window and bench.py launch per tile) against the oracle's accumulate step
(NN/inference/predict_from_raw_data.py:611-614), bit for bit.

The kernel has two modes that share the instruction sequence up to the logit (`boa_head_tile`): logits mode writes the
fp32 logits of a tile, accumulate mode multiplies them by the Gaussian and adds them to the fp16 buffers.  Feeding the
logits-mode output of every tile to `oracle.sliding_window.accumulate_tile` in the reference's tile order must give the
device accumulators' bit patterns.  Every test asserts through `boa_debug_counter` that the MFMA kernel (not the fp32
VALU fallback `k_head<F0>`) is the one that ran.  The logit values themselves (fp16 weights / packed-fp16 norm) are
compared with an fp32 evaluation under the tolerance stated in `test_head_logits_vs_fp32`.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from boa_hip.device import Context
    c = Context(0)
    yield c
    c.close()


def _i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


class _Head:
    """Device state of one head: weights, bias, helper to run a tile in either mode."""

    def __init__(self, ctx, Cn, seed, F0=32):
        rng = np.random.default_rng(seed)
        self.ctx, self.C, self.F0 = ctx, Cn, F0
        self.w = (rng.standard_normal((Cn, F0)) * 0.4).astype(np.float32)
        self.b = (rng.standard_normal(Cn) * 0.5).astype(np.float32)
        self.d_w, self.d_b = ctx.from_numpy(self.w), ctx.from_numpy(self.b)

    @staticmethod
    def planar(act):
        """channels-last [P0][P1][P2][F0] -> the engine's chunk-planar layout [F0/16][P0][P1][P2][16]."""
        P, F0 = act.shape[:3], act.shape[3]
        return np.ascontiguousarray(act.reshape(*P, F0 // 16, 16).transpose(3, 0, 1, 2, 4))

    def tile_inputs(self, P, seed):
        """fp16 activation [P0][P1][P2][F0] (the last decoder conv's pre-norm output, channels-last here; `planar` gives the
        engine's layout) and its (scale, shift)."""
        rng = np.random.default_rng(seed)
        act = (rng.standard_normal((*P, self.F0)) * 1.5).astype(np.float16)
        ss = np.stack([rng.uniform(0.5, 1.5, self.F0), rng.normal(0, 0.3, self.F0)], axis=1).astype(np.float32)
        return act, ss

    def logits(self, d_act, d_ss, P):
        pv = int(np.prod(P))
        out = self.ctx.alloc(self.C * pv * 4)
        from boa_hip._lib import check
        check(self.ctx.lib.boa_head_tile(self.ctx.h, d_act.vp, d_ss.vp, self.F0, _i3(P), self.C, self.d_w.vp, self.d_b.vp, 0.01,
                                         out.vp, None, None, None, None, None), "boa_head_tile(logits)")
        res = out.download((self.C, *P), np.float32)
        out.free()
        return res

    def accumulate(self, d_act, d_ss, P, d_g, acc, n, PV, start):
        from boa_hip._lib import check
        check(self.ctx.lib.boa_head_tile(self.ctx.h, d_act.vp, d_ss.vp, self.F0, _i3(P), self.C, self.d_w.vp, self.d_b.vp, 0.01,
                                         None, d_g.vp if d_g is not None else None, acc.vp, n.vp, _i3(PV), _i3(start)),
              "boa_head_tile(accumulate)")

    def free(self):
        self.d_w.free()
        self.d_b.free()


def _run_case(ctx, PV, P, step, Cn, use_gaussian, seed, n_distinct=None):
    """Tile loop over the padded grid PV through the production head; returns device / oracle (acc, n) bit patterns."""
    from boa_hip import sliding_window as sw
    from oracle import sliding_window as osw
    origins = sw.get_sliding_window_origins(list(PV), list(P), step)
    assert all(int(o[2]) % 8 == 0 for o in origins) and PV[2] % 8 == 0 and P[2] % 32 == 0, "geometry must select k_head_mfma"
    head = _Head(ctx, Cn, seed)
    g16 = np.ascontiguousarray(sw.compute_gaussian(tuple(P), 1. / 8, 10)) if use_gaussian else None
    d_g = ctx.from_numpy(g16.view(np.uint16)) if use_gaussian else None
    nv = int(np.prod(PV))
    acc, n = ctx.zeros(Cn * nv * 2), ctx.zeros(nv * 2)
    o_acc, o_n = np.zeros((Cn, *PV), np.float16), np.zeros(PV, np.float16)
    ctx.counters(reset=True)
    n_distinct = n_distinct or len(origins)
    cache = {}
    for t, o in enumerate(origins):
        k = t % n_distinct
        if k not in cache:                                                   # a few distinct activations ...
            cache[k] = ctx.from_numpy(head.planar(head.tile_inputs(P, seed * 1000 + k)[0]).view(np.uint16))
        d_ss = ctx.from_numpy(head.tile_inputs((1, 1, 1), seed * 77 + t)[1])  # ... and every tile its own (scale, shift)
        L = head.logits(cache[k], d_ss, P)                                  # this tile's fp32 logits, from the kernel itself
        head.accumulate(cache[k], d_ss, P, d_g, acc, n, PV, o)
        osw.accumulate_tile(o_acc, o_n, L, g16, tuple(int(x) for x in o))    # reference arithmetic on those exact logits
        d_ss.free()
    ctx.sync()
    cnt = ctx.counters()
    assert cnt["head_mfma"] == 2 * len(origins) and cnt["head_valu"] == 0, cnt   # the production kernel ran, never the fallback
    got_acc, got_n = acc.download((Cn, *PV), np.uint16), n.download(tuple(PV), np.uint16)
    for b in [acc, n, *cache.values()] + ([d_g] if d_g is not None else []):
        b.free()
    head.free()
    return got_acc, got_n, o_acc.view(np.uint16), o_n.view(np.uint16), len(origins)


@pytest.mark.parametrize("case", [
    # G3-style geometries (tests/golden/g3: 40x36x33 / 12x40x20 with patch 16^3), last axis widened to what the MFMA head
    # needs (P2 % 32 == 0, z origins % 8 == 0); `b`: the padded grid of a volume thinner than the patch along axis 0
    dict(name="a", PV=(40, 36, 64), P=(16, 16, 32), step=0.5, C=3, gauss=True),
    dict(name="b-padded", PV=(16, 40, 64), P=(16, 16, 32), step=0.5, C=3, gauss=True),
    dict(name="c-25-classes", PV=(24, 20, 96), P=(16, 16, 32), step=0.5, C=25, gauss=True),
    dict(name="d-31-classes", PV=(20, 16, 64), P=(16, 16, 32), step=0.5, C=31, gauss=True),
    dict(name="e-no-gaussian", PV=(20, 20, 64), P=(16, 16, 32), step=0.5, C=2, gauss=False),
])
def test_production_head_accumulate_bit_exact(ctx, case):
    got_acc, got_n, want_acc, want_n, nt = _run_case(ctx, case["PV"], case["P"], case["step"], case["C"], case["gauss"], seed=11)
    assert nt >= 6
    np.testing.assert_array_equal(got_n, want_n)
    np.testing.assert_array_equal(got_acc, want_acc)


def test_production_head_accumulate_bit_exact_512(ctx):
    """configs[1] geometry: 512^3, patch 128^3, step 0.8 -> 125 overlapping tiles in the reference's x->y->z order, three
    classes of real (non-constant) logits per tile: the accumulators after the whole loop equal the oracle's sequential fp16
    `+=` at every one of the 4 x 134 M entries."""
    got_acc, got_n, want_acc, want_n, nt = _run_case(ctx, (512, 512, 512), (128, 128, 128), 0.8, 3, True, seed=5, n_distinct=3)
    assert nt == 125
    assert np.array_equal(got_n, want_n)
    assert np.array_equal(got_acc, want_acc)


def test_head_logits_vs_fp32(ctx):
    """Logit VALUES of the MFMA head against an fp32 evaluation of lrelu(x * scale + shift) @ W^T + b on the same fp16
    activations.  The kernel rounds W, scale and shift to fp16 and evaluates the norm as one fp16 fma: tolerance 2e-2
    absolute on logits of magnitude ~10 (measured ~6e-3), i.e. 2e-3 of the logit range."""
    P, Cn = (8, 8, 64), 25
    head = _Head(ctx, Cn, seed=3)
    act, ss = head.tile_inputs(P, 99)
    d_act, d_ss = ctx.from_numpy(head.planar(act).view(np.uint16)), ctx.from_numpy(ss)
    ctx.counters(reset=True)
    got = head.logits(d_act, d_ss, P)
    assert ctx.counters()["head_mfma"] == 1
    y = act.astype(np.float32) * ss[:, 0] + ss[:, 1]
    y = np.where(y > 0, y, np.float32(0.01) * y)
    want = np.einsum("xyzf,cf->cxyz", y.astype(np.float64), head.w.astype(np.float64)) + head.b[:, None, None, None]
    err = np.abs(got - want).max()
    rng_ = want.max() - want.min()
    print(f"head logits: max |err| {err:.3g} on a range of {rng_:.3g}")
    assert err <= 2e-2 and err <= 2e-3 * rng_
    d_act.free()
    d_ss.free()
    head.free()


def test_unaligned_z_origin_same_logits_bit_exact(ctx):
    """A tile whose z origin is not 8-voxel aligned cannot use the 16-byte accumulator accesses of `k_head_mfma`'s accumulate mode
    (the common case for real CT sizes: step 0.8 of 361 slices gives origins 0 / 78 / 155 / 233).  Round 3: the SAME MFMA logits
    (logits mode) + the generic accumulate step -- accumulators bit-identical to the oracle's accumulate on those logits, and to
    what an aligned tile with the same data would add (round 2 fell back to an fp32 VALU head with slightly different logits)."""
    from boa_hip import sliding_window as sw
    from oracle import sliding_window as osw
    P, PV, Cn = (16, 16, 32), (16, 16, 70), 4
    head = _Head(ctx, Cn, seed=8)
    act, ss = head.tile_inputs(P, 5)
    d_act, d_ss = ctx.from_numpy(head.planar(act).view(np.uint16)), ctx.from_numpy(ss)
    g16 = np.ascontiguousarray(sw.compute_gaussian(P, 1. / 8, 10))
    d_g = ctx.from_numpy(g16.view(np.uint16))
    nv = int(np.prod(PV))
    acc, n = ctx.zeros(Cn * nv * 2), ctx.zeros(nv * 2)
    L = head.logits(d_act, d_ss, P)
    o_acc, o_n = np.zeros((Cn, *PV), np.float16), np.zeros(PV, np.float16)
    ctx.counters(reset=True)
    for start in ((0, 0, 19), (0, 0, 3), (0, 0, 38)):          # overlapping, all unaligned
        head.accumulate(d_act, d_ss, P, d_g, acc, n, PV, start)
        osw.accumulate_tile(o_acc, o_n, L, g16, start)
    cnt = ctx.counters()
    assert cnt["head_valu"] == 0 and cnt["head_mfma"] == 3
    np.testing.assert_array_equal(n.download(PV, np.uint16), o_n.view(np.uint16))
    np.testing.assert_array_equal(acc.download((Cn, *PV), np.uint16), o_acc.view(np.uint16))
    for b in (d_act, d_ss, d_g, acc, n):
        b.free()
    head.free()
