"""The network AT THE GEOMETRY THE BENCH RUNS ("what's weak" #1): `total` part model 291 -- patch 128^3, six
stages, features 32/64/128/256/320/320, 25 classes (NN/utilities/plans_handling/plans_handler.py:59-92, forward call
NN/inference/predict_from_raw_data.py:543) -- and the 5 mm BCA net geometry (7 classes), tile forwards on the device in
production (fp16 MFMA) and exact (fp32) mode against `oracle.network` (torch-CPU fp32 PlainConvUNet, ~3 s per tile on 8 threads).

At 128^3 the tile chooser picks the R=4 row-reuse `k_conv_ws` variants, the stride-2 R=1 variants, resident vs streamed weights
and the 8^3 / 4^3 bottleneck tiles that the 32^3 toy nets of test_gpu_seams.py never reach; the launch counters assert that the
production kernels (not the fallbacks) ran.  Bars = measured on MI355X + margin (printed by the test):
  exact mode : max |logit error| <= 2e-5 of the logit range (measured 1.9e-6 .. 3.6e-6), label flips <= 2e-5 of the voxels
               (measured 4.3e-6 .. 8.6e-6 = 9 .. 18 of 2 097 152 voxels: fp32 summation-order near-ties of random-weight nets)
  fp16 mode  : max |logit error| <= 2.7e-3 of the logit range (measured 1.2e-3 .. 2.1e-3: worst case + 25 %), label flips <= 7e-3
               (measured 2.5e-3 .. 5.7e-3; random weights put far more voxels at near-ties than trained nets); every flipped voxel's
               oracle top-2 margin is below twice the logit error, and the test prints the margin histogram of the flipped voxels.
test_structured_net_* repeats the fp16 comparison on the closest stand-in for a TRAINED net this image allows (smooth features, a
confident head): see there."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PATCH = (128, 128, 128)
FEATURES = (32, 64, 128, 256, 320, 320)
# fp16 flips of the structured (confident) net: bar = measured on MI355X + margin (the test prints the measured value)
STRUCTURED_FP16_FLIP_BAR = 4.5e-3   # measured 3.5e-3


@pytest.fixture(scope="module")
def ctx():
    from boa_hip.device import Context
    c = Context(0)
    yield c
    c.close()


def _model(num_classes, seed, spacing=(1.5, 1.5, 1.5)):
    import torch
    from boa_hip import plans
    from oracle.network import build_from_arch
    pj, dj = plans.synthetic_plans(patch=PATCH, features=FEATURES, num_classes=num_classes, spacing=spacing)
    cfg = plans.model_config_from_plans(pj, dj)
    sd = plans.synthetic_state_dict(cfg.geometry, seed)
    blob = plans.weight_blob_from_state_dict(cfg.geometry, sd)
    net = build_from_arch(pj["configurations"]["3d_fullres"]["architecture"]["arch_kwargs"], 1, num_classes)
    missing = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not missing.missing_keys, missing
    return cfg, blob, net


def _normalised_phantom(cfg, shape, seed):
    from boa_hip import synthetic
    ct = synthetic.ct_phantom(list(shape), seed=seed).astype(np.float32)
    ip = cfg.intensity_properties["0"]
    return ((np.clip(ct, ip["percentile_00_5"], ip["percentile_99_5"]) - ip["mean"]) / max(ip["std"], 1e-8)).astype(np.float32)[None]


def _compare(ctx, cfg, blob, net, vol, origins, tag):
    from boa_hip.predictor import HipPredictor
    from oracle.network import network_fn_from_module
    fn = network_fn_from_module(net, threads=8)
    refs = []
    for o in origins:
        refs.append(fn(vol[:, o[0]:o[0] + PATCH[0], o[1]:o[1] + PATCH[1], o[2]:o[2] + PATCH[2]][None])[0])
    out = {}
    for prec, err_bar, flip_bar in (("fp32", 2e-5, 2e-5), ("fp32_ref", 2e-5, 2e-5), ("fp16", 2.7e-3, 7e-3)):
        ctx.counters(reset=True)
        p = HipPredictor(ctx, cfg.geometry, max_batch=len(origins), precision=prec)
        p.set_parameters([blob])
        got = p.network_forward(vol, np.asarray(origins, dtype=np.int32))
        p.close()
        cnt = ctx.counters()
        if prec == "fp32":   # the split-precision kernels (k_conv_ws<X3> etc.), not the fp32 reference mode
            assert cnt["conv_x3"] > 0 and cnt["x3"] > 0 and cnt["f32"] == 0 and cnt["conv_ws"] == 0, cnt
        if prec == "fp16":   # the production kernels, not their fallbacks
            assert cnt["conv_ws"] > 0 and cnt["first_mfma"] > 0 and cnt["head_mfma"] > 0, cnt
            assert cnt["conv_simple"] == 0 and cnt["head_valu"] == 0 and cnt["first_valu"] == 0, cnt
        for i, ref in enumerate(refs):
            rng_ = float(ref.max() - ref.min())
            err = float(np.abs(got[i] - ref).max())
            flips = float((got[i].argmax(0) != ref.argmax(0)).mean())
            print(f"{tag} {prec} tile {i}: max|err| {err:.4g} = {err / rng_:.3g} of the range {rng_:.4g}; label flips {flips:.3g}")
            assert np.isfinite(got[i]).all()
            assert err <= err_bar * rng_, (prec, err, rng_)
            assert flips <= flip_bar, (prec, flips)
            # where the flips sit: a label can only change where the oracle's winner leads its runner-up by less than twice the
            # logit error (near-ties of the random head), never on a confident voxel
            flipped = got[i].argmax(0) != ref.argmax(0)
            if flipped.any():
                top2 = np.partition(ref, -2, axis=0)[-2:]
                worst = float((top2[1] - top2[0])[flipped].max())
                print(f"{tag} {prec} tile {i}: largest oracle top-2 margin of a flipped voxel {worst / rng_:.3g} of the range")
                assert worst <= 2.0 * err + 1e-6 * rng_, (prec, worst, err)
                if prec == "fp16":   # where the flips sit, and how many voxels live there at all
                    mg = (top2[1] - top2[0]) / rng_
                    edges = [0, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 1.0]
                    hf, _ = np.histogram(mg[flipped], bins=edges)
                    ha, _ = np.histogram(mg, bins=edges)
                    print(f"{tag} fp16 tile {i}: flipped / all voxels by oracle top-2 margin (units of the logit range): "
                          + ", ".join(f"[{edges[k]:g}, {edges[k + 1]:g}) {hf[k]} / {ha[k]}" for k in range(len(hf))))
        out[prec] = got
    return out


def test_part_model_291_tiles_vs_oracle(ctx):
    """Two 128^3 tiles of the phantom (one body-interior, one crossing the body surface and the air background)."""
    cfg, blob, net = _model(25, 291)
    vol = _normalised_phantom(cfg, (160, 160, 192), seed=7)
    _compare(ctx, cfg, blob, net, vol, [(16, 16, 32), (32, 0, 64)], "model 291")


def test_bca_geometry_tile_vs_oracle(ctx):
    """body_parts net geometry of the bench (7 classes, plans spacing 5 mm slices), one tile of Gaussian noise."""
    cfg, blob, net = _model(7, 543, spacing=(5.0, 1.5, 1.5))
    vol = np.random.default_rng(543).standard_normal((1, 128, 136, 144)).astype(np.float32)
    _compare(ctx, cfg, blob, net, vol, [(0, 8, 16)], "body_parts")


def test_structured_net_fp16_flips_vs_oracle(ctx):
    """The fp16 label-flip fraction on a net whose logits look like a segmentation's rather than like noise (the real
    checkpoints are not available offline): `plans.synthetic_state_dict(structured=0.05)` -- every spatial kernel is a random channel
    mixing times a smoothing stencil, so the last decoder features are smooth multi-scale functions of the CT -- and a CONFIDENT
    two-class head (w1 = -w0: one template, for and against), 128^3 production geometry, a tile that crosses the body surface.
    Measured on the torch-CPU oracle: 98 % of the voxels have a top-2 margin above 1e-3 of the logit range, 94 % above 3e-3, 80 %
    above 1e-2 (the random 25-class head of the bench: 90 / 74 / 37 %).  Asserted: the margin distribution (so that the test keeps
    measuring what it says), the fp16 logit error bar of the production test, NO flip above twice the logit error, and the flip
    fraction (measured 3.5e-3 + 25 %).  What it shows: the flip fraction is (voxels whose top-2 margin is below ~the logit error) x
    ~1/2 -- with a logit error of 1.9e-3 of the range, 6 % of this net's voxels sit below a margin of 3e-3 and 0.35 % flip; a
    confident head alone does not move that, because LeakyReLU nets with synthetic weights have a margin density that is FLAT at
    zero.  A trained checkpoint pushes its margin density at zero down (that is what the loss does), and its flip rate follows the
    same product; that density is the one number this image cannot supply (no weights offline).  The label contract therefore
    rests on the fp32 mode (flips 1.4e-6 here), not on an extrapolation of this test."""
    import torch
    from boa_hip import plans
    from boa_hip.predictor import HipPredictor
    from oracle.network import build_from_arch, network_fn_from_module
    pj, dj = plans.synthetic_plans(patch=PATCH, features=FEATURES, num_classes=2)
    cfg = plans.model_config_from_plans(pj, dj)
    sd = plans.synthetic_state_dict(cfg.geometry, 291, structured=0.05)
    head = max(int(k.split(".")[2]) for k in sd if k.startswith("decoder.seg_layers.") and k.endswith(".weight"))
    q = np.random.default_rng(1).standard_normal(FEATURES[0]).astype(np.float32)
    q /= np.linalg.norm(q)
    sd[f"decoder.seg_layers.{head}.weight"] = np.stack([q, -q]).reshape(2, FEATURES[0], 1, 1, 1)
    sd[f"decoder.seg_layers.{head}.bias"] = np.zeros(2, np.float32)
    blob = plans.weight_blob_from_state_dict(cfg.geometry, sd)
    net = build_from_arch(pj["configurations"]["3d_fullres"]["architecture"]["arch_kwargs"], 1, 2)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    vol = _normalised_phantom(cfg, (160, 160, 192), seed=7)
    o = (16, 16, 32)
    ref = network_fn_from_module(net, threads=8)(vol[:, o[0]:o[0] + 128, o[1]:o[1] + 128, o[2]:o[2] + 128][None])[0]
    rng_ = float(ref.max() - ref.min())
    mg = np.abs(ref[1] - ref[0]) / rng_
    frac = {t: float((mg > t).mean()) for t in (1e-3, 3e-3, 1e-2)}
    print("structured net: fraction of voxels with oracle top-2 margin above 1e-3 / 3e-3 / 1e-2 of the range:", frac)
    assert frac[1e-3] >= 0.97 and frac[3e-3] >= 0.92 and frac[1e-2] >= 0.75, frac
    res = {}
    for prec in ("fp16", "fp32"):
        p = HipPredictor(ctx, cfg.geometry, max_batch=1, precision=prec)
        p.set_parameters([blob])
        got = p.network_forward(vol, np.asarray([o], dtype=np.int32))[0]
        p.close()
        err = float(np.abs(got - ref).max())
        flipped = got.argmax(0) != ref.argmax(0)
        worst = float(mg[flipped].max()) if flipped.any() else 0.0
        res[prec] = (err / rng_, float(flipped.mean()), worst)
        print(f"structured net {prec}: max|err| {err / rng_:.3g} of the range {rng_:.4g}, label flips {flipped.mean():.3g}, "
              f"largest margin of a flipped voxel {worst:.3g} of the range")
        assert worst * rng_ <= 2.0 * err + 1e-6 * rng_, (prec, worst, err)
    assert res["fp16"][0] <= 2.7e-3 and res["fp32"][0] <= 2e-5, res
    assert res["fp32"][1] <= 2e-5, res
    assert res["fp16"][1] <= STRUCTURED_FP16_FLIP_BAR, res


def test_tile_batch_16_and_25_bit_identical_to_1(ctx):
    """The product default tile batch (16) and the bench's (25) give the same per-tile logits, bit for bit, as batch 1
    (test_gpu_batch_invariance.py covers 1 / 3 / 8 on small nets)."""
    from boa_hip import plans
    from boa_hip.predictor import HipPredictor
    pj, dj = plans.synthetic_plans(patch=(64, 64, 64), features=(32, 64, 128, 256), num_classes=5)
    geom = plans.model_config_from_plans(pj, dj).geometry
    blob = plans.weight_blob_from_state_dict(geom, plans.synthetic_state_dict(geom, 9))
    x = np.random.default_rng(4).standard_normal((1, 96, 96, 112)).astype(np.float32)
    rng = np.random.default_rng(5)
    origins = np.stack([rng.integers(0, 33, 25), rng.integers(0, 33, 25), rng.integers(0, 49, 25)], axis=1).astype(np.int32)
    outs = {}
    for mb in (1, 16, 25):
        p = HipPredictor(ctx, geom, max_batch=mb)
        p.set_parameters([blob])
        outs[mb] = p.network_forward(x, origins)
        p.close()
    assert np.isfinite(outs[1]).all() and np.ptp(outs[1]) > 1.0
    np.testing.assert_array_equal(outs[1].view(np.uint32), outs[16].view(np.uint32))
    np.testing.assert_array_equal(outs[1].view(np.uint32), outs[25].view(np.uint32))


def test_tile_batch_25_at_128_bit_identical_to_4(ctx):
    """Same at the production geometry: 25 tiles of part model 291 in one launch sequence vs batches of 4."""
    from boa_hip.predictor import HipPredictor
    cfg, blob, _ = _model(25, 291)
    vol = _normalised_phantom(cfg, (160, 160, 160), seed=11)
    rng = np.random.default_rng(6)
    origins = rng.integers(0, 33, size=(25, 3)).astype(np.int32)
    sums = {}
    for mb in (4, 25):
        p = HipPredictor(ctx, cfg.geometry, max_batch=mb)
        p.set_parameters([blob])
        got = p.network_forward(vol, origins)
        p.close()
        sums[mb] = got
    np.testing.assert_array_equal(sums[4].view(np.uint32), sums[25].view(np.uint32))
