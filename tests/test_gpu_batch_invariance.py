"""A tile's logits must not depend on which other tiles share its conv-stack launch: InstanceNorm statistics are reduced
per sample over virtual workgroups whose tile runs are a function of the layer geometry alone (k_conv_ws `tile_walk`,
k_conv_first_mfma).  Checked bit for bit on the per-tile logits (`network_forward`) at tile batch 1 / 3 / 8 and on whole
label volumes; the tile-sharded == unsharded consequence is in test_gpu_tile_shard.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from boa_hip.device import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("patch,features,vol", [
    ((32, 32, 32), (32, 64, 128), (44, 40, 48)),          # one tile run per virtual workgroup
    ((64, 64, 64), (32, 64), (80, 64, 96)),               # 512 spatial tiles per sample at full resolution: runs of 2 tiles
    ((32, 64, 128), (32, 64, 128, 256), (40, 64, 160)),   # anisotropic, 4 stages, tiles % virtual workgroups != 0
])
def test_tile_logits_independent_of_batch(ctx, patch, features, vol):
    from boa_hip import plans, sliding_window as sw
    from boa_hip.predictor import HipPredictor
    pj, dj = plans.synthetic_plans(patch=patch, features=features, num_classes=5)
    geom = plans.model_config_from_plans(pj, dj).geometry
    blob = plans.weight_blob_from_state_dict(geom, plans.synthetic_state_dict(geom, 3))
    x = np.random.default_rng(1).standard_normal((1, *vol)).astype(np.float32)
    origins = np.array(sw.get_sliding_window_origins(list(vol), list(patch), 0.5), dtype=np.int32)[:8]
    outs = {}
    for mb in (1, 3, 8):
        p = HipPredictor(ctx, geom, max_batch=mb)
        p.set_parameters([blob])
        outs[mb] = p.network_forward(x, origins)
        p.close()
    cnt = ctx.counters()
    assert cnt["conv_ws"] > 0 and cnt["first_mfma"] > 0 and cnt["head_mfma"] > 0, cnt   # the production kernels ran
    assert np.isfinite(outs[1]).all() and np.ptp(outs[1]) > 1.0
    np.testing.assert_array_equal(outs[1].view(np.uint32), outs[3].view(np.uint32))
    np.testing.assert_array_equal(outs[1].view(np.uint32), outs[8].view(np.uint32))
    # and a tile gives the same logits wherever it sits in the list
    p = HipPredictor(ctx, geom, max_batch=4)
    p.set_parameters([blob])
    rev = p.network_forward(x, origins[::-1].copy())
    p.close()
    np.testing.assert_array_equal(rev[::-1].view(np.uint32), outs[1].view(np.uint32))


def test_label_volume_independent_of_batch(ctx):
    from boa_hip import plans
    from boa_hip.predictor import HipPredictor
    pj, dj = plans.synthetic_plans(patch=(32, 32, 64), features=(32, 64, 128), num_classes=7)
    geom = plans.model_config_from_plans(pj, dj).geometry
    blobs = [plans.weight_blob_from_state_dict(geom, plans.synthetic_state_dict(geom, s)) for s in (1, 2)]
    x = np.random.default_rng(2).standard_normal((1, 70, 50, 128)).astype(np.float32)
    labs = []
    for mb in (1, 4, 8):
        p = HipPredictor(ctx, geom, tile_step_size=0.5, max_batch=mb)
        p.set_parameters(blobs)                                            # two folds
        labs.append(p.predict_segmentation(x))
        p.close()
    assert len(np.unique(labs[0])) > 2
    np.testing.assert_array_equal(labs[0], labs[1])
    np.testing.assert_array_equal(labs[0], labs[2])
