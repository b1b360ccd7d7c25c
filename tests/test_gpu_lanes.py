"""boa_hip/lanes.py: `total` + total measurements on one context / stream and the BCA half on a second one of the same GPU must
deliver exactly what the one-stream run delivers (labels bit for bit, tables equal), run after run; a failure in the `total`
lane must surface on the calling thread and leave both contexts usable."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bca_models(folds):
    from boa_hip import plans
    out = {}
    for name, nc, seed in (("body_parts", 7, 543), ("body_regions", 12, 542)):
        pj, dj = plans.synthetic_plans(num_classes=nc, spacing=(5.0, 1.5, 1.5))
        cfg = plans.model_config_from_plans(pj, dj)
        out[name] = (cfg, [plans.weight_blob_from_state_dict(cfg.geometry, plans.synthetic_state_dict(cfg.geometry, seed + f))
                           for f in range(folds)])
    return out


def _collect(out):
    host = {k: out[k].download() for k in ("total", "body_parts", "body_regions", "tissues")}
    for k in host:
        out[k].free()
    return host, out["total_measurements"], out["bca_measurements"], out["vertebrae"]


def test_two_lanes_equal_one_stream():
    from boa_hip import label_maps, synthetic
    from boa_hip.devarray import DevArray
    from boa_hip.device import Context
    from boa_hip.lanes import TotalBcaRunner
    from boa_hip.pipeline import BcaPipelineHip
    from boa_hip.task import SegmentationTask
    shape = (224, 192, 256)
    ct = synthetic.ct_phantom(shape, seed=11)
    aff = np.diag([-1.5, -1.5, 1.5, 1.0])
    lm = label_maps.measurement_label_map("total")
    ctx_a, ctx_b, ctx_c = Context(0), Context(0), Context(0)
    try:
        parts = [(tid, cfg, [blob]) for tid, cfg, blob, _ in synthetic.total_part_models()]
        total = SegmentationTask(ctx_a, "total", parts, resample=1.5, multimodel=True, max_batch=8)
        bm = _bca_models(2)
        pipe_a = BcaPipelineHip(ctx_a, bm["body_parts"], bm["body_regions"], fast_bca=True, max_batch=8)
        pipe_b = BcaPipelineHip(ctx_b, bm["body_parts"], bm["body_regions"], fast_bca=True, max_batch=8)
        pipe_c = BcaPipelineHip(ctx_b, bm["body_parts"], bm["body_regions"], fast_bca=True, max_batch=8, parts_ctx=ctx_c)
        d_ct = DevArray.from_numpy(ctx_a, ct)
        one = TotalBcaRunner(total, pipe_a, lm)
        two = TotalBcaRunner(total, pipe_b, lm)
        three = TotalBcaRunner(total, pipe_c, lm)     # `total` | body_regions | body_parts
        assert not one.two_lanes and two.two_lanes and three.two_lanes
        ref = _collect(one.run_resident(d_ct, aff))
        assert len(np.unique(ref[0]["total"])) > 20 and ref[0]["tissues"].any()
        for run in (two, three, two, three):      # a race between the streams would show as run-to-run differences
            got = _collect(run.run_resident(d_ct, aff))
            for k in ref[0]:
                np.testing.assert_array_equal(got[0][k], ref[0][k], err_msg=k)
            assert got[1] == ref[1]
            assert got[2] == ref[2]
            assert got[3] == ref[3]
        # the BCA pipeline alone on two streams (host-array entry point)
        alone_a = pipe_a.run(ct, aff, total_seg=ref[0]["total"])
        alone_c = pipe_c.run(ct, aff, total_seg=ref[0]["total"])
        for k in ("body_parts", "body_regions", "tissues"):
            np.testing.assert_array_equal(alone_c[k], alone_a[k], err_msg=k)
            np.testing.assert_array_equal(alone_c[k], ref[0][k], err_msg=k)
        assert alone_c["bca_measurements"] == alone_a["bca_measurements"] == ref[2]
        # crop_body: body_regions waits for the body mask of the other stream
        crop_a = pipe_a.run(ct, aff, crop_body=True)
        crop_c = pipe_c.run(ct, aff, crop_body=True)
        for k in ("body_parts", "body_regions", "tissues"):
            np.testing.assert_array_equal(crop_c[k], crop_a[k], err_msg=k)
        assert crop_c["bca_measurements"] == crop_a["bca_measurements"]
        pipe_c.close()
        # the one-stream runner is unaffected by the second context's work
        again = _collect(one.run_resident(d_ct, aff))
        for k in ref[0]:
            np.testing.assert_array_equal(again[0][k], ref[0][k], err_msg=k)
        d_ct.free()
        total.close()
        pipe_a.close()
        pipe_b.close()
    finally:
        ctx_a.close()
        ctx_b.close()
        ctx_c.close()


def test_lane_error_reaches_the_caller():
    from boa_hip import label_maps, synthetic
    from boa_hip.devarray import DevArray
    from boa_hip.device import Context
    from boa_hip.lanes import TotalBcaRunner
    from boa_hip.pipeline import BcaPipelineHip
    from boa_hip.task import SegmentationTask
    shape = (128, 128, 128)
    ct = synthetic.ct_phantom(shape, seed=5)
    aff = np.diag([-1.5, -1.5, 1.5, 1.0])
    ctx_a, ctx_b = Context(0), Context(0)
    try:
        parts = [(tid, cfg, [blob]) for tid, cfg, blob, _ in synthetic.total_part_models()][:1]
        total = SegmentationTask(ctx_a, "total", parts, resample=1.5, multimodel=True, max_batch=4)
        bm = _bca_models(1)
        pipe_b = BcaPipelineHip(ctx_b, bm["body_parts"], bm["body_regions"], fast_bca=True, max_batch=4)
        run = TotalBcaRunner(total, pipe_b, label_maps.measurement_label_map("total"))
        d_ct = DevArray.from_numpy(ctx_a, ct)

        def boom(*a, **k):
            raise ValueError("lane A broke")

        good = run._total
        run._total = boom
        with pytest.raises(RuntimeError, match="lane A"):
            run.run_resident(d_ct, aff)
        run._total = good
        out = run.run_resident(d_ct, aff)      # both contexts still work
        host = _collect(out)[0]
        assert host["total"].shape == shape and host["body_parts"].shape == shape
        d_ct.free()
        total.close()
        pipe_b.close()
    finally:
        ctx_a.close()
        ctx_b.close()


def test_two_lanes_equal_one_stream_at_bench_size():
    """The configuration bench.py reports as `two_lanes` (512^3 @1.5 mm, five part models, both BCA nets with five folds): labels
    and tables of the two-stream run equal the one-stream run's."""
    from boa_hip import label_maps, synthetic
    from boa_hip.devarray import DevArray
    from boa_hip.device import Context
    from boa_hip.lanes import TotalBcaRunner
    from boa_hip.pipeline import BcaPipelineHip
    from boa_hip.task import SegmentationTask
    shape = (512, 512, 512)
    ct = synthetic.ct_phantom(shape, seed=20260928)
    aff = np.diag([-1.5, -1.5, 1.5, 1.0])
    lm = label_maps.measurement_label_map("total")
    ctx_a, ctx_b = Context(0), Context(0)
    try:
        parts = [(tid, cfg, [blob]) for tid, cfg, blob, _ in synthetic.total_part_models()]
        total = SegmentationTask(ctx_a, "total", parts, resample=1.5, multimodel=True, max_batch=16)
        bm = _bca_models(5)
        pipe_a = BcaPipelineHip(ctx_a, bm["body_parts"], bm["body_regions"], fast_bca=False, max_batch=16)
        pipe_b = BcaPipelineHip(ctx_b, bm["body_parts"], bm["body_regions"], fast_bca=False, max_batch=16)
        d_ct = DevArray.from_numpy(ctx_a, ct)
        ref = _collect(TotalBcaRunner(total, pipe_a, lm).run_resident(d_ct, aff))
        got = _collect(TotalBcaRunner(total, pipe_b, lm).run_resident(d_ct, aff))
        for k in ref[0]:
            np.testing.assert_array_equal(got[0][k], ref[0][k], err_msg=k)
        assert got[1] == ref[1] and got[2] == ref[2] and got[3] == ref[3]
        assert len(np.unique(ref[0]["total"])) > 50
        d_ct.free()
        total.close()
        pipe_a.close()
        pipe_b.close()
    finally:
        ctx_a.close()
        ctx_b.close()
