"""BASELINE.json `configs` as -m gpu tests (synthetic weights of the documented geometries):
  configs[0]  128^3 CT, `--models total --fast-total` (Dataset297 at 3 mm) through the file-level drop-in surface, against
              the oracle pipeline (torch-CPU fp32 net) -- includes the pad_nd_image path (resampled volume thinner than the patch)
  configs[2]  512x512x768 `total+bca` on one GPU: runs, is reproducible bit for bit, tables are consistent
  configs[4]  512x512x1024 `total`: the reference's triple z-split at full size -- the middle third of the result equals
              the middle part predicted on its own
  configs[3]  (8 volumes on 8 GPUs) is the bench's volume-sharded mode: covered by tests/test_distributed_cpu.py and the
              RCCL tests of test_gpu_tile_shard.py, which skip on a one-GPU box.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from boa_hip.device import Context
    c = Context(0)
    yield c
    c.close()


def test_config0_total_fast_128_dropin_vs_oracle(tmp_path, monkeypatch):
    """`compute_all_models(ct, folder, "total", {"fast": True})`: one model (Dataset297), resample 3.0, step 0.5
    (TS/nnunet.py:507-514: only `total` below 3 mm uses 0.8), single-model task (no part merge, labels = argmax)."""
    import torch
    from boa_hip import label_maps, model_store, nifti, plans
    from boa_hip.compute.inference import compute_all_models
    from boa_hip.synthetic import ct_phantom
    from oracle import pipeline as opipe
    from oracle.network import build_from_arch, network_fn_from_module
    root = tmp_path / "results"
    pj, dj = plans.synthetic_plans(patch=(64, 96, 64), features=(32, 64, 128), num_classes=118, spacing=(3.0, 3.0, 3.0))
    cfg = plans.model_config_from_plans(pj, dj)
    sd = plans.synthetic_state_dict(cfg.geometry, seed=297)
    model_store.write_model_folder(str(root), 297, "TotalSegmentator_3mm", "nnUNetTrainer_4000epochs_NoMirroring", pj, dj, [sd])
    monkeypatch.setenv("nnUNet_results", str(root))
    ct = ct_phantom((128, 128, 128), seed=1)
    aff = np.diag([1.5, 1.5, 1.5, 1.0])
    ct_path = tmp_path / "ct.nii.gz"
    nifti.save(ct_path, ct, aff)
    out = tmp_path / "seg"
    params = {"preview": False, "fast": True, "ml": True, "nr_thr_resamp": 1, "nr_thr_saving": 1, "quiet": True, "verbose": False,
              "device": "gpu", "license_number": None}
    stats = compute_all_models(ct_path, out, "total", params)
    assert stats == {"num_voxels": 128 ** 3, "num_slices": 128, "num_slices_resampled": 128}
    got, gaff, hdr = nifti.load(out / "total.nii.gz")
    assert got.shape == ct.shape and got.dtype == np.uint8 and np.allclose(gaff, aff)
    assert nifti.parse_label_xml(hdr.extensions[0][1]) == label_maps.CLASS_MAP_TOTAL
    assert (out / "total-measurements.json").is_file()
    net = build_from_arch(pj["configurations"]["3d_fullres"]["architecture"]["arch_kwargs"], 1, 118)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    omodel = (network_fn_from_module(net, 8), (64, 96, 64), 118, cfg.intensity_properties["0"], None)
    want = opipe.predict_image(ct, (1.5, 1.5, 1.5), [omodel], None, "total", 3.0, multimodel=False)
    agree = float((got == want).mean())
    print("configs[0] total_fast 128^3 label agreement with the oracle pipeline", agree, "labels", len(np.unique(got)))
    assert agree >= 0.993   # measured 0.9964;           # 118 classes of a random-weight fp16 net; every other step is exact
    # the same call in the fp32 modes: the label file of the CPU path up to fp32 summation-order near-ties.  The network runs at
    # 3 mm (64^3) and its labels are upsampled by exactly 2 per axis (nearest), so one flipped network voxel is 8 file voxels.
    #   fp32_ref (plain fp32 MFMAs, the cross-check mode): the contract bar, flips <= 1e-5 of the FILE's voxels (20 file voxels =
    #       2 network voxels at this size).
    #   fp32 ($BOA_NET_PRECISION=fp32 = the split-precision mode on the f16 matrix cores, INTEGRATION.md "precision modes"): the
    #       same arithmetic contract evaluated with another summation grouping; measured 2-4 flipped network voxels of 262 144 over
    #       boxes and builds (118 classes make top-2 near-ties ~5x as likely as the 25 classes of
    #       tests/test_gpu_production_geometry.py, whose 2e-5 bar this is): 2e-5 of the NETWORK's voxels = 5.
    bars = {"fp32_ref": max(8, 1e-5 * want.size), "fp32": max(8, 2e-5 * (want.size / 8) * 8)}
    for prec in ("fp32", "fp32_ref"):
        monkeypatch.setenv("BOA_NET_PRECISION", prec)
        out32 = tmp_path / f"seg_{prec}"
        compute_all_models(ct_path, out32, "total", params)
        got32, _, _ = nifti.load(out32 / "total.nii.gz")
        flips = int((got32 != want).sum())
        print(f"configs[0] {prec} mode: label flips", flips, "of", want.size, "=", flips / 8, "network voxels of", want.size // 8,
              "bar", bars[prec])
        assert flips <= bars[prec]


def _bca_models(folds):
    from boa_hip import plans
    out = {}
    for name, nc, seed in (("body_parts", 7, 543), ("body_regions", 12, 542)):
        pj, dj = plans.synthetic_plans(num_classes=nc, spacing=(5.0, 1.5, 1.5))
        cfg = plans.model_config_from_plans(pj, dj)
        out[name] = (cfg, [plans.weight_blob_from_state_dict(cfg.geometry, plans.synthetic_state_dict(cfg.geometry, seed + f))
                           for f in range(folds)])
    return out


def test_config2_total_bca_512x512x768(ctx):
    """configs[2]: whole-body 512x512x768 @1.5 mm, `total` (1 000 tile forwards) + `bca` with the FIVE folds per net the config
    names (2 x 5 x 147 tiles at 5 mm slices, step 0.5) + total measurements on one GPU.  Properties: two runs agree
    bit for bit (labels and tables), every table is consistent with the label volumes it summarises."""
    from boa_hip import label_maps, synthetic
    from boa_hip import measurements as M
    from boa_hip.devarray import DevArray
    from boa_hip.pipeline import BcaPipelineHip
    from boa_hip.totalseg import TotalSegmentatorHip
    shape = (512, 512, 768)
    ct = synthetic.ct_phantom(shape, seed=3)
    aff = np.diag([-1.5, -1.5, 1.5, 1.0])
    ts = TotalSegmentatorHip(ctx, [(tid, cfg, [blob]) for tid, cfg, blob, _ in synthetic.total_part_models()])
    bm = _bca_models(5)
    pipe = BcaPipelineHip(ctx, bm["body_parts"], bm["body_regions"], fast_bca=False)
    runs = []
    for _ in range(2):
        total = ts.predict(ct, affine=aff)
        out = pipe.run(ct, aff, total_seg=total)
        runs.append((total, out))
    ts.close()
    pipe.close()
    (t0, o0), (t1, o1) = runs
    assert t0.shape == shape and t0.dtype == np.uint8 and len(np.unique(t0)) > 20
    np.testing.assert_array_equal(t0, t1)
    for k in ("body_parts", "body_regions", "tissues"):
        np.testing.assert_array_equal(o0[k], o1[k])
    assert o0["bca_measurements"] == o1["bca_measurements"]
    # tables vs volumes: per-slice tissue volumes sum to the voxel counts of the tissue map
    js = o0["bca_measurements"]
    ml = 1.5 ** 3 / 1000.0
    tis = o0["tissues"]
    slices = js["slices"]
    assert len(slices) == shape[2]
    for name, val in (("muscle", 1), ("bone", 2)):
        tot = sum(s[name] for s in slices) if name in slices[0] else None
        if tot is not None:
            assert abs(tot - float((tis == val).sum()) * ml) <= 1e-6 * max(tot, 1.0)
    # per-label HU statistics: the voxel counts of `total-measurements` equal numpy's
    lm = label_maps.measurement_label_map("total")
    d_f = DevArray.from_numpy(ctx, ct)
    d_ct = d_f.transpose((2, 1, 0)).contiguous(np.int16, force_copy=True)
    d_s = DevArray.from_numpy(ctx, t0)
    d_lab = d_s.transpose((2, 1, 0)).contiguous(force_copy=True)
    meas, _ = M.total_measurements(ctx, None, None, lm, (1.5, 1.5, 1.5), d_ct=d_ct.buf, d_lab=d_lab.buf, shape=d_ct.shape)
    for a in (d_f, d_ct, d_s, d_lab):
        a.free()
    counts = np.bincount(t0.ravel(), minlength=256)
    seg = meas["segmentations"]["total"]
    for name, lab in list(lm.items())[:40]:
        if counts[lab]:
            assert seg[name]["present"] and abs(seg[name]["volume_ml"] - counts[lab] * ml) <= 1e-9 * counts[lab] * ml
        else:
            assert not seg[name]["present"]


def test_config4_triple_split_512x512x1024(ctx):
    """configs[4] shape: 268 M voxels > 512*512*900 and z > 200 trigger the reference's triple z-split for the multi-model
    `total` task (TS/nnunet.py:489-505, recombination :583-586): 3 x 100 tiles per model = 1 500 tile forwards.  The
    middle third of the result must equal the middle part predicted on its own (split bookkeeping at full size)."""
    from boa_hip import synthetic
    from boa_hip.task import split_bounds
    from boa_hip.totalseg import TotalSegmentatorHip
    shape = (512, 512, 1024)
    ct = synthetic.ct_phantom(shape, seed=4)
    ct[:, :, 0] = 7                       # nothing to crop: the border planes are non-zero
    aff = np.diag([1.5, 1.5, 1.5, 1.0])
    ts = TotalSegmentatorHip(ctx, [(tid, cfg, [blob]) for tid, cfg, blob, _ in synthetic.total_part_models()])
    seg = ts.predict(ct, affine=aff)
    assert seg.shape == shape and len(np.unique(seg)) > 20
    parts, comb = split_bounds(shape[2])
    assert parts == [(0, 361), (322, 702), (663, 1024)]
    lo, hi = parts[1]
    mid = ts.predict(np.ascontiguousarray(ct[:, :, lo:hi]), affine=aff)
    ts.close()
    dst, src = comb[1]
    np.testing.assert_array_equal(seg[:, :, dst], mid[:, :, src])



def test_config4_models_all_512x512x1024(tmp_path, monkeypatch):
    """configs[4]: `--models all` (BOA/compute/constants: total + bca + lung_vessels + cerebral_bleed + hip_implant +
    pleural_pericard_effusion + liver_vessels) on one 512x512x1024 @1.5 mm CT through the file-level drop-in
    (`compute_all_models`), synthetic model folders with the documented 128^3 six-stage geometry for `total` / BCA and small nets for
    the rough 6 mm model and the cascade tasks.  Property test (no CPU reference at this size): every output of the folder contract
    exists with the input's grid, labels stay inside the task's class map, the stats dict is the reference's, the triple z-split is
    taken (268 M voxels), tables are consistent with the volumes, and a cascade result is zero outside its crop box."""
    import json
    from boa_hip import label_maps, model_store, nifti, plans, synthetic
    from boa_hip.compute.constants import ALL_MODELS
    from boa_hip.compute.inference import compute_all_models
    root = str(tmp_path / "results")
    for (tid, cfg, blob, (pj, dj, sd)), k in zip(synthetic.total_part_models(), range(5)):
        model_store.write_model_folder(root, tid, f"TotalSegmentator_part{k + 1}", "nnUNetTrainerNoMirroring", pj, dj, [sd])
    small = dict(patch=(64, 64, 64), features=(32, 64, 128))
    specs = [(298, 118, "TotalSegmentator_6mm", "nnUNetTrainer_4000epochs_NoMirroring", (6.0, 6.0, 6.0), 1)]
    for name in ("lung_vessels", "cerebral_bleed", "hip_implant", "pleural_pericard_effusion", "liver_vessels"):
        info = model_store.TASKS[name]
        specs.append((info["task_id"][0], max(label_maps.class_map(name)) + 1, name, info["trainer"], (1.5, 1.5, 1.5),
                      2 if info["folds"] is None else 1))
    for tid, nc, name, trainer, sp, nfolds in specs:
        pj, dj = plans.synthetic_plans(num_classes=nc, spacing=sp, **small)
        geom = plans.model_config_from_plans(pj, dj).geometry
        model_store.write_model_folder(root, tid, name, trainer, pj, dj, [plans.synthetic_state_dict(geom, seed=tid + f) for f in range(nfolds)])
    for tid, nc, name, trainer in ((543, 7, "BCA_body_parts", "nnUNetTrainer_1500epochs_NoMirroring"), (542, 12, "BCA_inference", "nnUNetTrainerNoMirroring")):
        pj, dj = plans.synthetic_plans(num_classes=nc, spacing=(5.0, 1.5, 1.5))
        geom = plans.model_config_from_plans(pj, dj).geometry
        model_store.write_model_folder(root, tid, name, trainer, pj, dj, [plans.synthetic_state_dict(geom, seed=tid + f) for f in range(5)])
    monkeypatch.setenv("nnUNet_results", root)
    shape = (512, 512, 1024)
    ct = synthetic.ct_phantom(shape, seed=44)
    aff = np.diag([-1.5, -1.5, 1.5, 1.0])
    ct_path = tmp_path / "ct.nii.gz"
    nifti.save(ct_path, ct, aff, threads=8)
    out = tmp_path / "seg"
    params = {"preview": False, "fast": False, "ml": True, "nr_thr_resamp": 1, "nr_thr_saving": 8, "quiet": True, "verbose": False,
              "device": "gpu", "license_number": None}
    models = sorted(ALL_MODELS - {"body_parts", "body_regions"})          # `bca` runs both BCA nets
    stats = compute_all_models(ct_path, out, models, params, fast_bca=False,
                               bca_params={"median_filtering": False, "examined_body_region": None, "save_pdf": False, "theme": "light"})
    assert stats == {"num_voxels": 512 * 512 * 1024, "num_slices": 1024, "num_slices_resampled": 1024}
    nvox = int(np.prod(shape))
    for name in ("total", "lung_vessels", "cerebral_bleed", "hip_implant", "pleural_pericard_effusion", "liver_vessels", "body_parts",
                 "body_regions", "tissues"):
        seg, saff, hdr = nifti.load(out / f"{name}.nii.gz")
        assert seg.shape == shape and seg.dtype == np.uint8, name
        np.testing.assert_allclose(saff, aff)
        labs = np.flatnonzero(np.bincount(seg.ravel(), minlength=256))
        if name in model_store.CASCADE_MODELS:      # labels are the network's class indices: within the task's class map range
            assert set(labs.tolist()) <= set(range(max(label_maps.class_map(name)) + 1)), (name, labs)
        if name == "total":
            assert len(labs) > 40 and hdr.extensions and nifti.parse_label_xml(hdr.extensions[0][1]) == label_maps.CLASS_MAP_TOTAL
            counts = np.bincount(seg.ravel(), minlength=256)
        del seg
    with open(out / "total-measurements.json") as f:
        tm = json.load(f)
    ml = 1.5 ** 3 / 1000.0
    lm = label_maps.measurement_label_map("total")
    for nm, lab in list(lm.items())[:30]:
        e = tm["segmentations"]["total"][nm]
        assert e["present"] == bool(counts[lab])
        if counts[lab]:
            assert abs(e["volume_ml"] - counts[lab] * ml) <= 1e-9 * counts[lab] * ml
    with open(out / "bca-measurements.json") as f:
        bj = json.load(f)
    assert len(bj["slices"]) == 1024 and "aggregated" in bj
    assert (out / "ct_pfav.nii.gz").is_file()
    assert nvox > 512 * 512 * 900            # the size that triggers the triple z-split of `total` (TS/nnunet.py:489-505)
