"""File-level drop-in: `compute_all_models` with the reference's signature on a synthetic CT NIfTI and
synthetic model folders in the `$nnUNet_results` layout; checks the folder contract and that the files carry exactly
what the array-level pipelines (tested against the oracle elsewhere) produce."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _write_models(root, sp_zyx_total, sp_zyx_bca):
    from boa_hip import label_maps, model_store, plans
    for tid, nc in zip(label_maps.PART_TASK_IDS, (25, 27, 19, 24, 27)):
        pj, dj = plans.synthetic_plans(patch=(32, 32, 32), features=(32, 64), num_classes=nc, spacing=sp_zyx_total)
        geom = plans.model_config_from_plans(pj, dj).geometry
        model_store.write_model_folder(root, tid, f"TotalSegmentator_part{tid - 290}", "nnUNetTrainerNoMirroring", pj, dj,
                                       [plans.synthetic_state_dict(geom, seed=tid)])
    # crop cascade: rough `total` at 6 mm (118 classes) and a native-resolution task model
    for tid, nc, name, trainer, sp in ((298, 118, "TotalSegmentator_6mm", "nnUNetTrainer_4000epochs_NoMirroring", (6.0, 6.0, 6.0)),
                                       (258, 3, "lung_vessels", "nnUNetTrainer", sp_zyx_total)):
        pj, dj = plans.synthetic_plans(patch=(32, 32, 32), features=(32, 64), num_classes=nc, spacing=sp)
        geom = plans.model_config_from_plans(pj, dj).geometry
        model_store.write_model_folder(root, tid, name, trainer, pj, dj, [plans.synthetic_state_dict(geom, seed=tid)])
    for tid, nc, name, trainer in ((543, 7, "BCA_body_parts", "nnUNetTrainer_1500epochs_NoMirroring"),
                                   (542, 12, "BCA_inference", "nnUNetTrainerNoMirroring")):
        pj, dj = plans.synthetic_plans(patch=(32, 32, 32), features=(32, 64), num_classes=nc, spacing=sp_zyx_bca)
        geom = plans.model_config_from_plans(pj, dj).geometry
        model_store.write_model_folder(root, tid, name, trainer, pj, dj,
                                       [plans.synthetic_state_dict(geom, seed=tid + f) for f in range(5)])


def test_compute_all_models_total_bca(tmp_path, monkeypatch):
    from boa_hip import label_maps, model_store, nifti
    from boa_hip.compute.inference import compute_all_models, get_context
    from boa_hip.pipeline import BcaPipelineHip
    from boa_hip.synthetic import ct_phantom
    from boa_hip.task import SegmentationTask
    root = tmp_path / "results"
    _write_models(str(root), (1.5, 1.5, 1.5), (5.0, 1.5, 1.5))
    monkeypatch.setenv("nnUNet_results", str(root))
    ct = ct_phantom((48, 40, 56), seed=5)
    aff = np.diag([-1.5, -1.5, 1.5, 1.0])                       # LPS file
    aff[:3, 3] = [30.0, 40.0, -100.0]
    ct_path = tmp_path / "ct.nii.gz"
    nifti.save(ct_path, ct, aff)
    out = tmp_path / "seg"
    params = {"preview": False, "fast": False, "ml": True, "nr_thr_resamp": 1, "nr_thr_saving": 1, "quiet": True,
              "verbose": False, "device": "gpu", "license_number": None}
    stats = compute_all_models(ct_path, out, ["total", "bca"], params, fast_bca=True,
                               bca_params={"median_filtering": False, "examined_body_region": None, "save_pdf": False,
                                           "theme": "light"})
    assert stats == {"num_voxels": 48 * 40 * 56, "num_slices": 56, "num_slices_resampled": 56}
    for f in ("total.nii.gz", "total-measurements.json", "ct_pfav.nii.gz", "body_parts.nii.gz", "body_regions.nii.gz",
              "tissues.nii.gz", "bca-measurements.json"):
        assert (out / f).is_file(), f
    total, taff, th = nifti.load(out / "total.nii.gz")
    assert total.dtype == np.uint8 and total.shape == ct.shape and np.allclose(taff, aff)
    assert nifti.parse_label_xml(th.extensions[0][1]) == label_maps.CLASS_MAP_TOTAL
    # same numbers as the array-level pipelines
    ctx = get_context("gpu")
    t = SegmentationTask(ctx, "total", model_store.load_task_models("total"), resample=1.5, multimodel=True)
    np.testing.assert_array_equal(total, t.predict_image(ct, aff))
    t.close()
    pm = model_store.load_task_models("body_parts", True)[0]
    rm = model_store.load_task_models("body_regions", True)[0]
    pipe = BcaPipelineHip(ctx, (pm[1], pm[2]), (rm[1], rm[2]), fast_bca=True)
    ref = pipe.run(ct, aff, total_seg=total)
    pipe.close()
    for name in ("body_parts", "body_regions", "tissues"):
        np.testing.assert_array_equal(nifti.load(out / f"{name}.nii.gz")[0], ref[name])
    with open(out / "bca-measurements.json") as f:
        js = json.load(f)
    assert js == json.loads(json.dumps(ref["bca_measurements"], default=float))
    with open(out / "total-measurements.json") as f:
        tm = json.load(f)
    assert set(tm) >= {"segmentations", "info"} and "total" in tm["segmentations"]
    present = [v for v in tm["segmentations"]["total"].values() if v["present"]]
    assert present and set(present[0]) == {"present", "volume_ml", "mean_hu", "std_hu", "min_hu", "median_hu", "max_hu",
                                           "25th_percentile_hu", "75th_percentile_hu", "cnr"}
    # recompute=False keeps existing outputs; unknown cascade models are refused, not skipped
    before = os.path.getmtime(out / "total.nii.gz")
    compute_all_models(ct_path, out, ["total"], params, recompute=False)
    assert os.path.getmtime(out / "total.nii.gz") == before
    with pytest.raises(NotImplementedError):
        compute_all_models(ct_path, out, ["coronary_arteries"], params)
    # crop-cascade model: rough 6 mm `total` -> lung mask -> native-resolution model; same result as the array-level driver
    from boa_hip.task import run_cascade_task
    compute_all_models(ct_path, out, ["lung_vessels"], params)
    lv, _, lh = nifti.load(out / "lung_vessels.nii.gz")
    info = model_store.TASKS["lung_vessels"]
    want = run_cascade_task(ctx, "lung_vessels", ct, aff, model_store.load_task_models("total_6mm"),
                            model_store.load_task_models("lung_vessels"), info["crop"], model_store.effective_crop_addon("lung_vessels"))
    np.testing.assert_array_equal(lv, want)
    assert lv.any() and set(np.unique(lv)) <= {0, 1, 2}
    assert nifti.parse_label_xml(lh.extensions[0][1]) == {1: "lung_vessels", 2: "lung_trachea_bronchia"}
    # the reference measures <ADDITIONAL_MODELS_OUTPUT_NAME>.nii.gz = lung_vessels_airways.nii.gz, which inference never
    # writes: the model is (silently, as in the reference) absent from total-measurements.json
    with open(out / "total-measurements.json") as f:
        assert "lung_vessels" not in json.load(f)["segmentations"]


def test_error_conventions(tmp_path, monkeypatch):
    """SURVEY 8b "Error conventions": the exception types of the reference's path -- ValueError for a non-3-D CT
    (BOA/compute/inference.py:37-38), ValueError for a 2-D array and TypeError for a structured dtype
    (TS/nnunet.py:403-411), ValueError when segmentation and CT spacing differ (BOA/compute/measurements.py:272-278),
    RuntimeError when the normalised logits contain inf (predict_from_raw_data.py:622-625), ValueError for unknown models."""
    from boa_hip import model_store, nifti, plans
    from boa_hip.compute.config import resolve_models
    from boa_hip.compute.inference import compute_all_models, get_context
    from boa_hip.compute.measurements import compute_measurements
    from boa_hip.predictor import HipPredictor
    from boa_hip.task import SegmentationTask
    root = tmp_path / "results"
    _write_models(str(root), (1.5, 1.5, 1.5), (5.0, 1.5, 1.5))
    monkeypatch.setenv("nnUNet_results", str(root))
    params = {"preview": False, "fast": False, "ml": True, "nr_thr_resamp": 1, "nr_thr_saving": 1, "quiet": True,
              "verbose": False, "device": "gpu", "license_number": None}
    aff = np.diag([1.5, 1.5, 1.5, 1.0])
    ct4 = np.zeros((8, 8, 8, 2), dtype=np.int16)
    nifti.save(tmp_path / "ct4d.nii.gz", ct4, aff)
    with pytest.raises(ValueError, match="Only 3D CT scans"):
        compute_all_models(tmp_path / "ct4d.nii.gz", tmp_path / "o1", ["total"], params)
    with pytest.raises(ValueError, match="Unknown model"):
        resolve_models("total+nonsense", strict=True)
    assert resolve_models("total+nonsense") == {"total"}            # non-strict: logged and ignored, as in the reference
    ctx = get_context("gpu")
    t = SegmentationTask(ctx, "total", model_store.load_task_models("total"), resample=1.5, multimodel=True)
    with pytest.raises(ValueError, match="2D images"):
        t.predict_image(np.zeros((16, 16), dtype=np.int16), aff)
    with pytest.raises(TypeError, match="structured"):
        t.predict_image(np.zeros((8, 8, 8), dtype=[("a", np.int16), ("b", np.int16)]), aff)
    t.close()
    # spacing mismatch between CT and segmentation
    ct = np.full((16, 16, 16), 30, dtype=np.int16)
    nifti.save(tmp_path / "ct.nii.gz", ct, aff)
    seg_dir = tmp_path / "seg"
    seg_dir.mkdir()
    nifti.save(seg_dir / "total.nii.gz", np.ones((16, 16, 16), dtype=np.uint8), np.diag([3.0, 3.0, 3.0, 1.0]))
    with pytest.raises(ValueError, match="spacing of the image and of the segmentation"):
        compute_measurements(tmp_path / "ct.nii.gz", seg_dir, ["total"], cnr_adjustment=False, ctx=ctx)
    # inf in the normalised logits: head weights (finite in fp16) that make the Gaussian-weighted fp16 sums overflow
    pj, dj = plans.synthetic_plans(patch=(32, 32, 32), features=(32, 64), num_classes=3, spacing=(1.5, 1.5, 1.5))
    cfg = plans.model_config_from_plans(pj, dj)
    sd = plans.synthetic_state_dict(cfg.geometry, seed=1)
    for k in sd:
        if "seg_layers" in k and k.endswith("weight"):
            sd[k] = sd[k] * 2e4
    p = HipPredictor(ctx, cfg.geometry, tile_step_size=0.5)
    p.set_parameters([plans.weight_blob_from_state_dict(cfg.geometry, sd)])
    x = np.random.default_rng(0).normal(0, 1, size=(1, 40, 36, 33)).astype(np.float32)
    with pytest.raises(RuntimeError, match="inf"):
        p.predict_sliding_window_return_logits(x)
    p.close()


def test_heartchambers_highres_cascade_with_remove_outside(tmp_path, monkeypatch):
    """The licensed cascade task (TS/python_api.py:493-505): robust_crop -> the 3 mm `total` model (Dataset297) makes the crop
    mask (margin 20 mm, python_api.py:726), the task model runs at native resolution on the crop, and the result is cleared
    outside the union of heart / aorta / inferior_vena_cava dilated by int(10 mm / mean spacing) voxels
    (TS/nnunet.py:711-716, TS/postprocessing.py:101-131).  File-level result == the array-level composition with the
    dilation done by scipy (the reference's own call)."""
    from scipy import ndimage
    from boa_hip import label_maps, model_store, nifti, plans
    from boa_hip.compute.inference import compute_all_models, get_context
    from boa_hip.synthetic import ct_phantom
    from boa_hip.task import SegmentationTask
    root = tmp_path / "results"
    for tid, nc, name, trainer, sp in ((297, 118, "TotalSegmentator_3mm", "nnUNetTrainer_4000epochs_NoMirroring", (3.0, 3.0, 3.0)),
                                       (301, 8, "heartchambers_highres", "nnUNetTrainer", (1.5, 1.5, 1.5))):
        pj, dj = plans.synthetic_plans(patch=(32, 32, 32), features=(32, 64), num_classes=nc, spacing=sp)
        geom = plans.model_config_from_plans(pj, dj).geometry
        model_store.write_model_folder(str(root), tid, name, trainer, pj, dj, [plans.synthetic_state_dict(geom, seed=tid)])
    monkeypatch.setenv("nnUNet_results", str(root))
    ct = ct_phantom((64, 56, 60), seed=9)
    aff = np.diag([1.5, 1.5, 1.5, 1.0])
    ct_path = tmp_path / "ct.nii.gz"
    nifti.save(ct_path, ct, aff)
    out = tmp_path / "seg"
    params = {"preview": False, "fast": False, "ml": True, "nr_thr_resamp": 1, "nr_thr_saving": 1, "quiet": True, "verbose": False,
              "device": "gpu", "license_number": "aca_XXXXXXXXXXXXXXX"}
    compute_all_models(ct_path, out, ["heartchambers_highres"], params)
    got = nifti.load(out / "heartchambers_highres.nii.gz")[0]
    ctx = get_context("gpu")
    rough = SegmentationTask(ctx, "total", model_store.load_task_models("total_fast"), resample=3.0, multimodel=False)
    organ = rough.predict_image(ct, aff)
    rough.close()
    inv = label_maps.CLASS_MAP_TOTAL_INV
    crop_mask = (organ == inv["heart"]).astype(np.uint8)
    t = SegmentationTask(ctx, "heartchambers_highres", model_store.load_task_models("heartchambers_highres"), resample=None, multimodel=False)
    seg = t.predict_image(ct, aff, crop_mask=crop_mask, crop_addon=[20, 20, 20])
    t.close()
    rm = np.isin(organ, [inv[n] for n in ("heart", "aorta", "inferior_vena_cava")])
    vx = int(10 / np.mean(np.array([1.5, 1.5, 1.5], dtype=np.float32)))
    want = seg.copy()
    want[ndimage.binary_dilation(rm, iterations=vx) == 0] = 0
    np.testing.assert_array_equal(got, want)
    # (random-weight nets: whether the rough model emits a `heart` label at all is luck; the dilation itself is checked against
    #  scipy in test_remove_outside_of_mask_vs_scipy)
    print("heartchambers: crop voxels", int(crop_mask.sum()), "labels kept", int((want > 0).sum()), "removed", int((seg != want).sum()))


def test_remove_outside_of_mask_vs_scipy():
    from scipy import ndimage
    from boa_hip.compute.inference import get_context
    from boa_hip.task import remove_outside_of_mask
    rng = np.random.default_rng(2)
    seg = rng.integers(0, 9, size=(23, 31, 19)).astype(np.uint8)
    mask = np.zeros(seg.shape, np.uint8)
    mask[5:9, 10:14, 3:6] = 1
    mask[20, 29, 17] = 3
    mask[0, 0, 0] = 1
    for addon in (1, 2, 5):
        want = seg.copy()
        want[ndimage.binary_dilation(mask, iterations=addon) == 0] = 0
        np.testing.assert_array_equal(remove_outside_of_mask(get_context("gpu"), seg, mask, addon), want)


def test_compute_all_models_with_legacy_plans_and_upstream_checkpoint_keys(tmp_path, monkeypatch):
    """model folders as the real weight archives ship them -- plans.json in the OLD nnU-Net format
    (`UNet_class_name`, `conv_kernel_sizes`, `pool_op_kernel_sizes`, ...: plans_handler.py:36-97), dataset.json with `modality`,
    checkpoints that carry upstream's alias keys (`...all_modules.N.*`, `decoder.encoder.*`), the deep-supervision heads and a
    torch.compile prefix -- through `compute_all_models(["total"])`: the label file must equal the one computed from plain
    folders holding the same weights."""
    from test_host_cpu import _upstream_checkpoint_keys
    from boa_hip import label_maps, model_store, nifti, plans
    from boa_hip.compute.inference import compute_all_models
    from boa_hip.synthetic import ct_phantom
    roots = {"plain": tmp_path / "plain", "upstream": tmp_path / "upstream"}
    for tid, nc in zip(label_maps.PART_TASK_IDS, (25, 27, 19, 24, 27)):
        pj, dj = plans.synthetic_plans(patch=(32, 32, 32), features=(32, 64, 128), num_classes=nc)
        geom = plans.model_config_from_plans(pj, dj).geometry
        sd = plans.synthetic_state_dict(geom, seed=tid)
        model_store.write_model_folder(str(roots["plain"]), tid, f"TotalSegmentator_part{tid - 290}", "nnUNetTrainerNoMirroring", pj, dj, [sd])
        lj, ldj = plans.legacy_plans_from(pj, dj)
        model_store.write_model_folder(str(roots["upstream"]), tid, f"TotalSegmentator_part{tid - 290}", "nnUNetTrainerNoMirroring", lj, ldj,
                                       [_upstream_checkpoint_keys(sd, geom.n_stages)])
    ct = ct_phantom((44, 40, 52), seed=8)
    aff = np.diag([1.5, 1.5, 1.5, 1.0])
    nifti.save(tmp_path / "ct.nii.gz", ct, aff)
    params = {"preview": False, "fast": False, "ml": True, "nr_thr_resamp": 1, "nr_thr_saving": 1, "quiet": True,
              "verbose": False, "device": "gpu", "license_number": None}
    labels = {}
    for kind, root in roots.items():
        monkeypatch.setenv("nnUNet_results", str(root))
        compute_all_models(tmp_path / "ct.nii.gz", tmp_path / kind, ["total"], params)
        labels[kind] = nifti.load(tmp_path / kind / "total.nii.gz")[0]
    assert len(np.unique(labels["plain"])) > 5
    np.testing.assert_array_equal(labels["upstream"], labels["plain"])
    # a checkpoint with a key the architecture does not have is refused, not silently truncated
    import torch
    pth = next((roots["upstream"]).glob("Dataset291_*/*/fold_0/checkpoint_final.pth"))
    ck = torch.load(pth, map_location="cpu", weights_only=False)
    ck["network_weights"]["_orig_mod.encoder.stages.0.0.convs.0.norm.running_mean"] = torch.zeros(32)
    torch.save(ck, pth)
    with pytest.raises(ValueError, match="unexpected checkpoint key"):
        compute_all_models(tmp_path / "ct.nii.gz", tmp_path / "bad", ["total"], params)


def test_dicom_folder_input_gives_the_labels_of_the_nifti_input(tmp_path, monkeypatch):
    """`analyze_ct` (BOA/commands.py:121-129) hands a DICOM folder to `get_image_info` and the resulting image.nii.gz to
    `compute_all_models`: the folder path of the drop-in.  A CT phantom written as an axial series (explicit VR, stored = HU + 1024,
    shuffled file order) must come out as the SAME label volume as the phantom saved directly as NIfTI with the series' geometry
    (DICOM reader unpinned vs GDCM / SimpleITK: tests/dicom_writer.py is the only writer it has met)."""
    from boa_hip import nifti
    from boa_hip.compute.inference import compute_all_models
    from boa_hip.compute.io import get_image_info
    from boa_hip.synthetic import ct_phantom
    from dicom_writer import write_series
    root = tmp_path / "results"
    _write_models(str(root), (1.5, 1.5, 1.5), (5.0, 1.5, 1.5))
    monkeypatch.setenv("nnUNet_results", str(root))
    ct = ct_phantom((48, 40, 36), seed=9)                        # file order (x, y, z), int16 HU in [-1024, 3071]
    stored = (ct.astype(np.int32) + 1024).astype(np.uint16).transpose(2, 1, 0)      # [z][rows = y][cols = x]
    rng = np.random.default_rng(0)
    order = list(rng.permutation(stored.shape[0]))
    write_series(tmp_path / "dcm", stored, origin=(-30.0, -40.0, 100.0), spacing=(1.5, 1.5), dz=1.5, order=order, bits_stored=12)
    ct_path, ct_info = get_image_info(tmp_path / "dcm", tmp_path / "proc")
    data, aff, _ = nifti.load(ct_path)
    np.testing.assert_array_equal(data, ct)
    want_aff = np.diag([-1.5, -1.5, 1.5, 1.0])
    want_aff[:3, 3] = [30.0, 40.0, 100.0]
    np.testing.assert_allclose(aff, want_aff, atol=1e-5)
    assert {e["name"]: e["value"] for e in ct_info}["Modality"] == "CT"
    params = {"preview": False, "fast": False, "ml": True, "nr_thr_resamp": 1, "nr_thr_saving": 1, "quiet": True,
              "verbose": False, "device": "gpu", "license_number": None}
    compute_all_models(ct_path, tmp_path / "from_dicom", ["total"], params)
    ref_path = tmp_path / "ct.nii.gz"
    nifti.save(ref_path, ct, want_aff)
    compute_all_models(ref_path, tmp_path / "from_nifti", ["total"], params)
    a, _, _ = nifti.load(tmp_path / "from_dicom" / "total.nii.gz")
    b, _, _ = nifti.load(tmp_path / "from_nifti" / "total.nii.gz")
    assert a.shape == ct.shape and (a == b).all() and a.max() > 0
