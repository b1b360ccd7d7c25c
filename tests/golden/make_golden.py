#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by EXECUTING THE REFERENCE's own functions.

Runs only in the dev container (needs /root/reference); the outputs (*.npz, *.json) are committed and
are the only thing the tests read.  Each vector stores the inputs and the reference's outputs.

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import gzip
import hashlib
import json
import os
import struct
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_harness as H  # noqa: E402

H.install()
import torch  # noqa: E402

torch.set_num_threads(1)


def save_npz(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrs.items()})


def save_json(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
    print("wrote", name)


def load_example_ct():
    """Minimal NIfTI-1 parse of the reference's own test fixture (NN/tests/example_data)."""
    p = os.path.join(H.EXT, "nnunetv2", "tests", "example_data", "example_ct_sm.nii.gz")
    raw = gzip.open(p, "rb").read()
    dim = struct.unpack("<8h", raw[40:56])
    datatype, bitpix = struct.unpack("<hh", raw[70:74])
    vox_offset = int(struct.unpack("<f", raw[108:112])[0])
    slope, inter = struct.unpack("<ff", raw[112:120])
    assert datatype == 4 and bitpix == 16, (datatype, bitpix)
    n = dim[1] * dim[2] * dim[3]
    a = np.frombuffer(raw, dtype="<i2", count=n, offset=vox_offset).reshape(dim[3], dim[2], dim[1])
    if slope not in (0.0, 1.0) or inter != 0.0:
        a = (a * slope + inter)
    return np.ascontiguousarray(a.transpose(2, 1, 0)).astype(np.int16)  # (x, y, z)


# ---------------------------------------------------------------------------------------------- G1
def g1_steps():
    from nnunetv2.inference.sliding_window_prediction import compute_steps_for_sliding_window as f
    cases = []
    for size, patch, step in [
        ((512, 512, 512), (128, 128, 128), 0.8), ((512, 512, 512), (128, 128, 128), 0.5),
        ((512, 512, 768), (128, 128, 128), 0.8), ((512, 512, 1024), (128, 128, 128), 0.8),
        ((512, 512, 361), (128, 128, 128), 0.8), ((512, 512, 380), (128, 128, 128), 0.8),
        ((342, 342, 342), (128, 128, 128), 0.5), ((342, 342, 342), (128, 128, 128), 0.8),
        ((128, 128, 128), (128, 128, 128), 0.5), ((129, 128, 200), (128, 128, 128), 0.5),
        ((154, 512, 512), (64, 192, 160), 0.5), ((230, 512, 512), (64, 192, 160), 0.5),
        ((110, 64, 64), (64, 64, 64), 0.5), ((40, 36, 33), (16, 16, 16), 0.5), ((40, 36, 33), (16, 16, 16), 0.8),
        ((257, 131, 99), (112, 128, 96), 0.5), ((1000, 999, 17), (17, 17, 17), 1.0), ((33, 33, 33), (32, 32, 32), 0.1),
    ]:
        cases.append({"size": size, "patch": patch, "step": step, "steps": f(size, patch, step)})
    save_json("g1_steps.json", cases)


# ---------------------------------------------------------------------------------------------- G2
def g2_gaussian():
    from nnunetv2.inference.sliding_window_prediction import compute_gaussian
    out = {}
    for ts in [(16, 16, 16), (32, 32, 32), (64, 48, 40), (24, 20, 28)]:
        g = compute_gaussian(ts, sigma_scale=1. / 8, value_scaling_factor=10, device=torch.device("cpu"))
        assert g.dtype == torch.float16
        out["g_" + "x".join(map(str, ts))] = g.numpy().view(np.uint16)
    save_npz("g2_gaussian.npz", **out)
    meta = {}
    for ts in [(128, 128, 128), (64, 192, 160)]:
        g = compute_gaussian(ts, sigma_scale=1. / 8, value_scaling_factor=10, device=torch.device("cpu")).numpy()
        bits = np.ascontiguousarray(g.view(np.uint16))
        sub = int(((bits & 0x7C00) == 0).sum())
        meta["x".join(map(str, ts))] = {
            "sha256": hashlib.sha256(bits.tobytes()).hexdigest(), "max": float(g.max()), "min": float(g.min()),
            "n_subnormal": sub, "center_row_bits": bits[ts[0] // 2, ts[1] // 2, :].tolist()}
    save_json("g2_gaussian_big.json", meta)


# ---------------------------------------------------------------------------------------------- G3
class _O:
    pass


def run_ref_predictor(net, x, patch, heads, step, use_gaussian=True, record=None):
    from nnunetv2.inference.predict_from_raw_data import nnUNetPredictor
    p = nnUNetPredictor(tile_step_size=step, use_gaussian=use_gaussian, use_mirroring=False,
                        perform_everything_on_device=False, device=torch.device("cpu"), verbose=False,
                        allow_tqdm=False)
    cm = _O(); cm.patch_size = list(patch); p.configuration_manager = cm
    lm = _O(); lm.num_segmentation_heads = heads; p.label_manager = lm
    p.allowed_mirroring_axes = None

    class Rec(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, t):
            o = self.inner(t)
            if record is not None:
                record.append(o[0].detach().clone().numpy())
            return o

    p.network = Rec(net)
    return p.predict_sliding_window_return_logits(torch.from_numpy(x)).numpy()


def g3_sliding_window():
    rng = np.random.default_rng(20260928)
    out = {}
    cases = [
        ("a", (40, 36, 33), (16, 16, 16), 0.5, 3), ("b", (12, 40, 20), (16, 16, 16), 0.5, 3),
        ("c", (40, 36, 33), (16, 16, 16), 0.8, 4), ("d", (24, 16, 31), (16, 16, 16), 0.5, 2),
    ]
    for name, shape, patch, step, heads in cases:
        torch.manual_seed(zlib.crc32(name.encode()) % 1000)  # deterministic (hash() is salted per process)
        net = torch.nn.Conv3d(1, heads, 3, padding=1)
        with torch.no_grad():
            net.weight.mul_(3.0)
        x = rng.standard_normal((1, *shape)).astype(np.float32) * 2
        rec = []
        logits = run_ref_predictor(net, x, patch, heads, step, True, rec)
        out[f"{name}_x"] = x
        out[f"{name}_w"] = net.weight.detach().numpy()
        out[f"{name}_b"] = net.bias.detach().numpy()
        out[f"{name}_patch"] = np.array(patch)
        out[f"{name}_step"] = np.array(step)
        out[f"{name}_tiles"] = np.stack(rec).astype(np.float32)
        out[f"{name}_logits_bits"] = logits.view(np.uint16)
        from nnunetv2.utilities.label_handling.label_handling import LabelManager
        lmgr = LabelManager({"background": 0, **{f"c{i}": i for i in range(1, heads)}}, None)
        out[f"{name}_seg"] = lmgr.convert_logits_to_segmentation(logits).astype(np.uint8)
    # no-gaussian variant
    torch.manual_seed(5)
    net = torch.nn.Conv3d(1, 2, 3, padding=1)
    x = rng.standard_normal((1, 20, 20, 20)).astype(np.float32)
    rec = []
    logits = run_ref_predictor(net, x, (16, 16, 16), 2, 0.5, False, rec)
    out["e_x"] = x; out["e_tiles"] = np.stack(rec); out["e_logits_bits"] = logits.view(np.uint16)
    out["e_patch"] = np.array((16, 16, 16)); out["e_step"] = np.array(0.5)
    save_npz("g3_sliding_window.npz", **out)


def g3b_fold_ensemble():
    """predict_logits_from_preprocessed_data fold mean (predict_from_raw_data.py:483-500) with 5 'folds'."""
    from nnunetv2.inference.predict_from_raw_data import nnUNetPredictor
    rng = np.random.default_rng(7)
    p = nnUNetPredictor(tile_step_size=0.5, use_gaussian=True, use_mirroring=False,
                        perform_everything_on_device=False, device=torch.device("cpu"), verbose=False,
                        allow_tqdm=False)
    cm = _O(); cm.patch_size = [16, 16, 16]; p.configuration_manager = cm
    lm = _O(); lm.num_segmentation_heads = 3; p.label_manager = lm
    p.allowed_mirroring_axes = None
    p.network = torch.nn.Conv3d(1, 3, 3, padding=1)
    params = []
    for k in range(5):
        torch.manual_seed(100 + k)
        params.append({kk: v.clone() for kk, v in torch.nn.Conv3d(1, 3, 3, padding=1).state_dict().items()})
    p.list_of_parameters = params
    x = rng.standard_normal((1, 24, 20, 18)).astype(np.float32) * 3
    per_fold = []
    for prm in params:
        p.network.load_state_dict(prm)
        per_fold.append(p.predict_sliding_window_return_logits(torch.from_numpy(x)).numpy().view(np.uint16))
    ens = p.predict_logits_from_preprocessed_data(torch.from_numpy(x)).numpy()
    save_npz("g3b_folds.npz", x=x, fold_logits_bits=np.stack(per_fold), ensemble_bits=ens.view(np.uint16),
             w=np.stack([prm["weight"].numpy() for prm in params]), b=np.stack([prm["bias"].numpy() for prm in params]))


# ---------------------------------------------------------------------------------------------- G4
def g4_ctnorm(ct):
    from nnunetv2.preprocessing.normalization.default_normalization_schemes import CTNormalization
    props = {"mean": -370.00039, "std": 436.5998, "percentile_00_5": -1004.0, "percentile_99_5": 1588.0}
    crop = ct[30:94, 20:84, 2:26].astype(np.float32)
    n = CTNormalization(use_mask_for_norm=False, intensityproperties=props)
    o = n.run(crop.copy())
    save_npz("g4_ctnorm.npz", x=crop.astype(np.int16), y=o, props=np.array(
        [props["mean"], props["std"], props["percentile_00_5"], props["percentile_99_5"]], dtype=np.float64))


# ---------------------------------------------------------------------------------------------- G5
def g5_resample(ct):
    from totalsegmentator.resampling import resample_img
    crop = ct[30:70, 20:60, 0:30].astype(np.float64)  # (x,y,z) 40x40x30
    rng = np.random.default_rng(3)
    lab = (rng.integers(0, 6, size=(10, 10, 8)).repeat(4, 0).repeat(4, 1).repeat(4, 2)[:40, :40, :30]).astype(np.float64)
    out = {"ct": crop.astype(np.int16), "lab": lab.astype(np.uint8)}
    zooms = {"half": (0.5, 0.5, 0.5), "twothirds": (2 / 3, 2 / 3, 2 / 3), "thick": (1.0, 1.0, 3.0 / 5.0),
             "up2": (2.0, 2.0, 2.0), "aniso": (1.3, 0.7, 1.9)}
    for k, z in zooms.items():
        out[f"zoom_{k}"] = np.array(z)
        out[f"ct3_{k}"] = resample_img(crop, zoom=np.array(z), order=3, nr_cpus=1).astype(np.int32)
        out[f"lab0_{k}"] = resample_img(lab, zoom=np.array(z), order=0, nr_cpus=1).astype(np.uint8)
    save_npz("g5_resample.npz", **out)


# ---------------------------------------------------------------------------------------------- G6/G7
def g67_argmax_merge():
    from nnunetv2.utilities.label_handling.label_handling import LabelManager
    from totalsegmentator.map_to_binary import class_map, class_map_5_parts, map_taskid_to_partname_ct
    rng = np.random.default_rng(11)
    C = 7
    lg = rng.standard_normal((C, 9, 10, 11)).astype(np.float16)
    lg[:, 0, 0, :] = np.float16(1.5)            # all tie -> 0
    lg[3, 1, :, :] = lg[5, 1, :, :] = np.float16(9)  # tie 3 vs 5 -> 3
    lg[:, 2, 0, 0] = np.float16(0.0); lg[2, 2, 0, 0] = np.float16(-0.0)
    lg[4, 3, 3, 3] = np.float16(np.nan)
    lg[6, 4, :, :] = np.float16(np.inf)
    lm = LabelManager({"background": 0, **{f"c{i}": i for i in range(1, C)}}, None)
    seg = lm.convert_logits_to_segmentation(lg)
    save_npz("g6_argmax.npz", logits_bits=lg.view(np.uint16), seg=seg.astype(np.uint8))

    # G7: label tables + merge order (TS/nnunet.py:536-556, transcribed loop executed on reference tables)
    class_map_inv = {v: k for k, v in class_map["total"].items()}
    tids = [291, 292, 293, 294, 295]
    shape = (12, 13, 14)
    segs = [rng.integers(0, len(class_map_5_parts[map_taskid_to_partname_ct[t]]) + 1, size=shape).astype(np.uint8)
            for t in tids]
    comb = np.zeros(shape, dtype=np.uint8)
    for t, seg_ in zip(tids, segs):
        segf = seg_.astype(np.float64)
        for jdx, cname in class_map_5_parts[map_taskid_to_partname_ct[t]].items():
            comb[segf == jdx] = class_map_inv[cname]
    save_npz("g7_merge.npz", segs=np.stack(segs), combined=comb)
    save_json("g7_label_tables.json", {
        "total": {str(k): v for k, v in class_map["total"].items()},
        "parts": {str(t): {str(k): v for k, v in class_map_5_parts[map_taskid_to_partname_ct[t]].items()} for t in tids},
        "body_regions": {str(k): v for k, v in class_map.get("body_regions", {}).items()} if "body_regions" in class_map else {},
    })


# ---------------------------------------------------------------------------------------------- G8
def phantom_bca(rng, shape=(96, 48, 48)):
    Z, Y, X = shape
    zz, yy, xx = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    r = np.sqrt(((yy - Y / 2) / (Y * 0.45)) ** 2 + ((xx - X / 2) / (X * 0.42)) ** 2)
    regions = np.zeros(shape, dtype=np.uint8)
    regions[r < 1.0] = 1                                   # subcutaneous
    regions[r < 0.85] = 2                                  # muscle
    regions[(r < 0.6) & (zz < 50)] = 3                     # abdominal cavity
    regions[(r < 0.5) & (zz >= 44)] = 4                    # thoracic cavity (overlap -> thorax wins)
    regions[(r < 0.3) & (zz >= 48) & (zz < 70)] = 9         # mediastinum
    regions[(r < 0.15) & (zz >= 52) & (zz < 64)] = 7        # pericardium
    regions[(np.abs(yy - Y * 0.75) < 3) & (np.abs(xx - X / 2) < 3)] = 5  # bone (spine)
    regions[(r < 0.1) & (zz < 4)] = 6
    parts = np.zeros(shape, dtype=np.uint8)
    parts[r < 1.0] = 1
    parts[(xx < X * 0.18) & (r < 1.0)] = 5
    parts[(xx > X * 0.82) & (r < 1.0)] = 6
    parts[zz >= 90] = 2
    ct = np.full(shape, -1000, dtype=np.int16)
    body = r < 1.0
    ct[body] = rng.normal(-100, 40, size=body.sum()).astype(np.int16)
    m = regions == 2
    ct[m] = rng.normal(30, 60, size=m.sum()).astype(np.int16)
    m = regions == 5
    ct[m] = rng.normal(600, 300, size=m.sum()).astype(np.int16)
    for rg in (3, 4, 9, 7):
        m = regions == rg
        ct[m] = rng.normal(-60, 90, size=m.sum()).astype(np.int16)
    return ct, regions, parts


def g8_bca():
    import types
    fake = H.FakeSitk("SimpleITK")
    sys.modules["SimpleITK"] = fake
    import importlib
    import body_composition_analysis.tissue.subclassification as S
    S.sitk = fake
    import body_composition_analysis.report.builder as B
    B.sitk = fake
    import pathlib
    rng = np.random.default_rng(42)
    ct, regions, parts = phantom_bca(rng)
    spacing = (0.8, 0.8, 5.0)
    img = fake.Image(ct, spacing); reg = fake.Image(regions, spacing); prt = fake.Image(parts, spacing)
    tis = S.subclassify_tissues(img, reg, pathlib.Path("/tmp/x"), False, None)
    tis_med = S.subclassify_tissues(img, reg, pathlib.Path("/tmp/x"), True, "LPS")
    tis.spacing = spacing

    # no-op the rendering calls in Builder.prepare / generate_aggregated_measurements
    class _Img:
        def to_image(self, **k):
            return b""
    B.create_tissue_summary = lambda *a, **k: _Img()
    B.create_tissue_heatmaps = lambda *a, **k: []
    B.create_equidistant_overview = lambda *a, **k: []
    B.create_aggregation_image = lambda *a, **k: np.zeros((2, 2, 3), np.uint8)
    B.to_png_data_url = lambda *a, **k: ""
    B.jinja2 = types.SimpleNamespace(Environment=lambda **k: None, FileSystemLoader=lambda *a, **k: None,
                                     select_autoescape=lambda *a, **k: None)
    b = B.Builder(image=img, body_parts=prt, body_regions=reg, tissues=tis)
    b.examined_body_part = B.AggregatableBodyPart.from_body_regions(reg)
    vertebrae = {"L3": (18, 24), "T12": (40, 45)}
    prep = b.prepare(vertebrae=vertebrae, total=None, total_measurements=None)
    js = b.create_json(**prep)
    save_npz("g8_bca.npz", ct=ct, regions=regions, parts=parts, tissues=tis.arr, tissues_median=tis_med.arr,
             spacing=np.array(spacing))
    save_json("g8_bca_measurements.json", {"json": js, "vertebrae": vertebrae})


# ---------------------------------------------------------------------------------------------- G9
def g9_measurements():
    sys.modules.setdefault("SimpleITK", H.FakeSitk("SimpleITK"))
    try:
        import body_organ_analysis.compute.measurements as M
    except Exception:
        import body_organ_analysis.compute.measurements as M
    rng = np.random.default_rng(9)
    shape = (20, 48, 48)
    ct = rng.normal(0, 300, size=shape).astype(np.int16)
    lab = rng.integers(0, 9, size=(5, 12, 12)).repeat(4, 0).repeat(4, 1).repeat(4, 2).astype(np.uint8)
    lab[lab == 8] = 0
    label_map = {"spleen": 1, "aorta": 2, "autochthon_left": 3, "autochthon_right": 4, "lung_upper_lobe_left": 5,
                 "absent_organ": 7, "single": 6}
    lab[lab == 6] = 0
    lab[3, 3, 3] = 6
    spacing = np.array([0.9, 0.9, 2.5])
    res = M.metrics_for_each_region(ct_data=ct, region_data=lab, label_map=label_map, autochthon_mean=41.5,
                                    autochthon_std=17.25, img_spacing=spacing)
    res_none = M.metrics_for_each_region(ct_data=ct, region_data=lab, label_map={"spleen": 1}, autochthon_mean=None,
                                         autochthon_std=None, img_spacing=spacing)

    def clean(d):
        return {k: {kk: (None if vv is None else (bool(vv) if isinstance(vv, (bool, np.bool_)) else float(vv)))
                    for kk, vv in v.items()} for k, v in d.items()}
    # pulmonary-fat style mask through compute_lung_measurement
    fat, lm = M.compute_lung_measurement(ct_data=ct, region_data=lab, ids=[5, 1], autochthon_mean=41.5,
                                         autochthon_std=17.25, img_spacing=spacing)
    save_npz("g9_measurements.npz", ct=ct, lab=lab, spacing=spacing, fat_mask=fat.astype(np.uint8))
    save_json("g9_measurements.json", {"label_map": label_map, "with_ref": clean(res), "no_ref": clean(res_none),
                                       "lung": clean({"x": lm})["x"], "auto": [41.5, 17.25]})


# ---------------------------------------------------------------------------------------------- G10
def g10_config():
    from body_organ_analysis.compute.config import resolve_models, resolve_device
    from body_organ_analysis.compute.util import convert_resampling_slices
    from body_organ_analysis.compute.constants import ALL_MODELS, BASE_MODELS, LICENSE_MODELS, AVAILABLE_MODELS
    specs = [None, "all", "ALL", "", "total+body_parts", "bca", "body-parts", "body+total", "total+bca",
             "bca+body_regions+lung_vessels", "heartchambers_highres", "total+total"]
    rm = {str(s): sorted(resolve_models(s)) for s in specs}
    devs = {}
    for d in [None, "cpu", "gpu", "cuda", "gpu:2", "cuda:1", "mps"]:
        env = dict(os.environ)
        for k in ("DEVICE", "NVIDIA_ID", "NVIDIA_VISIBLE_DEVICES"):
            os.environ.pop(k, None)
        devs[str(d)] = resolve_device(d)
        os.environ.clear(); os.environ.update(env)
    crs = [[s, c, t, convert_resampling_slices(s, c, t)] for s, c, t in
           [(512, 1.5, 1.5), (768, 1.5, 5.0), (300, 0.7, 1.5), (1600, 1.0, 5.0), (411, 2.5, None), (101, 1.25, 5.0)]]
    save_json("g10_config.json", {"resolve_models": rm, "resolve_device": devs, "convert_resampling_slices": crs,
                                  "ALL_MODELS": sorted(ALL_MODELS), "BASE_MODELS": sorted(BASE_MODELS),
                                  "LICENSE_MODELS": sorted(LICENSE_MODELS), "AVAILABLE_MODELS": sorted(AVAILABLE_MODELS)})


# ---------------------------------------------------------------------------------------------- G11
def g11_measurement_label_maps():
    """The per-model label maps exactly as compute_measurements builds them (BOA/compute/measurements.py:16-18,
    289-293: every `class_map` task whose name starts with the model name contributes, e.g. `total_mr_*` / `total_v1_*`
    entries appear under "total" as `mr_*` / `v1_*`), the class maps of the tasks BOA can run and the output names."""
    from body_organ_analysis.compute.measurements import CNR_ADJUSTED_REGIONS, reverse_class_map_complete
    from body_organ_analysis.compute.util import ADDITIONAL_MODELS_OUTPUT_NAME
    from totalsegmentator.map_to_binary import class_map
    models = ["total", "lung_vessels", "cerebral_bleed", "hip_implant", "coronary_arteries", "pleural_pericard_effusion",
              "liver_vessels", "heartchambers_highres", "body_parts", "body_regions"]
    lm = {}
    for m in models:
        # [name, id] pairs in the reference's dict order (the order of the regions in total-measurements.json)
        lm[m] = list({k[len(m) + 1:]: v for k, v in reverse_class_map_complete.items()
                      if k.startswith(m) and not k.startswith(m + "_v2")}.items())
    cm = {m: {str(k): v for k, v in class_map[m].items()} for m in models if m in class_map}
    save_json("g11_measurement_label_maps.json",
              {"label_maps": lm, "class_maps": cm, "output_names": ADDITIONAL_MODELS_OUTPUT_NAME,
               "cnr_adjusted_regions": {k: sorted(v) for k, v in CNR_ADJUSTED_REGIONS.items()}})


# ---------------------------------------------------------------------------------------------- G12
def g12_cropping():
    """Bounding-box helpers at the reference's call sites: TS/cropping.py get_bbox_from_mask (:11-38, called from
    crop_to_mask :75-110 with the mm -> voxel addon) + crop_to_bbox (:41-49), and the nonzero mask of nnU-Net's
    crop_to_nonzero (NN/preprocessing/cropping/cropping.py:6-17: data != 0 followed by binary_fill_holes; its bbox is what
    the un-vendored acvl_utils helper turns into slices)."""
    from totalsegmentator.cropping import get_bbox_from_mask, crop_to_bbox
    from nnunetv2.preprocessing.cropping.cropping import create_nonzero_mask
    rng = np.random.default_rng(12)
    out = {}
    cases = []
    for i, (shape, addon, ov) in enumerate([((20, 17, 23), 0, 0), ((20, 17, 23), [3, 1, 2], 0), ((9, 30, 12), [13, 13, 33], 0),
                                            ((16, 16, 16), 2, 0), ((12, 10, 8), [1, 1, 1], -900), ((7, 7, 7), [20, 20, 20], 0)]):
        m = np.zeros(shape, dtype=np.uint8)
        if i != 3:                                    # case 3: empty mask ("Could not crop")
            lo = [int(rng.integers(0, s // 2)) for s in shape]
            hi = [int(rng.integers(l + 1, s + 1)) for l, s in zip(lo, shape)]
            blob = rng.random(tuple(h - l for l, h in zip(lo, hi))) > 0.6
            blob.flat[0] = True
            m[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = blob
        if ov == -900:
            m = (m.astype(np.int16) * 1000 - 950)     # HU-like image with the default outside_value
        bbox = get_bbox_from_mask(m, outside_value=ov, addon=addon)
        img = rng.integers(-1000, 1000, size=shape).astype(np.int16)
        out[f"c{i}_mask"] = m
        out[f"c{i}_img"] = img
        out[f"c{i}_addon"] = np.array([addon] * 3 if isinstance(addon, int) else addon)
        out[f"c{i}_outside"] = np.array(ov)
        out[f"c{i}_bbox"] = np.array(bbox)
        out[f"c{i}_crop"] = crop_to_bbox(img, bbox)
        cases.append(i)
    # crop_to_nonzero's mask: hollow shell (holes get filled, bbox unchanged), two blobs, all zero
    for j, kind in enumerate(["shell", "blobs", "zero"]):
        d = np.zeros((1, 14, 15, 16), dtype=np.float32)
        if kind == "shell":
            d[0, 3:11, 2:12, 4:13] = 1.5
            d[0, 5:9, 4:10, 6:11] = 0.0
        elif kind == "blobs":
            d[0, 1:3, 1:4, 2:5] = -2.0
            d[0, 9:13, 8:14, 10:15] = 7.0
        mask = create_nonzero_mask(d)
        out[f"n{j}_data"] = d
        out[f"n{j}_mask"] = mask.astype(np.uint8)
    out["n_cases"] = np.array([len(cases), 3])
    save_npz("g12_cropping.npz", **out)


# ---------------------------------------------------------------------------------------------- G13
def g13_nnunet_resampling():
    """The reference's own resample_data_or_seg_to_shape (NN/preprocessing/resampling/default_resampling.py:83-196) and its
    helpers, executed with `skimage.transform.resize` (absent here) replaced by the oracle's restatement of its published
    algorithm (oracle/nnunet_resample.py:skimage_resize = scipy.ndimage.zoom(grid_mode=True, mode="nearest") + clip).  Pins
    the axis choice, the per-slice loop, the nearest sampling along the anisotropic axis and the dtypes -- NOT skimage."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.nnunet_resample import skimage_resize
    import nnunetv2.preprocessing.resampling.default_resampling as dr

    def resize(image, output_shape, order=None, mode="reflect", anti_aliasing=None, **kw):
        assert mode == "edge" and anti_aliasing is False and not kw, (mode, anti_aliasing, kw)
        return skimage_resize(image, output_shape, order)
    dr.resize = resize
    rng = np.random.default_rng(13)
    out = {}
    cases = [  # (name, data shape, dtype, current spacing, new spacing, order)
        ("iso3d", (1, 14, 18, 16), np.float32, (1.5, 1.5, 1.5), (1.2, 1.0, 1.4), 3),
        ("sepz_same_z", (1, 6, 20, 22), np.float32, (5.0, 0.8, 0.8), (5.0, 1.0, 0.9), 3),
        ("sepz_new_z", (1, 7, 20, 18), np.float32, (5.0, 0.8, 0.8), (3.0, 1.0, 1.0), 3),
        ("logits_back", (3, 9, 16, 17), np.float16, (5.0, 1.0, 0.9), (5.0, 0.8, 0.8), 1),
        ("logits_iso", (4, 12, 10, 11), np.float16, (1.0, 1.0, 1.0), (1.5, 1.5, 1.5), 1),
        ("new_aniso", (1, 20, 16, 16), np.float32, (1.0, 1.0, 1.0), (4.0, 1.0, 1.0), 3),
        ("identity", (1, 8, 9, 10), np.float32, (1.5, 1.5, 1.5), (1.5, 1.5, 1.5), 3),
    ]
    names = []
    for name, shp, dt, cur, new, order in cases:
        d = (rng.standard_normal(shp) * 3).astype(dt)
        new_shape = dr.compute_new_shape(shp[1:], cur, new)
        sep, axis = dr.determine_do_sep_z_and_axis(None, cur, new)
        res = dr.resample_data_or_seg_to_shape(d, new_shape, cur, new, is_seg=False, order=order, order_z=0, force_separate_z=None)
        assert res.dtype == dt
        out[f"{name}_in"] = d.view(np.uint16) if dt == np.float16 else d
        out[f"{name}_out"] = res.view(np.uint16) if dt == np.float16 else res
        out[f"{name}_meta"] = np.array([*cur, *new, order, int(sep), -1 if axis is None else int(axis), *new_shape], dtype=np.float64)
        names.append(name)
    out["names"] = np.array(names)
    # decision table of determine_do_sep_z_and_axis / compute_new_shape on awkward spacings
    tab = []
    for cur, new in [((5.0, 0.78, 0.78), (5.0, 0.8, 0.8)), ((0.24, 1.25, 1.25), (1.0, 1.0, 1.0)), ((3.0, 1.0, 1.0), (1.0, 1.0, 1.0)),
                     ((3.01, 1.0, 1.0), (1.0, 1.0, 1.0)), ((1.0, 1.0, 1.0), (1.0, 1.0, 6.0)), ((2.0, 2.0, 2.0), (2.0, 2.0, 2.0)),
                     ((1.5, 0.5, 1.5), (1.0, 1.0, 1.0)), ((0.7, 0.7, 5.0), (0.7, 0.7, 5.0))]:
        sep, axis = dr.determine_do_sep_z_and_axis(None, cur, new)
        tab.append([*cur, *new, int(sep), -1 if axis is None else int(axis), *dr.compute_new_shape((37, 201, 199), cur, new)])
    out["decisions"] = np.array(tab, dtype=np.float64)
    save_npz("g13_nnunet_resampling.npz", **out)


def g14_overview():
    """BCA/report/plots/check.py create_equidistant_overview (five slices: HU window + colour overlay at 25 % opacity) with the
    reference's own apply_hu_window / blend_overlay (BCA/report/plots/overlay.py).  SimpleITK is absent: its two calls here are
    `GetArrayViewFromImage`, which returns the (z,y,x) array of the image -- the harness hands the arrays over directly."""
    import SimpleITK as sitk  # the stub
    from body_composition_analysis.report.plots import check
    sitk.GetArrayViewFromImage = lambda a: a
    check.sitk.GetArrayViewFromImage = lambda a: a
    rng = np.random.default_rng(14)
    out = {}
    for i, shape in enumerate([(11, 20, 24), (2, 9, 7), (37, 16, 16)]):
        img = rng.integers(-1100, 1600, size=shape).astype(np.int16)
        segs, cmaps = [], []
        for k, nlab in enumerate((5, 12)):
            seg = (rng.integers(0, nlab, size=shape) * (rng.random(shape) < 0.6)).astype(np.uint8)
            cmap = {l: tuple(int(v) for v in rng.integers(0, 256, 3)) for l in range(nlab)}
            cmap[0] = (0, 0, 0)
            segs.append(seg)
            cmaps.append(np.array([cmap[l] for l in range(nlab)], dtype=np.uint8))
        res = check.create_equidistant_overview(img, [(s, [tuple(c) for c in cm]) for s, cm in zip(segs, cmaps)])
        out[f"c{i}_img"] = img
        for k in range(2):
            out[f"c{i}_seg{k}"] = segs[k]
            out[f"c{i}_cmap{k}"] = cmaps[k]
        out[f"c{i}_names"] = np.array([r[0] for r in res])
        out[f"c{i}_out"] = np.stack([np.stack(r[1:]) for r in res])      # [5 slices][2 segmentations][Y][X][3] float64
    save_npz("g14_overview.npz", **out)


# ---------------------------------------------------------------------------------------------- G15
def _nifti_expect(raw: bytes, endian="<"):
    """Independent field-by-field parse of a NIfTI-1 blob (offsets of the NIfTI-1.1 standard's header table) and the
    affine / scaled data nibabel's `load(...).affine` / `.get_fdata()` documents: sform when sform_code > 0, else the
    quaternion form when qform_code > 0; data * scl_slope + scl_inter unless slope is 0 / (1, 0)."""
    e = endian
    dim = struct.unpack(e + "8h", raw[40:56])
    datatype, bitpix = struct.unpack(e + "hh", raw[70:74])
    pixdim = struct.unpack(e + "8f", raw[76:108])
    vox_offset, slope, inter = struct.unpack(e + "3f", raw[108:120])
    qform_code, sform_code = struct.unpack(e + "hh", raw[252:256])
    qb, qc, qd, qx, qy, qz = struct.unpack(e + "6f", raw[256:280])
    srow = struct.unpack(e + "12f", raw[280:328])
    if sform_code > 0:
        aff = [list(srow[0:4]), list(srow[4:8]), list(srow[8:12]), [0.0, 0.0, 0.0, 1.0]]
    elif qform_code > 0:
        qa = max(0.0, 1.0 - (qb * qb + qc * qc + qd * qd)) ** 0.5
        R = [[qa * qa + qb * qb - qc * qc - qd * qd, 2 * qb * qc - 2 * qa * qd, 2 * qb * qd + 2 * qa * qc],
             [2 * qb * qc + 2 * qa * qd, qa * qa + qc * qc - qb * qb - qd * qd, 2 * qc * qd - 2 * qa * qb],
             [2 * qb * qd - 2 * qa * qc, 2 * qc * qd + 2 * qa * qb, qa * qa + qd * qd - qc * qc - qb * qb]]
        qfac = -1.0 if pixdim[0] < 0 else 1.0
        z = [pixdim[1], pixdim[2], pixdim[3] * qfac]
        aff = [[R[i][j] * z[j] for j in range(3)] + [(qx, qy, qz)[i]] for i in range(3)] + [[0.0, 0.0, 0.0, 1.0]]
    else:
        aff = None
    np_dt = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8"}[datatype]
    n = dim[1] * dim[2] * dim[3]
    a = np.frombuffer(raw, dtype=e + np_dt, count=n, offset=int(vox_offset)).reshape(dim[3], dim[2], dim[1]).transpose(2, 1, 0)
    a = np.ascontiguousarray(a).astype(np_dt)     # (x, y, z) = file axis order, native byte order
    f = a.astype(np.float64)
    if np.isfinite(slope) and slope != 0 and not (slope == 1.0 and inter == 0.0):
        f = f * np.float64(slope) + np.float64(inter)
    return {"shape": list(dim[1:4]), "datatype": datatype, "bitpix": bitpix, "zooms": list(pixdim[1:4]), "qform_code": qform_code,
            "sform_code": sform_code, "affine": aff, "scl_slope": slope, "scl_inter": inter,
            "voxels_sha256": hashlib.sha256(a.tobytes()).hexdigest(), "voxel_sum": int(a.astype(np.int64).sum()),
            "voxel_min": float(a.min()), "voxel_max": float(a.max()),
            "fdata_sum": float(f.sum()), "fdata_at": {"0,0,0": float(f[0, 0, 0]), "60,50,15": float(f[60, 50, 15]), "121,100,29": float(f[121, 100, 29])}}


def nifti_variants(raw: bytes):
    """Byte-patched variants of a NIfTI-1 blob (same patches applied by tests/test_host_cpu.py): name -> (blob, endian)."""
    out = {"orig": (raw, "<")}
    b = bytearray(raw)                                  # qform only: sform_code 0, a 90-degree-about-z quaternion, qfac -1
    b[252:256] = struct.pack("<hh", 1, 0)
    b[76:80] = struct.pack("<f", -1.0)
    b[256:280] = struct.pack("<6f", 0.0, 0.0, 0.70710678, 10.5, -20.25, 7.0)
    out["qform_only"] = (bytes(b), "<")
    b = bytearray(raw)                                  # neither code set: nibabel falls back to the pixdim / centre affine
    b[252:256] = struct.pack("<hh", 0, 0)
    out["no_form"] = (bytes(b), "<")
    b = bytearray(raw)                                  # rescale slope / intercept (get_fdata applies them)
    b[112:120] = struct.pack("<ff", 2.0, -1024.0)
    out["scaled"] = (bytes(b), "<")
    # big-endian copy of the whole file (header fields and voxels byte-swapped)
    hdr_fmt = "i10s18sihcB8h3f4h8f3fhBB4f2i80s24s2h3f3f12f16s4s"
    v = struct.unpack("<" + hdr_fmt, raw[:348])
    dim = v[7:15]
    n = dim[1] * dim[2] * dim[3]
    off = int(v[30])
    vox = np.frombuffer(raw, dtype="<i2", count=n, offset=off).astype(">i2").tobytes()
    out["big_endian"] = (struct.pack(">" + hdr_fmt, *v) + raw[348:off] + vox, ">")
    return out


def g15_nifti():
    """The reference's own test volumes (NN/tests/example_data/*.nii.gz: data files, committed as fixtures) and what
    NN/imageio/nibabel_reader_writer.py:38-99 gets from nibabel for them (shape, zooms, affine, dtype, voxels), restated field by
    field from the NIfTI-1 standard -- nibabel itself is absent from the image (PARITY UNPINNED vs nibabel's code, pinned vs the
    standard and vs files this repo did not write)."""
    import shutil
    src = os.path.join(H.EXT, "nnunetv2", "tests", "example_data")
    exp = {}
    for name in ("example_ct_sm.nii.gz", "example_ct_sm_T300_output.nii.gz"):
        shutil.copyfile(os.path.join(src, name), os.path.join(HERE, "ref_" + name))
        os.chmod(os.path.join(HERE, "ref_" + name), 0o644)
        raw = gzip.open(os.path.join(src, name), "rb").read()
        exp[name] = {"orig": _nifti_expect(raw)}
        if name == "example_ct_sm.nii.gz":
            for k, (blob, en) in nifti_variants(raw).items():
                exp[name][k] = _nifti_expect(blob, en)
            sh = exp[name]["no_form"]["shape"]
            # nibabel's fallback (Nifti1Header.get_base_affine -> shape_zoom_affine(shape, zooms, x_flip=True), published
            # behaviour): diag(-zx, zy, zz), origin at the volume centre
            z = exp[name]["no_form"]["zooms"]
            z = [-z[0], z[1], z[2]]
            exp[name]["no_form"]["affine"] = [[z[0], 0, 0, -(sh[0] - 1) / 2.0 * z[0]], [0, z[1], 0, -(sh[1] - 1) / 2.0 * z[1]],
                                              [0, 0, z[2], -(sh[2] - 1) / 2.0 * z[2]], [0, 0, 0, 1.0]]
    save_json("g15_nifti.json", exp)


if __name__ == "__main__":
    ct = load_example_ct()
    print("example ct", ct.shape, ct.dtype, ct.min(), ct.max())
    only = sys.argv[1:]
    fns = dict(g1=g1_steps, g2=g2_gaussian, g3=g3_sliding_window, g3b=g3b_fold_ensemble, g4=lambda: g4_ctnorm(ct),
               g5=lambda: g5_resample(ct), g67=g67_argmax_merge, g10=g10_config, g9=g9_measurements, g8=g8_bca,
               g11=g11_measurement_label_maps, g12=g12_cropping, g13=g13_nnunet_resampling, g14=g14_overview, g15=g15_nifti)
    for k, f in fns.items():
        if not only or k in only:
            f()
