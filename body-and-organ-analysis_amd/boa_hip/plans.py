"""plans.json / dataset.json / checkpoint -> device network geometry and weight blob.

Mirrors what the reference reads from a trained-model folder:
  NN/inference/predict_from_raw_data.py:67-129  initialize_from_trained_model_folder (dataset.json, plans.json,
      fold_k/checkpoint_final.pth with keys network_weights / trainer_name / init_args.configuration /
      inference_allowed_mirroring_axes)
  NN/utilities/plans_handling/plans_handler.py:32-97   old-format -> architecture dict reconstruction
  NN/utilities/plans_handling/plans_handler.py:214-325 PlansManager (transpose_forward/backward, intensity properties)
  NN/utilities/label_handling/label_handling.py       num_segmentation_heads = number of labels (no regions)
torch is used only to unpickle checkpoints (`load_checkpoint`).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from ._lib import MAX_STAGES, NetDesc


@dataclass
class NetGeometry:
    in_channels: int
    num_classes: int
    features: List[int]
    kernels: List[List[int]]
    strides: List[List[int]]
    n_conv_enc: List[int]
    n_conv_dec: List[int]
    patch_size: List[int]
    norm_eps: float = 1e-5
    lrelu_slope: float = 0.01

    @property
    def n_stages(self):
        return len(self.features)

    def to_desc(self) -> NetDesc:
        if self.n_stages > MAX_STAGES:
            raise ValueError(f"{self.n_stages} stages > {MAX_STAGES}")
        d = NetDesc()
        d.n_stages = self.n_stages
        d.in_channels = self.in_channels
        d.num_classes = self.num_classes
        for s in range(self.n_stages):
            d.features[s] = int(self.features[s])
            d.n_conv_enc[s] = int(self.n_conv_enc[s])
            for a in range(3):
                d.kernel[s][a] = int(self.kernels[s][a])
                d.stride[s][a] = int(self.strides[s][a])
        for s in range(self.n_stages - 1):
            d.n_conv_dec[s] = int(self.n_conv_dec[s])
        for a in range(3):
            d.patch[a] = int(self.patch_size[a])
        d.norm_eps = float(self.norm_eps)
        d.lrelu_slope = float(self.lrelu_slope)
        return d

    def flops_per_tile(self) -> float:
        """2 * MACs of one tile forward (convs, transposed convs, head)."""
        dims = list(self.patch_size)
        cin = self.in_channels
        macs = 0.0
        stage_dims = []
        for s in range(self.n_stages):
            taps = int(np.prod(self.kernels[s]))
            for i in range(self.n_conv_enc[s]):
                if i == 0:
                    dims = [(d + 2 * ((k - 1) // 2) - k) // st + 1 for d, k, st in zip(dims, self.kernels[s], self.strides[s])]
                macs += np.prod(dims) * taps * cin * self.features[s]
                cin = self.features[s]
            stage_dims.append(list(dims))
        for k in range(self.n_stages - 1):
            sb = self.n_stages - 1 - k
            below, skip = self.features[sb], self.features[sb - 1]
            vox = np.prod(stage_dims[sb - 1])
            macs += vox * below * skip
            taps = int(np.prod(self.kernels[sb - 1]))
            ci = 2 * skip
            for i in range(self.n_conv_dec[k]):
                macs += vox * taps * ci * skip
                ci = skip
        macs += np.prod(self.patch_size) * self.features[0] * self.num_classes
        return 2.0 * float(macs)

    def activation_bytes_per_tile(self) -> float:
        """fp16 bytes if every layer reads its input once and writes its output once."""
        dims = list(self.patch_size)
        total = 0.0
        cin = self.in_channels
        stage_dims = []
        for s in range(self.n_stages):
            for i in range(self.n_conv_enc[s]):
                vin = np.prod(dims)
                if i == 0:
                    dims = [(d + 2 * ((k - 1) // 2) - k) // st + 1 for d, k, st in zip(dims, self.kernels[s], self.strides[s])]
                total += 2.0 * (vin * cin + np.prod(dims) * self.features[s])
                cin = self.features[s]
            stage_dims.append(list(dims))
        for k in range(self.n_stages - 1):
            sb = self.n_stages - 1 - k
            below, skip = self.features[sb], self.features[sb - 1]
            vox = np.prod(stage_dims[sb - 1])
            total += 2.0 * (np.prod(stage_dims[sb]) * below + vox * skip)
            ci = 2 * skip
            for i in range(self.n_conv_dec[k]):
                total += 2.0 * vox * (ci + skip)
                ci = skip
        total += 2.0 * np.prod(self.patch_size) * (self.features[0] + self.num_classes)
        return float(total)


@dataclass
class ModelConfig:
    geometry: NetGeometry
    spacing: List[float]
    transpose_forward: List[int]
    transpose_backward: List[int]
    normalization_schemes: List[str]
    intensity_properties: Dict[str, Dict[str, float]]
    labels: Dict[str, int]
    configuration_name: str = "3d_fullres"
    extra: dict = field(default_factory=dict)


def _arch_from_configuration(cfg: dict) -> dict:
    """New-format `architecture.arch_kwargs`, or the reconstruction of plans_handler.py:36-97 for old plans."""
    if "architecture" in cfg:
        arch = cfg["architecture"]
        name = arch["network_class_name"]
        if not name.endswith("PlainConvUNet"):
            raise ValueError(f"unsupported network class {name!r}: only PlainConvUNet is implemented on device")
        kw = arch["arch_kwargs"]
        return dict(n_stages=kw["n_stages"], features_per_stage=list(kw["features_per_stage"]),
                    kernel_sizes=[list(k) for k in kw["kernel_sizes"]], strides=[_as3(s) for s in kw["strides"]],
                    n_conv_per_stage=kw["n_conv_per_stage"], n_conv_per_stage_decoder=kw["n_conv_per_stage_decoder"],
                    norm_eps=(kw.get("norm_op_kwargs") or {}).get("eps", 1e-5),
                    slope=(kw.get("nonlin_kwargs") or {}).get("negative_slope", 0.01))
    if cfg.get("UNet_class_name") != "PlainConvUNet":
        raise ValueError(f"unsupported UNet_class_name {cfg.get('UNet_class_name')!r}")
    n_stages = len(cfg["n_conv_per_stage_encoder"])
    return dict(
        n_stages=n_stages,
        features_per_stage=[min(cfg["UNet_base_num_features"] * 2 ** i, cfg["unet_max_num_features"]) for i in range(n_stages)],
        kernel_sizes=[list(k) for k in cfg["conv_kernel_sizes"]], strides=[_as3(s) for s in cfg["pool_op_kernel_sizes"]],
        n_conv_per_stage=list(cfg["n_conv_per_stage_encoder"]), n_conv_per_stage_decoder=list(cfg["n_conv_per_stage_decoder"]),
        norm_eps=1e-5, slope=0.01)


def _as3(s):
    return [int(s)] * 3 if isinstance(s, (int, float)) else [int(v) for v in s]


def _resolve_configuration(plans: dict, name: str) -> dict:
    cfg = dict(plans["configurations"][name])
    if "inherits_from" in cfg:  # plans_handler.py:231-247
        parent = _resolve_configuration(plans, cfg["inherits_from"])
        parent.update(cfg)
        parent.pop("inherits_from", None)
        cfg = parent
    return cfg


def model_config_from_plans(plans: dict, dataset_json: dict, configuration: str = "3d_fullres") -> ModelConfig:
    cfg = _resolve_configuration(plans, configuration)
    arch = _arch_from_configuration(cfg)
    n = arch["n_stages"]
    nce = arch["n_conv_per_stage"]
    ncd = arch["n_conv_per_stage_decoder"]
    nce = [nce] * n if isinstance(nce, int) else list(nce)
    ncd = [ncd] * (n - 1) if isinstance(ncd, int) else list(ncd)
    if len(cfg["patch_size"]) != 3:
        raise ValueError("only 3-D configurations are supported")
    labels = dataset_json["labels"]
    if any(isinstance(v, (list, tuple)) for v in labels.values()):
        raise ValueError("region-based labels are not supported")
    chans = dataset_json.get("channel_names", dataset_json.get("modality"))
    geom = NetGeometry(in_channels=len(chans), num_classes=len(labels), features=arch["features_per_stage"],
                       kernels=arch["kernel_sizes"], strides=arch["strides"], n_conv_enc=nce, n_conv_dec=ncd,
                       patch_size=list(cfg["patch_size"]), norm_eps=arch["norm_eps"], lrelu_slope=arch["slope"])
    ip = plans.get("foreground_intensity_properties_per_channel", plans.get("foreground_intensity_properties_by_modality", {}))
    return ModelConfig(geometry=geom, spacing=list(cfg["spacing"]), transpose_forward=list(plans.get("transpose_forward", [0, 1, 2])),
                       transpose_backward=list(plans.get("transpose_backward", [0, 1, 2])),
                       normalization_schemes=list(cfg.get("normalization_schemes", ["CTNormalization"])),
                       intensity_properties={str(k): v for k, v in ip.items()}, labels=dict(labels),
                       configuration_name=configuration,
                       extra={k: cfg[k] for k in cfg if k.startswith("resampling_fn_")})


def load_model_folder(model_folder: str, configuration: Optional[str] = None):
    """Read dataset.json + plans.json of `<trainer>__nnUNetPlans__<config>` (file_path_utilities.py:19-26)."""
    with open(os.path.join(model_folder, "dataset.json")) as f:
        dataset_json = json.load(f)
    with open(os.path.join(model_folder, "plans.json")) as f:
        plans = json.load(f)
    if configuration is None:
        configuration = os.path.basename(os.path.normpath(model_folder)).split("__")[-1]
    return model_config_from_plans(plans, dataset_json, configuration)


def load_checkpoint(path: str) -> dict:
    """torch.load(checkpoint_final.pth) -> {name: float32 ndarray} of `network_weights`
    (predict_from_raw_data.py:86-94)."""
    import torch
    ck = torch.load(path, map_location="cpu", weights_only=False)
    sd = ck["network_weights"] if "network_weights" in ck else ck
    return {k: v.detach().cpu().numpy().astype(np.float32) for k, v in sd.items()}


def weight_blob_from_state_dict(geom: NetGeometry, sd: Dict[str, np.ndarray]) -> np.ndarray:
    """Concatenate the tensors in the order boa_net_create expects (include/boa_hip.h).  Missing or mis-shaped
    keys raise KeyError / ValueError (the loader must fail loudly on unexpected checkpoints)."""
    sd = {k[len("_orig_mod."):] if k.startswith("_orig_mod.") else k: v for k, v in sd.items()}
    out = []
    used = set()

    def take(key, shape):
        if key not in sd:
            raise KeyError(f"checkpoint lacks {key!r}")
        a = np.asarray(sd[key], dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{key}: shape {a.shape}, expected {tuple(shape)}")
        used.add(key)
        out.append(a.reshape(-1))

    cin = geom.in_channels
    for s in range(geom.n_stages):
        f = geom.features[s]
        for i in range(geom.n_conv_enc[s]):
            p = f"encoder.stages.{s}.0.convs.{i}"
            take(f"{p}.conv.weight", (f, cin, *geom.kernels[s]))
            take(f"{p}.conv.bias", (f,))
            take(f"{p}.norm.weight", (f,))
            take(f"{p}.norm.bias", (f,))
            cin = f
    for k in range(geom.n_stages - 1):
        sb = geom.n_stages - 1 - k
        below, skip = geom.features[sb], geom.features[sb - 1]
        take(f"decoder.transpconvs.{k}.weight", (below, skip, *geom.strides[sb]))
        take(f"decoder.transpconvs.{k}.bias", (skip,))
        ci = 2 * skip
        for i in range(geom.n_conv_dec[k]):
            p = f"decoder.stages.{k}.convs.{i}"
            take(f"{p}.conv.weight", (skip, ci, *geom.kernels[sb - 1]))
            take(f"{p}.conv.bias", (skip,))
            take(f"{p}.norm.weight", (skip,))
            take(f"{p}.norm.bias", (skip,))
            ci = skip
    last = geom.n_stages - 2
    take(f"decoder.seg_layers.{last}.weight", (geom.num_classes, geom.features[0], 1, 1, 1))
    take(f"decoder.seg_layers.{last}.bias", (geom.num_classes,))
    _check_leftover_keys(sd, used, last)
    return np.ascontiguousarray(np.concatenate(out), dtype=np.float32)


def _check_leftover_keys(sd: Dict[str, np.ndarray], used: set, last_seg: int) -> None:
    """Every key the blob did not consume must be one the upstream PlainConvUNet is known to save IN ADDITION (module aliases of
    the same tensors, the deep-supervision heads of the coarser decoder levels) -- and an alias must hold the same numbers as the
    tensor it aliases.  Anything else means the checkpoint is not the architecture the plans describe: fail loudly
    (the key names are upstream knowledge, unpinned in the reference tree).
      encoder.stages.S.0.convs.I.all_modules.{0,1}.*   = ...convs.I.{conv,norm}.*   (nn.Sequential view of ConvDropoutNormReLU)
      decoder.stages.K.convs.I.all_modules.{0,1}.*     likewise
      decoder.encoder.*                                = encoder.*                  (the decoder keeps a reference to the encoder)
      decoder.seg_layers.K.* for K != last             deep-supervision heads, unused at inference
      *.num_batches_tracked / running_mean / running_var  never present for InstanceNorm(track_running_stats=False): rejected"""
    import re

    def same(a, b):
        return a in sd and b in sd and np.asarray(sd[a]).shape == np.asarray(sd[b]).shape and np.array_equal(sd[a], sd[b])

    for key in sd:
        if key in used:
            continue
        base = key[len("decoder."):] if key.startswith("decoder.encoder.") else key
        m = re.match(r"^(.*\.convs\.\d+)\.all_modules\.([01])\.(weight|bias)$", base)
        if m:
            base = f"{m.group(1)}.{'conv' if m.group(2) == '0' else 'norm'}.{m.group(3)}"
        if base != key:
            if base not in used:
                raise ValueError(f"unexpected checkpoint key {key!r} (aliases {base!r}, which the architecture does not have)")
            if base not in sd or not same(key, base):
                raise ValueError(f"checkpoint key {key!r} should alias {base!r} but holds different values")
            continue
        m = re.match(r"^decoder\.seg_layers\.(\d+)\.(weight|bias)$", key)
        if m and int(m.group(1)) != last_seg:
            continue
        raise ValueError(f"unexpected checkpoint key {key!r}: not part of the PlainConvUNet the plans describe")


def synthetic_plans(patch=(128, 128, 128), features=(32, 64, 128, 256, 320, 320), num_classes=25, in_channels=1,
                    spacing=(1.5, 1.5, 1.5), kernels=None, strides=None):
    """A plans.json / dataset.json pair of the documented `total` 3d_fullres geometry, for benchmarks
    and tests: the real plans ship inside the weight archives, which are not available offline.  Intensity
    properties are PLACEHOLDERS (clearly synthetic)."""
    n = len(features)
    kernels = kernels or [[3, 3, 3]] * n
    strides = strides or ([[1, 1, 1]] + [[2, 2, 2]] * (n - 1))
    plans = {
        "dataset_name": "Dataset000_Synthetic", "plans_name": "nnUNetPlans",
        "transpose_forward": [0, 1, 2], "transpose_backward": [0, 1, 2],
        "image_reader_writer": "NibabelIOWithReorient",
        "foreground_intensity_properties_per_channel": {
            "0": {"mean": -370.0, "std": 436.0, "percentile_00_5": -1004.0, "percentile_99_5": 1588.0,
                  "min": -1024.0, "max": 3071.0, "median": -200.0}},
        "configurations": {"3d_fullres": {
            "data_identifier": "nnUNetPlans_3d_fullres", "preprocessor_name": "DefaultPreprocessor", "batch_size": 2,
            "patch_size": list(patch), "spacing": list(spacing), "normalization_schemes": ["CTNormalization"],
            "use_mask_for_norm": [False],
            "resampling_fn_data": "resample_data_or_seg_to_shape", "resampling_fn_seg": "resample_data_or_seg_to_shape",
            "resampling_fn_probabilities": "resample_data_or_seg_to_shape",
            "architecture": {
                "network_class_name": "dynamic_network_architectures.architectures.unet.PlainConvUNet",
                "arch_kwargs": {
                    "n_stages": n, "features_per_stage": list(features), "conv_op": "torch.nn.modules.conv.Conv3d",
                    "kernel_sizes": [list(k) for k in kernels], "strides": [list(s) for s in strides],
                    "n_conv_per_stage": [2] * n, "n_conv_per_stage_decoder": [2] * (n - 1), "conv_bias": True,
                    "norm_op": "torch.nn.modules.instancenorm.InstanceNorm3d",
                    "norm_op_kwargs": {"eps": 1e-05, "affine": True}, "dropout_op": None, "dropout_op_kwargs": None,
                    "nonlin": "torch.nn.LeakyReLU", "nonlin_kwargs": {"inplace": True}},
                "_kw_requires_import": ["conv_op", "norm_op", "dropout_op", "nonlin"]}}}}
    dataset = {"channel_names": {str(i): "CT" for i in range(in_channels)},
               "labels": {"background": 0, **{f"class_{i}": i for i in range(1, num_classes)}},
               "file_ending": ".nii.gz", "numTraining": 0}
    return plans, dataset


def legacy_plans_from(plans: dict, dataset_json: dict, configuration: str = "3d_fullres"):
    """The same model in the OLDER nnU-Net plans / dataset.json format (tests, model-folder fixtures): the keys
    plans_handler.py:36-97 reconstructs the architecture from (`UNet_class_name`, `UNet_base_num_features`,
    `unet_max_num_features`, `n_conv_per_stage_encoder` / `_decoder`, `conv_kernel_sizes`, `pool_op_kernel_sizes`,
    `num_pool_per_axis`), `foreground_intensity_properties_by_modality` and dataset.json's `modality`
    (plans_handler.py:275-283, label_handling / predict_from_raw_data.py:76).  Only geometries whose feature counts follow
    min(base * 2^i, max) can be written that way."""
    import copy
    pj, dj = copy.deepcopy(plans), copy.deepcopy(dataset_json)
    cfg = pj["configurations"][configuration]
    kw = cfg.pop("architecture")["arch_kwargs"]
    feats = list(kw["features_per_stage"])
    base, cap = feats[0], max(feats)
    if feats != [min(base * 2 ** i, cap) for i in range(len(feats))]:
        raise ValueError(f"features {feats} cannot be expressed in the legacy plans format")
    strides = [list(s) for s in kw["strides"]]
    cfg.update({"UNet_class_name": "PlainConvUNet", "UNet_base_num_features": base, "unet_max_num_features": cap,
                "n_conv_per_stage_encoder": list(kw["n_conv_per_stage"]), "n_conv_per_stage_decoder": list(kw["n_conv_per_stage_decoder"]),
                "conv_kernel_sizes": [list(k) for k in kw["kernel_sizes"]], "pool_op_kernel_sizes": strides,
                "num_pool_per_axis": [sum(1 for s in strides if s[a] > 1) for a in range(3)]})
    if "foreground_intensity_properties_per_channel" in pj:
        pj["foreground_intensity_properties_by_modality"] = pj.pop("foreground_intensity_properties_per_channel")
    if "channel_names" in dj:
        dj["modality"] = dj.pop("channel_names")
    return pj, dj


def synthetic_state_dict(geom: NetGeometry, seed: int = 0, structured: float = 0.0) -> Dict[str, np.ndarray]:
    """Seeded random weights with the upstream key names (Kaiming-normal convs a=0.01, bias 0 for convs as
    nnU-Net's InitWeights_He does; IN gamma/beta perturbed so the affine path is exercised).

    `structured` > 0: the closest stand-in for a TRAINED net this image allows (no checkpoints offline).  Every spatial conv kernel is a
    random channel mixing times a smoothing stencil (binomial 1-2-1 per axis) plus `structured`-scaled Kaiming noise: the stack then
    computes smooth features of the input, the random 1x1x1 head turns them into class maps that are piecewise smooth -- most voxels
    have a clear winner (top-2 margin of several per cent of the logit range), near-ties only along the class boundaries, which is
    how a real segmentation's logits look and what the fp16 label-flip bound has to be measured on."""
    rng = np.random.default_rng(seed)
    sd = {}

    def conv(key, cout, cin, k, transposed=False):
        fan_in = (cout if transposed else cin) * int(np.prod(k))
        std = np.sqrt(2.0 / (1 + 0.01 ** 2)) / np.sqrt(fan_in)
        shape = (cin, cout, *k) if transposed else (cout, cin, *k)
        w = rng.standard_normal(shape) * std
        if structured > 0 and int(np.prod(k)) > 1:
            st = np.ones(1)
            for kk in k:     # separable binomial stencil (a transposed conv with kernel == stride gets the box)
                ax = {1: [1.0], 2: [1.0, 1.0], 3: [1.0, 2.0, 1.0]}.get(int(kk), [1.0] * int(kk))
                st = np.multiply.outer(st, np.asarray(ax))
            st = st.reshape([int(v) for v in k])
            st = st / np.sqrt((st ** 2).sum())
            mix = rng.standard_normal(shape[:2]) * std * np.sqrt(float(np.prod(k)))
            w = mix[..., None, None, None] * st + structured * w
        sd[key + ".weight"] = w.astype(np.float32)
        sd[key + ".bias"] = (rng.standard_normal(cout) * 0.02).astype(np.float32)

    def norm(key, c):
        sd[key + ".weight"] = (1.0 + 0.1 * rng.standard_normal(c)).astype(np.float32)
        sd[key + ".bias"] = (0.1 * rng.standard_normal(c)).astype(np.float32)

    cin = geom.in_channels
    for s in range(geom.n_stages):
        for i in range(geom.n_conv_enc[s]):
            p = f"encoder.stages.{s}.0.convs.{i}"
            conv(p + ".conv", geom.features[s], cin, geom.kernels[s])
            norm(p + ".norm", geom.features[s])
            cin = geom.features[s]
    for k in range(geom.n_stages - 1):
        sb = geom.n_stages - 1 - k
        below, skip = geom.features[sb], geom.features[sb - 1]
        conv(f"decoder.transpconvs.{k}", skip, below, geom.strides[sb], transposed=True)
        ci = 2 * skip
        for i in range(geom.n_conv_dec[k]):
            p = f"decoder.stages.{k}.convs.{i}"
            conv(p + ".conv", skip, ci, geom.kernels[sb - 1])
            norm(p + ".norm", skip)
            ci = skip
        conv(f"decoder.seg_layers.{k}", geom.num_classes, skip, [1, 1, 1])
    return sd
