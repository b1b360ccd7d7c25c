"""Oracle: the segmentation network, torch-CPU fp32 (TEST INFRASTRUCTURE, see oracle/__init__.py).

PARITY UNPINNED: the reference builds `dynamic_network_architectures==0.4.3`
`architectures.unet.PlainConvUNet` (uv.lock:806-807), a third-party package that is neither vendored in
/root/reference nor installed here.  This file restates its published structure from torch.nn,
following the architecture kwargs the reference itself synthesises
(NN/utilities/plans_handling/plans_handler.py:59-92: conv_bias=True, InstanceNorm eps=1e-5 affine,
LeakyReLU(inplace) default slope 0.01, per-stage kernel/stride lists) and the call sites
NN/utilities/get_network_from_plans.py:34-38 and NN/inference/predict_from_raw_data.py:104-111,543
(deep supervision disabled at inference).  State-dict keys follow upstream naming:
  encoder.stages.{s}.0.convs.{i}.{conv,norm}.{weight,bias}
  decoder.transpconvs.{s}.{weight,bias}; decoder.stages.{s}.convs.{i}.{conv,norm}.*;
  decoder.seg_layers.{s}.{weight,bias}
(upstream additionally stores `...all_modules.N.*` and `decoder.encoder.*` aliases of the same tensors).
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn


class ConvNormAct(nn.Module):
    def __init__(self, cin, cout, kernel, stride):
        super().__init__()
        pad = [(k - 1) // 2 for k in kernel]
        self.conv = nn.Conv3d(cin, cout, kernel, stride, pad, bias=True)
        self.norm = nn.InstanceNorm3d(cout, eps=1e-5, affine=True)
        self.nonlin = nn.LeakyReLU(0.01)

    def forward(self, x):
        return self.nonlin(self.norm(self.conv(x)))


class StackedConvBlocks(nn.Module):
    def __init__(self, n, cin, cout, kernel, first_stride):
        super().__init__()
        blocks = [ConvNormAct(cin, cout, kernel, first_stride)]
        blocks += [ConvNormAct(cout, cout, kernel, [1, 1, 1]) for _ in range(n - 1)]
        self.convs = nn.Sequential(*blocks)

    def forward(self, x):
        return self.convs(x)


class Encoder(nn.Module):
    def __init__(self, cin, features, kernels, strides, n_conv):
        super().__init__()
        stages = []
        for s, f in enumerate(features):
            stages.append(nn.Sequential(StackedConvBlocks(n_conv[s], cin, f, kernels[s], strides[s])))
            cin = f
        self.stages = nn.Sequential(*stages)

    def forward(self, x):
        skips = []
        for st in self.stages:
            x = st(x)
            skips.append(x)
        return skips


class Decoder(nn.Module):
    def __init__(self, features, kernels, strides, n_conv_dec, num_classes):
        super().__init__()
        n = len(features)
        stages, ups, segs = [], [], []
        for s in range(1, n):
            below, skip = features[-s], features[-(s + 1)]
            st = strides[-s]
            ups.append(nn.ConvTranspose3d(below, skip, st, st, bias=True))
            stages.append(StackedConvBlocks(n_conv_dec[s - 1], 2 * skip, skip, kernels[-(s + 1)], [1, 1, 1]))
            segs.append(nn.Conv3d(skip, num_classes, 1, 1, 0, bias=True))
        self.stages = nn.ModuleList(stages)
        self.transpconvs = nn.ModuleList(ups)
        self.seg_layers = nn.ModuleList(segs)

    def forward(self, skips):
        x = skips[-1]
        for s in range(len(self.stages)):
            u = self.transpconvs[s](x)
            x = torch.cat((u, skips[-(s + 2)]), 1)
            x = self.stages[s](x)
        return self.seg_layers[-1](x)


class PlainConvUNetOracle(nn.Module):
    def __init__(self, input_channels, num_classes, features_per_stage, kernel_sizes, strides,
                 n_conv_per_stage, n_conv_per_stage_decoder):
        super().__init__()
        n = len(features_per_stage)
        if isinstance(n_conv_per_stage, int):
            n_conv_per_stage = [n_conv_per_stage] * n
        if isinstance(n_conv_per_stage_decoder, int):
            n_conv_per_stage_decoder = [n_conv_per_stage_decoder] * (n - 1)
        self.encoder = Encoder(input_channels, features_per_stage, kernel_sizes, strides, n_conv_per_stage)
        self.decoder = Decoder(features_per_stage, kernel_sizes, strides, n_conv_per_stage_decoder, num_classes)

    def forward(self, x):
        return self.decoder(self.encoder(x))


def build_from_arch(arch: dict, input_channels: int, num_classes: int) -> PlainConvUNetOracle:
    """arch = plans["configurations"][cfg]["architecture"]["arch_kwargs"] (new-format plans)."""
    return PlainConvUNetOracle(
        input_channels, num_classes, arch["features_per_stage"], arch["kernel_sizes"], arch["strides"],
        arch["n_conv_per_stage"], arch["n_conv_per_stage_decoder"])


def network_fn_from_module(net: nn.Module, threads: int | None = None):
    """Wrap a torch module as the float32 ndarray -> ndarray callable the sliding-window oracle wants."""
    net = net.eval()

    def fn(patch: np.ndarray) -> np.ndarray:
        if threads:
            torch.set_num_threads(threads)
        with torch.inference_mode():
            return net(torch.from_numpy(np.ascontiguousarray(patch, dtype=np.float32))).numpy()

    return fn
