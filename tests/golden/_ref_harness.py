"""Stub-import harness used ONLY by make_golden.py in the dev container.

It puts the read-only reference checkout (/root/reference) on sys.path and serves permissive dummy
modules for the third-party packages the reference imports but this image lacks, so that the
reference's own hot-path functions can be executed on CPU to produce golden vectors.
Nothing here ships to the GPU box as a dependency of any test: tests read the committed *.npz / *.json
fixtures only.  (SURVEY.md Appendix A documents the recipe.)
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import sys
import tempfile
import types

REF = "/root/reference"
EXT = os.path.join(REF, "body_organ_analysis", "_external")

_STUB_ROOTS = {
    "acvl_utils", "batchgenerators", "batchgeneratorsv2", "dynamic_network_architectures", "blosc2",
    "SimpleITK", "nibabel", "tifffile", "skimage", "cv2", "dotenv", "boa_contrast", "pydicom",
    "weasyprint", "xlsxwriter", "p_tqdm", "fury", "xvfbwrapper", "seaborn", "unidecode",
    "dataclasses_json", "plotly", "kaleido", "requests_toolbelt", "dicom2nifti", "nnunet",
}


class _DummyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Dummy

    def __call__(cls, *a, **k):
        return type.__call__(cls)

    def __iter__(cls):
        return iter(())

    def __or__(cls, other):
        return cls

    def __ror__(cls, other):
        return cls


class _Dummy(metaclass=_DummyMeta):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Dummy()

    def __call__(self, *a, **k):
        return _Dummy()

    def __iter__(self):
        return iter(())

    def __getitem__(self, k):
        return _Dummy()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Dummy


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _pad_nd_image(image, new_shape=None, mode="constant", kwargs=None, return_slicer=False,
                  shape_must_be_divisible_by=None):
    """acvl_utils 0.2.5 pad_nd_image semantics (symmetric pad; below=d//2, above=d//2+d%2)."""
    import numpy as np
    import torch
    if kwargs is None:
        kwargs = {}
    old_shape = np.array(image.shape)
    n = len(new_shape)
    new_shape = list(old_shape[:-n]) + list(new_shape)
    new_shape = np.array([max(a, b) for a, b in zip(new_shape, old_shape)])
    diff = new_shape - old_shape
    below = diff // 2
    above = diff // 2 + diff % 2
    pad_list = [[int(b), int(a)] for b, a in zip(below, above)]
    if any(p != [0, 0] for p in pad_list):
        if isinstance(image, np.ndarray):
            res = np.pad(image, pad_list, mode, constant_values=kwargs.get("value", 0))
        else:
            tp = [i for j in pad_list for i in j[::-1]][::-1]
            res = torch.nn.functional.pad(image, tp, mode, **kwargs)
    else:
        res = image
    if not return_slicer:
        return res
    pad_arr = np.array(pad_list)
    pad_arr[:, 1] = np.array(res.shape) - pad_arr[:, 1]
    slicer = tuple(slice(*i) for i in pad_arr)
    return res, slicer


_installed = False


def install():
    global _installed
    if _installed:
        return
    _installed = True
    os.environ.setdefault("HOME", tempfile.mkdtemp(prefix="boa_ref_home_"))
    os.environ["HOME"] = tempfile.mkdtemp(prefix="boa_ref_home_")
    sys.path[:0] = [EXT, REF]
    sys.meta_path.insert(0, _StubFinder())
    import numpy as np
    if not hasattr(np, "float_"):
        np.float_ = np.float64
    import importlib
    pad_mod = importlib.import_module("acvl_utils.cropping_and_padding.padding")
    pad_mod.pad_nd_image = _pad_nd_image
    fo = importlib.import_module("batchgenerators.utilities.file_and_folder_operations")
    from typing import List
    fo.join = os.path.join
    fo.List = List
    fo.os = os
    fo.__all__ = ["join", "List", "os"]


class FakeSitkImage:
    """Enough of sitk.Image for subclassify_tissues / Builder: array is (z, y, x)."""

    def __init__(self, arr, spacing=(1.0, 1.0, 1.0)):
        self.arr = arr
        self.spacing = tuple(float(s) for s in spacing)  # (x, y, z) like sitk

    def GetSpacing(self):
        return self.spacing

    def GetDepth(self):
        return self.arr.shape[0]

    def CopyInformation(self, other):
        self.spacing = other.spacing


class FakeSitk(types.ModuleType):
    Image = FakeSitkImage

    @staticmethod
    def GetArrayFromImage(img):
        return img.arr.copy()

    @staticmethod
    def GetArrayViewFromImage(img):
        return img.arr

    @staticmethod
    def GetImageFromArray(arr):
        return FakeSitkImage(arr)

    @staticmethod
    def WriteImage(*a, **k):
        return None
