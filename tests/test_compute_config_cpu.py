"""CPU: the compute/ mirror reproduces the reference's model/device resolution (golden G10 + the cases of the
reference's own tests/test_config.py)."""
import json
import os
from unittest import mock

import pytest

from conftest import GOLDEN
from boa_hip.compute import config, constants, util


def _g():
    with open(os.path.join(GOLDEN, "g10_config.json")) as f:
        return json.load(f)


def test_constants():
    g = _g()
    assert sorted(constants.ALL_MODELS) == g["ALL_MODELS"]
    assert sorted(constants.BASE_MODELS) == g["BASE_MODELS"]
    assert sorted(constants.LICENSE_MODELS) == g["LICENSE_MODELS"]
    assert sorted(constants.AVAILABLE_MODELS) == g["AVAILABLE_MODELS"]


def test_resolve_models_table():
    for spec, want in _g()["resolve_models"].items():
        s = None if spec == "None" else spec
        assert sorted(config.resolve_models(s)) == want, spec


def test_resolve_models_reference_cases():
    all_resolved = set(constants.ALL_MODELS) - {"body_parts", "body_regions"}
    assert config.resolve_models(None) == all_resolved
    assert config.resolve_models("bca") == {"bca", "total"}
    assert config.resolve_models("body-parts") == {"body_parts"}
    assert config.resolve_models("body+total") == {"total"}
    with pytest.raises(ValueError):
        config.resolve_models("body+total", strict=True)
    assert config.resolve_models("all", license_number="1" * 18, is_valid_license=lambda n: True) == \
        all_resolved | constants.LICENSE_MODELS
    assert config.resolve_models("all", license_number="1" * 18, is_valid_license=lambda n: False) == all_resolved


def test_resolve_device_table():
    for dev, want in _g()["resolve_device"].items():
        with mock.patch.dict(os.environ, {}, clear=False):
            for k in ("DEVICE", "NVIDIA_ID", "NVIDIA_VISIBLE_DEVICES"):
                os.environ.pop(k, None)
            assert config.resolve_device(None if dev == "None" else dev) == want, dev


def test_env_helpers_and_slices():
    with mock.patch.dict(os.environ, {"A": " True ", "B": "todo", "C": " x "}):
        assert config.env_bool("A") and not config.env_bool("ZZZ") and config.env_bool("ZZZ", True)
        assert config.env_str("B", "d") == "d" and config.env_str("C") == "x"
    for s, c, t, want in _g()["convert_resampling_slices"]:
        assert util.convert_resampling_slices(s, c, t) == want


def test_cascade_crop_margin_is_the_references_effective_value():
    """TS/python_api.py:726: `crop_addon = [20,20,20] if crop_model is None else crop_addon`.  None of the five cascade
    tasks BOA runs names a crop_model (python_api.py:236-259,311-329; only craniofacial tasks do, :460), so the margin
    that reaches nnUNet_predict_image is 20 mm for all of them -- not the per-task table value."""
    from boa_hip import model_store
    for t in model_store.CASCADE_MODELS:
        assert model_store.TASKS[t].get("crop_model") is None
        assert model_store.effective_crop_addon(t) == [20, 20, 20], t
    # the table itself keeps the reference's per-task values (what a task WITH a crop_model would use)
    assert model_store.TASKS["pleural_pericard_effusion"]["crop_addon"] == [50, 50, 50]
    with pytest.raises(KeyError):
        model_store.effective_crop_addon("total")


def test_create_mask_and_int16_guard():
    import numpy as np
    a = np.array([[0, 3, 7], [7, 200, 3]], dtype=np.uint8)
    assert (util.create_mask(a, 7) == (a == 7)).all()
    assert (util.create_mask(a, (v for v in [3, 200])) == np.isin(a, [3, 200])).all()     # generators are fine
    assert (util.create_mask(a.astype(np.float64) * np.array([1, np.nan, 1]), [3]) == np.array([[0, 0, 0], [0, 0, 1]], bool)).all()
    assert util.require_int16_exact(np.array([-1024.0, 3071.0])).dtype == np.int16
    for bad in (np.array([0.5]), np.array([40000.0]), np.array([70000], dtype=np.int32)):
        with pytest.raises(ValueError):
            util.require_int16_exact(bad)
