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
