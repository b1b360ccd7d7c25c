"""Which models to run and on which device: the flag / environment conventions of the reference's `compute/config.py`
(behaviour pinned by the reference's tests/test_config.py, restated here as tables in tests/golden/g10_config.json).
Only the conventions are shared -- the implementation is table driven."""
from __future__ import annotations

import logging
import os
import re
from typing import Callable, Optional, Set

from . import constants as K

logger = logging.getLogger(__name__)

_TRUE_WORDS = frozenset(("1", "true"))
_UNSET_WORDS = frozenset(("", "todo"))                  # placeholder values of the shipped .env template count as "not set"
_DEVICE_RE = re.compile(r"^(?P<kind>[^:]*)(?::(?P<index>.*))?$")
_GPU_ALIASES = frozenset(("cuda", "hip"))               # "hip" is this engine's own alias


def _env(name: str) -> Optional[str]:
    value = os.getenv(name)
    return None if value is None else value.strip()


def env_bool(name: str, default: bool = False) -> bool:
    value = _env(name)
    if value is None:
        return default
    return value.lower() in _TRUE_WORDS


def env_str(name: str, default: Optional[str] = None) -> Optional[str]:
    value = _env(name)
    if value is None or value.lower() in _UNSET_WORDS:
        return default
    return value


def _all_models(license_number, is_valid_license) -> Set[str]:
    licensed = bool(license_number) and is_valid_license is not None and bool(is_valid_license(license_number))
    return set(K.ALL_MODELS) | (set(K.LICENSE_MODELS) if licensed else set())


def resolve_models(spec: Optional[str], strict: bool = False, license_number: Optional[str] = None,
                   is_valid_license: Optional[Callable[[str], bool]] = None) -> Set[str]:
    """'a+b+c' (dashes and underscores interchangeable), 'all' or nothing -> set of model names.  Unknown names raise
    when `strict`, else they are logged and dropped.  `bca` implies `total` and subsumes its two networks.
    (`is_valid_license` stands in for TotalSegmentator's licence check, which is not part of the hot path.)"""
    wanted = (spec or "all")
    if wanted.lower() == "all":
        chosen = _all_models(license_number, is_valid_license)
    else:
        tokens = [tok.replace("-", "_") for tok in wanted.split("+")]
        unknown = sorted(set(tokens) - set(K.AVAILABLE_MODELS))
        if unknown and strict:
            raise ValueError(f"Unknown model(s): {', '.join(unknown)}. Available: {', '.join(sorted(K.AVAILABLE_MODELS))}")
        if unknown:
            logger.error("Ignoring invalid model entries: %s. Available models are: %s.", set(unknown), sorted(K.AVAILABLE_MODELS))
        chosen = {tok for tok in tokens if tok in K.AVAILABLE_MODELS}
    if "bca" in chosen:
        chosen.add("total")
        chosen.difference_update(("body_regions", "body_parts"))
    return chosen


def resolve_device(device: Optional[str] = None) -> str:
    """'gpu' | 'gpu:<id>' | 'cpu' ... from the argument or $DEVICE; 'cuda' / 'hip' mean 'gpu'; the index may also come from
    $NVIDIA_ID and is exported as $NVIDIA_VISIBLE_DEVICES (kept for compatibility with the reference's deployments)."""
    m = _DEVICE_RE.match(device or os.environ.get("DEVICE", "gpu"))
    kind = m.group("kind")
    index = m.group("index") or os.environ.get("NVIDIA_ID", "")
    if kind in _GPU_ALIASES:
        kind = "gpu"
    if kind == "gpu" and index:
        os.environ.setdefault("NVIDIA_VISIBLE_DEVICES", index)
        return f"gpu:{index}"
    return kind
