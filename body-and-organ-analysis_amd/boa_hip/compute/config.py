"""Model / device resolution with the reference's semantics (BOA/compute/config.py:13-69, pinned by the
reference's tests/test_config.py and by tests/golden/g10_config.json)."""
from __future__ import annotations

import logging
import os
from typing import Callable, Optional, Set

from .constants import ALL_MODELS, AVAILABLE_MODELS, LICENSE_MODELS

logger = logging.getLogger(__name__)


def env_bool(name: str, default: bool = False) -> bool:
    raw = os.getenv(name)
    return default if raw is None else raw.strip().lower() in {"1", "true"}


def env_str(name: str, default: Optional[str] = None) -> Optional[str]:
    raw = os.getenv(name)
    if raw is None or raw.strip().lower() in {"", "todo"}:
        return default
    return raw.strip()


def resolve_models(spec: Optional[str], strict: bool = False, license_number: Optional[str] = None,
                   is_valid_license: Optional[Callable[[str], bool]] = None) -> Set[str]:
    """`is_valid_license` stands in for totalsegmentator.config.is_valid_license (not part of the hot path)."""
    if not spec or spec.lower() == "all":
        models = set(ALL_MODELS)
        if license_number and is_valid_license is not None and is_valid_license(license_number):
            models |= LICENSE_MODELS
    else:
        models = {s.replace("-", "_") for s in spec.split("+")}
        invalid = models - AVAILABLE_MODELS
        if invalid:
            if strict:
                raise ValueError(f"Unknown model(s): {', '.join(sorted(invalid))}. "
                                 f"Available: {', '.join(sorted(AVAILABLE_MODELS))}")
            logger.error("Ignoring invalid model entries: %s. Available models are: %s.", invalid, sorted(AVAILABLE_MODELS))
            models -= invalid
    if "bca" in models:
        models = (models | {"total"}) - {"body_regions", "body_parts"}
    return models


def resolve_device(device: Optional[str] = None) -> str:
    device_str = device or os.environ.get("DEVICE", "gpu")
    device_str, _, gpu_id = device_str.partition(":")
    if device_str in ("cuda", "hip"):  # "hip" is this engine's alias; the reference accepts "cuda"
        device_str = "gpu"
    gpu_id = gpu_id or os.environ.get("NVIDIA_ID", "")
    if gpu_id and device_str == "gpu":
        os.environ.setdefault("NVIDIA_VISIBLE_DEVICES", gpu_id)
        device_str = f"gpu:{gpu_id}"
    return device_str
