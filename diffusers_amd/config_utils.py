"""Building engine objects from the reference's ``config`` mappings (``ConfigMixin.from_config``,
configuration_utils.py:179-260): the one-line way to put an engine model / scheduler where a reference one was.

    unet = from_reference_config(diffusers_amd.UNet2DConditionModel, pipe.unet.config)
    unet.load_state_dict(pipe.unet.state_dict(), device="cuda")
"""
from __future__ import annotations

from typing import Any, Mapping


def _tuples(v: Any) -> Any:
    return tuple(_tuples(x) for x in v) if isinstance(v, list) else v


def from_reference_config(cls, config: Mapping[str, Any], **overrides):
    """``cls(**config)`` with the reference's bookkeeping removed: private entries (``_class_name``, ``_diffusers_version``,
    ``_name_or_path`` ...) are dropped and JSON lists become tuples.  Everything else is passed on, so an option the engine
    class does not implement still raises (TypeError for an unknown key, ValueError / NotImplementedError for an unsupported
    value) instead of being ignored.  Schedulers additionally accept configs of OTHER scheduler classes through their own
    ``from_config`` (which drops foreign keys, as the reference does)."""
    src = {k: _tuples(v) for k, v in dict(config).items() if not str(k).startswith("_")}
    src.update(overrides)
    if hasattr(cls, "_defaults") and hasattr(cls, "from_config"):      # scheduler classes
        return cls.from_config(src)
    return cls(**src)
