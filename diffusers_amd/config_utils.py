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


def check_to(model, args, kwargs):
    """``nn.Module.to`` of the engine's model mirrors.  The packed weights live on ONE HIP device in bf16 (set by
    ``load_state_dict(..., device=)``); a request for that placement is a no-op, anything else -- another dtype, the CPU,
    another device index -- cannot be honoured and raises instead of being silently ignored.  Before ``load_state_dict`` there is
    nothing to move: only the dtype is checked."""
    import torch

    want_dtype, want_dev = kwargs.pop("dtype", None), kwargs.pop("device", None)
    kwargs.pop("non_blocking", None), kwargs.pop("copy", None), kwargs.pop("memory_format", None)
    if kwargs:
        raise TypeError(f"{type(model).__name__}.to(): unexpected arguments {sorted(kwargs)}")
    for a in args:
        if isinstance(a, torch.dtype):
            want_dtype = a
        elif isinstance(a, (str, torch.device, int)):
            want_dev = a
        elif torch.is_tensor(a):
            want_dtype, want_dev = a.dtype, a.device
        elif a is not None and not isinstance(a, bool):
            raise TypeError(f"{type(model).__name__}.to(): cannot interpret argument {a!r}")
    if want_dtype is not None and want_dtype != torch.bfloat16:
        raise ValueError(f"{type(model).__name__}.to({want_dtype}): the HIP engine computes in bfloat16 only "
                         "(no fp16 / fp32 path); keep the reference model for other dtypes")
    if want_dev is not None:
        dev = torch.device("cuda", want_dev) if isinstance(want_dev, int) else torch.device(want_dev)
        if dev.type != "cuda":
            raise ValueError(f"{type(model).__name__}.to({dev}): the HIP engine has no CPU path")
        have = getattr(model, "device", None)
        if have is not None and dev.index is not None and torch.device(have).index not in (None, dev.index):
            raise ValueError(f"{type(model).__name__}.to({dev}): the packed weights live on {have}; re-pack them with "
                             f"load_state_dict(state_dict, device='{dev}')")
    return model
