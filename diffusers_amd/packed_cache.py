"""On-disk cache of a model's PACKED weights (the caller-side set-up around the hot path).

``load_state_dict`` turns a reference-format ``state_dict`` into the kernels' layouts (implicit-GEMM conv weights, fused
Q/K projections, interleaved GEGLU rows, zero-padded channels, folded biases ...).  That work is deterministic, so it
is done once: ``save_packed`` writes every packed tensor of a loaded model to one ``safetensors`` file, ``load_packed``
rebuilds the model from it without touching the original checkpoint or re-running the packing arithmetic.

How a model is rebuilt without per-class code: every engine class can run its ``load_state_dict`` on META tensors (shape
arithmetic only, nothing allocated), which yields the complete object skeleton; the packed tensors are then dropped into
the skeleton by walking both object graphs in the same deterministic order.  The reference counterpart of the
checkpoint side is ``ModelMixin.from_pretrained`` / ``save_pretrained`` (models/modeling_utils.py:886-1468, :629-884); a
LoRA-fused checkpoint is cached the same way after fusing it into the ``state_dict`` (loaders/peft.py ``fuse_lora``).
"""
from __future__ import annotations

import hashlib
import json
from collections import OrderedDict
from pathlib import Path
from typing import Callable, Dict, Tuple

import torch

from . import init as dinit

FORMAT = "diffusers_amd.packed/1"

# class name -> parameter inventory of its reference state_dict (what the meta skeleton is built from)
_SHAPES: Dict[str, Callable] = {
    "UNet2DConditionModel": dinit.unet_param_shapes,
    "AutoencoderKL": dinit.vae_decoder_param_shapes,
    "AutoencoderKLWan": dinit.wan_vae_decoder_param_shapes,
    "FluxTransformer2DModel": dinit.flux_param_shapes,
    "WanTransformer3DModel": dinit.wan_param_shapes,
    "UNet2DModel": dinit.unet2d_param_shapes,
}


def _walk(obj, path: str, visit, seen: set) -> None:
    """Depth-first over attributes / list items / dict values in insertion order; ``visit(container, key, path, tensor)``
    for every tensor leaf.  ``config`` objects and caches that are rebuilt lazily are skipped."""
    if isinstance(obj, (str, bytes, int, float, bool, type(None), torch.dtype, torch.device)):
        return
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, dict):
        items = [(k, v) for k, v in obj.items()]
        get = lambda k: obj[k]                      # noqa: E731
    elif isinstance(obj, (list, tuple)):
        items = list(enumerate(obj))
        get = lambda k: obj[k]                      # noqa: E731
    elif hasattr(obj, "__dict__"):
        items = [(k, v) for k, v in vars(obj).items() if k not in ("config", "_rope_cache", "_graph", "_static", "_cond_cache", "_source", "_lora")]
        get = lambda k: getattr(obj, k)             # noqa: E731
    else:
        return
    for k, v in items:
        p = f"{path}.{k}" if path else str(k)
        if isinstance(v, torch.Tensor):
            visit(obj, k, p, v)
        else:
            _walk(get(k), p, visit, seen)


def packed_tensors(model) -> "OrderedDict[str, torch.Tensor]":
    """Every tensor the loaded model holds, keyed by its attribute path (aliases of one tensor keep the first path)."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    first: Dict[int, str] = {}

    def visit(container, key, path, t):
        if id(t) in first:
            return
        first[id(t)] = path
        out[path] = t
    _walk(model, "", visit, set())
    return out


def _config_dict(model) -> dict:
    cfg = dict(model.config)
    return json.loads(json.dumps(cfg, default=lambda o: list(o) if isinstance(o, (tuple, set)) else str(o)))


def fingerprint(state_dict: Dict[str, torch.Tensor]) -> str:
    """Cheap identity of a checkpoint (names, shapes, dtypes and a strided sample of the bytes) for cache invalidation."""
    h = hashlib.sha256()
    for k in sorted(state_dict):
        t = state_dict[k]
        h.update(k.encode())
        h.update(str((tuple(t.shape), str(t.dtype))).encode())
        flat = t.detach().reshape(-1)
        step = max(1, flat.numel() // 64)
        h.update(flat[::step].to(torch.float32).cpu().numpy().tobytes())
    return h.hexdigest()


def save_packed(model, path, source_fingerprint: str = "") -> Path:
    """Write the packed tensors of a loaded engine model (any class of this package) to ``path`` (safetensors)."""
    from safetensors.torch import save_file
    name = type(model).__name__
    if name not in _SHAPES:
        raise TypeError(f"save_packed: {name} has no parameter inventory")
    tensors = {k: v.detach().contiguous().cpu() for k, v in packed_tensors(model).items()}
    meta = {"format": FORMAT, "class": name, "config": json.dumps(_config_dict(model)), "source": source_fingerprint}
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    save_file(tensors, str(path), metadata=meta)
    return path


def read_metadata(path) -> dict:
    from safetensors import safe_open
    with safe_open(str(path), framework="pt") as f:
        return dict(f.metadata() or {})


def load_packed(cls, path, device="cuda", expect_fingerprint: str = ""):
    """Rebuild a ``cls`` model from a ``save_packed`` file on ``device``.  Raises if the file was written for another
    class / format, if ``expect_fingerprint`` (see :func:`fingerprint`) does not match, or if the file's tensors do not
    fit the skeleton this version of the package builds (a stale cache must be re-packed, never half-used)."""
    from safetensors import safe_open
    meta = read_metadata(path)
    if meta.get("format") != FORMAT or meta.get("class") != cls.__name__:
        raise ValueError(f"load_packed: {path} holds {meta.get('class')!r} ({meta.get('format')!r}), not {cls.__name__}")
    if expect_fingerprint and meta.get("source") != expect_fingerprint:
        raise ValueError("load_packed: the cache was packed from a different checkpoint")
    cfg = {k: _tuples(v) for k, v in json.loads(meta["config"]).items()}
    model = cls(**cfg)
    shapes = _SHAPES[cls.__name__](dict(model.config))
    skeleton_sd = {k: torch.empty(tuple(s), dtype=torch.bfloat16, device="meta") for k, s in shapes.items()}
    model.load_state_dict(skeleton_sd, device="meta")
    dev = torch.device(device)
    slots: "OrderedDict[str, Tuple[object, object, torch.Tensor]]" = OrderedDict()
    alias: Dict[int, str] = {}
    aliased = []

    def visit(container, key, p, t):
        if id(t) in alias:
            aliased.append((container, key, alias[id(t)]))
            return
        alias[id(t)] = p
        slots[p] = (container, key, t)
    _walk(model, "", visit, set())
    loaded: Dict[str, torch.Tensor] = {}
    with safe_open(str(path), framework="pt", device=str(dev)) as f:
        keys = set(f.keys())
        if keys != set(slots):
            missing, extra = sorted(set(slots) - keys)[:4], sorted(keys - set(slots))[:4]
            raise ValueError(f"load_packed: stale cache (missing {missing}, unexpected {extra})")
        for p, (container, key, t) in slots.items():
            v = f.get_tensor(p)
            if tuple(v.shape) != tuple(t.shape) or v.dtype != t.dtype:
                raise ValueError(f"load_packed: stale cache ({p}: {tuple(v.shape)} {v.dtype} vs {tuple(t.shape)} {t.dtype})")
            loaded[p] = v
            _assign(container, key, v)
    for container, key, p in aliased:
        _assign(container, key, loaded[p])
    _retarget_devices(model, dev, set())
    return model


def _tuples(v):
    return tuple(_tuples(x) for x in v) if isinstance(v, list) else v


def _assign(container, key, value) -> None:
    if isinstance(container, dict):
        container[key] = value
    elif isinstance(container, list):
        container[key] = value
    elif isinstance(container, tuple):
        raise TypeError("packed_cache: tensors inside tuples cannot be re-assigned; keep them in lists")
    else:
        setattr(container, key, value)


def _retarget_devices(obj, dev: torch.device, seen: set) -> None:
    """``device`` attributes recorded while building the meta skeleton point at the real device afterwards."""
    if isinstance(obj, (str, bytes, int, float, bool, type(None), torch.Tensor, torch.dtype, torch.device)) or id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, dict):
        for v in obj.values():
            _retarget_devices(v, dev, seen)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _retarget_devices(v, dev, seen)
    elif hasattr(obj, "__dict__"):
        for k, v in list(vars(obj).items()):
            if isinstance(v, torch.device) and v.type == "meta":
                setattr(obj, k, dev)
            else:
                _retarget_devices(v, dev, seen)
