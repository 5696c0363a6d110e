"""Checkpoint-side set-up of the engine models: ``from_pretrained`` over the reference's on-disk layout,
pack-once caching, and LoRA adapters fused into the packed weights.

Reference counterparts: ``ModelMixin.from_pretrained`` / ``save_pretrained`` (models/modeling_utils.py:886-1468, :629-884:
``config.json`` + ``diffusion_pytorch_model[.variant].safetensors`` or a sharded ``*.safetensors.index.json`` in a model
directory or a pipeline sub-folder), ``PeftAdapterMixin.load_lora_adapter`` / ``fuse_lora`` (loaders/peft.py:79-330, :646-700)
and the kohya -> diffusers key conversion of loaders/lora_conversion_utils.py:184-232.

What is different here, by design:
  * a checkpoint is PACKED, not just loaded (implicit-GEMM conv layouts, fused Q|K rows, GEGLU interleave, LayerNorm folds ...).
    ``from_pretrained`` therefore keeps a pack-once cache next to the checkpoint (``packed_cache.save_packed``), keyed by a
    fingerprint of the checkpoint files; the second start of a 2.6 B-parameter U-Net reads the packed file and runs no packing
    arithmetic at all;
  * tensors are read lazily, one at a time, straight from the safetensors files (``LazyCheckpoint``): the reference-format
    weights are never resident as a whole next to the packed ones;
  * a LoRA adapter is FUSED (the engine has no unfused adapter path: its kernels read packed weights): ``fuse_lora`` re-packs
    the model from the lazy view ``W + scale * (alpha / r) * B A`` and writes the result INTO the existing packed tensors, so
    device addresses -- and with them any captured HIP graph -- stay valid.  ``unfuse_lora`` re-packs from the base view.
"""
from __future__ import annotations

import hashlib
import json
import os
import warnings
from collections.abc import Mapping
from pathlib import Path
from typing import Dict, Iterator, List, Optional

import torch

from . import packed_cache

CONFIG_NAME = "config.json"
WEIGHTS_NAME = "diffusion_pytorch_model"
PACKED_DIR = "diffusers_amd_packed"


# ----------------------------------------------------------------------------------------------------------------------
# lazy, read-only views of checkpoints
# ----------------------------------------------------------------------------------------------------------------------
class LazyCheckpoint(Mapping):
    """Read-only mapping name -> tensor over one or more ``.safetensors`` files; a tensor is read when it is asked for and
    not kept.  (The engine's ``load_state_dict`` asks for every tensor exactly once, while packing it.)"""

    def __init__(self, files: List[Path], device: str = "cpu"):
        from safetensors import safe_open
        self._files = [Path(f) for f in files]
        self._device = device
        self._where: Dict[str, Path] = {}
        for f in self._files:
            with safe_open(str(f), framework="pt") as h:
                for k in h.keys():
                    if k in self._where:
                        raise ValueError(f"tensor {k!r} appears in both {self._where[k].name} and {f.name}")
                    self._where[k] = f

    def __getitem__(self, key: str) -> torch.Tensor:
        from safetensors import safe_open
        f = self._where[key]
        with safe_open(str(f), framework="pt", device=self._device) as h:
            return h.get_tensor(key)

    def __iter__(self) -> Iterator[str]:
        return iter(self._where)

    def __len__(self) -> int:
        return len(self._where)

    def __contains__(self, key) -> bool:
        return key in self._where


COMPONENT_PREFIXES = ("unet.", "transformer.", "text_encoder.", "text_encoder_2.", "text_encoder_3.")
_LORA_TAGS = ((".lora_A.weight", ".lora_B.weight"), (".lora_A.default.weight", ".lora_B.default.weight"),
              (".lora.down.weight", ".lora.up.weight"), ("_lora.down.weight", "_lora.up.weight"),
              (".lora_down.weight", ".lora_up.weight"))


def _lora_pairs(lora_sd: Mapping, prefixes=("unet.", "transformer."), strict: bool = False) -> Dict[str, dict]:
    """module path -> {"A": [r][in...], "B": [out][r...], "alpha": float | None} from a LoRA state dict in PEFT
    (``lora_A.weight`` / ``lora_B.weight``), legacy diffusers (``lora.down.weight`` / ``lora.up.weight``, attention-processor
    ``to_q_lora.down.weight``) or kohya-after-conversion naming.

    Component filter, as ``load_lora_adapter`` does it (loaders/peft.py:194-195): when the dict carries component prefixes
    (a pipeline-level file: ``unet.`` / ``transformer.`` / ``text_encoder.`` ...), only the keys under one of ``prefixes``
    (this model's) are kept, with the prefix removed; the other components' keys are ignored, not an error.  A dict with no
    component prefix at all is taken to be this model's own.  A prefixed dict with nothing for this model warns and returns an
    empty mapping like the reference (``strict=True``: raises)."""
    out: Dict[str, dict] = {}
    keys = list(lora_sd)
    prefixed = any(k.startswith(COMPONENT_PREFIXES) for k in keys)

    def strip(k: str):
        if not prefixed:
            return k
        for p in prefixes:
            if k.startswith(p):
                return k[len(p):]
        return None            # another component's weight

    for k in keys:
        name = strip(k)
        if name is None:
            continue
        base = which = None
        for a_tag, b_tag in _LORA_TAGS:
            if name.endswith(a_tag):
                base, which, legacy = name[: -len(a_tag)], "A", a_tag.startswith("_lora")
            elif name.endswith(b_tag):
                base, which, legacy = name[: -len(b_tag)], "B", b_tag.startswith("_lora")
            else:
                continue
            break
        if base is None:
            if name.endswith(".alpha"):
                out.setdefault(name[: -len(".alpha")], {})["alpha"] = float(torch.as_tensor(lora_sd[k]).item())
            continue
        if legacy:
            # LoRAAttnProcessor naming: `<attn>.processor.to_q_lora.down.weight` -> `<attn>.to_q`; its output projection is
            # `to_out_lora`, which lives at `<attn>.to_out.0` in the model
            base = base.replace(".processor.", ".")
            if base.endswith(".to_out"):
                base += ".0"
        out.setdefault(base, {})[which] = lora_sd[k]
    bad = [m for m, v in out.items() if "A" not in v or "B" not in v]
    if bad:
        raise ValueError(f"LoRA state dict has unpaired matrices for {bad[:4]}")
    if prefixed and not out:
        # e.g. a text-encoder-only file handed to the pipeline-level loader: the reference logs and moves on
        # (loaders/peft.py:359-366), so that loading such a file succeeds and simply leaves this model untouched
        msg = (f"No LoRA keys found under {prefixes}; the file's components are "
               f"{sorted({k.split('.', 1)[0] for k in keys})}")
        if strict:
            raise ValueError(msg)
        warnings.warn(msg + " -- this model is left unchanged", stacklevel=2)
    return out


class LoraFusedView(Mapping):
    """``base`` with ``W + scale * (alpha / r) * B @ A`` wherever ``lora`` has a pair for ``<module>.weight`` (Linear:
    [out][r] @ [r][in]; Conv2d: B [out][r][1][1] x A [r][in][k][k], loaders/peft.py + peft's LoRA merge).  The sum is
    formed in fp32 and returned in the base tensor's dtype; everything else passes through."""

    def __init__(self, base: Mapping, lora_sd: Mapping, scale: float = 1.0):
        self.base, self.scale = base, float(scale)
        self.pairs = _lora_pairs(lora_sd)
        unknown = [m for m in self.pairs if m + ".weight" not in base]
        if unknown:
            raise KeyError(f"LoRA targets modules the checkpoint does not have: {unknown[:4]} (of {len(unknown)})")

    def __getitem__(self, key: str) -> torch.Tensor:
        w = self.base[key]
        if not key.endswith(".weight"):
            return w
        ent = self.pairs.get(key[: -len(".weight")])
        if ent is None:
            return w
        a, b = ent["A"].to(torch.float32), ent["B"].to(torch.float32)
        r = a.shape[0]
        s = self.scale * ((ent["alpha"] / r) if ent.get("alpha") is not None else 1.0)
        if w.dim() == 4:
            delta = (b.reshape(b.shape[0], r) @ a.reshape(r, -1)).reshape(w.shape)
        else:
            delta = b @ a
        if tuple(delta.shape) != tuple(w.shape):
            raise ValueError(f"LoRA delta {tuple(delta.shape)} does not fit {key} {tuple(w.shape)}")
        return (w.to(torch.float32) + s * delta.to(w.device)).to(w.dtype)

    def __iter__(self):
        return iter(self.base)

    def __len__(self):
        return len(self.base)

    def __contains__(self, key):
        return key in self.base


# ----------------------------------------------------------------------------------------------------------------------
# locating a checkpoint
# ----------------------------------------------------------------------------------------------------------------------
def _resolve_dir(name_or_path, subfolder: Optional[str]) -> Path:
    p = Path(name_or_path)
    if not p.exists():
        # the reference falls back to the Hub here (modeling_utils.py:1006-1040); offline this raises a clear error
        try:
            from huggingface_hub import snapshot_download
            p = Path(snapshot_download(str(name_or_path), allow_patterns=["*.json", "*.safetensors"]))
        except Exception as e:  # noqa: BLE001  (any hub / network failure)
            raise OSError(f"{name_or_path!r} is not a local directory and could not be fetched from the Hub: {e}") from e
    if subfolder:
        p = p / subfolder
    if not (p / CONFIG_NAME).exists():
        raise OSError(f"{p} has no {CONFIG_NAME}")
    return p


def _weight_files(d: Path, variant: Optional[str], stem: str = WEIGHTS_NAME) -> List[Path]:
    """The safetensors file(s) of a component directory (``stem`` = "diffusion_pytorch_model" for diffusers models, "model" for
    transformers ones)."""
    base = stem
    stem = base + (f".{variant}" if variant else "")
    single = d / f"{stem}.safetensors"
    if single.exists():
        return [single]
    index = d / f"{stem}.safetensors.index.json"
    if not index.exists():       # the reference writes the variant before ".index.json" for sharded checkpoints
        index = d / f"{base}.safetensors.index{'.' + variant if variant else ''}.json"
    if index.exists():
        files = sorted(set(json.loads(index.read_text())["weight_map"].values()))
        return [d / f for f in files]
    if (d / f"{stem}.bin").exists():
        raise OSError(f"{d} holds a pickle checkpoint ({stem}.bin); convert it to safetensors (the engine does not unpickle)")
    raise OSError(f"no {stem}.safetensors (or sharded index) in {d}")


def checkpoint_fingerprint(files: List[Path], extra: str = "") -> str:
    """Identity of a checkpoint without reading all of it: names, sizes, modification times, the safetensors headers and
    64 strided 4 KiB samples of each file's data."""
    h = hashlib.sha256(extra.encode())
    for f in sorted(files):
        st = f.stat()
        h.update(f.name.encode())
        h.update(str((st.st_size, st.st_mtime_ns)).encode())
        with open(f, "rb") as fh:
            n = int.from_bytes(fh.read(8), "little")
            h.update(fh.read(min(n, 1 << 20)))
            step = max(4096, (st.st_size // 64) & ~4095)
            for off in range(0, st.st_size, step):
                fh.seek(off)
                h.update(fh.read(4096))
    return h.hexdigest()[:24]


def read_config(d: Path) -> dict:
    cfg = json.loads((d / CONFIG_NAME).read_text())
    return {k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items() if not k.startswith("_")}


# ----------------------------------------------------------------------------------------------------------------------
# the mixin
# ----------------------------------------------------------------------------------------------------------------------
class PretrainedMixin:
    """``from_pretrained`` / ``fuse_lora`` / ``unfuse_lora`` for the engine's model classes."""

    _source: Optional[dict] = None     # {"files": [...], "device": ...}: where this model's reference-format weights live

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder: Optional[str] = None, torch_dtype=None,
                        variant: Optional[str] = None, device="cuda", use_safetensors: Optional[bool] = None,
                        cache_packed: bool = True, cache_dir: Optional[os.PathLike] = None, **config_overrides):
        """Build the model from a reference checkpoint directory (``config.json`` + safetensors), packing its weights for
        the kernels on ``device``.  With ``cache_packed`` the packed tensors are written once to
        ``<checkpoint>/diffusers_amd_packed/`` (or ``cache_dir``) and reused while the checkpoint files are unchanged.
        ``torch_dtype`` other than bfloat16 and ``use_safetensors=False`` are refused (no fp16 / fp32 path, no pickle)."""
        if torch_dtype not in (None, torch.bfloat16):
            raise ValueError(f"{cls.__name__}.from_pretrained: torch_dtype={torch_dtype} -- the HIP engine computes in bfloat16")
        if use_safetensors is False:
            raise ValueError(f"{cls.__name__}.from_pretrained: pickle checkpoints are not read; use safetensors")
        from .config_utils import from_reference_config
        d = _resolve_dir(pretrained_model_name_or_path, subfolder)
        files = _weight_files(d, variant)
        cfg = read_config(d)
        cfg.update(config_overrides)
        from . import ops
        fp = checkpoint_fingerprint(files, extra=cls.__name__ + json.dumps(cfg, sort_keys=True, default=str)
                                    + (f"+lnfold{int(ops.LN_FOLD)}qkv" if ops.LN_FOLD else "") + "+tembcat")   # the fold / the stacked time projections change the packed inventory
        cdir = Path(cache_dir) if cache_dir is not None else d / PACKED_DIR
        cfile = cdir / f"{cls.__name__}-{fp}.safetensors"
        model = None
        if cache_packed and cfile.exists():
            try:
                model = packed_cache.load_packed(cls, cfile, device=device, expect_fingerprint=fp)
            except ValueError:          # written by another version of the package: re-pack below
                model = None
        if model is None:
            model = from_reference_config(cls, cfg)
            model.load_state_dict(LazyCheckpoint(files), device=device)
            if cache_packed:
                try:
                    packed_cache.save_packed(model, cfile, source_fingerprint=fp)
                except OSError:         # read-only checkpoint directory: run without the cache
                    pass
        model._source = {"files": [str(f) for f in files]}
        model._lora = None
        return model

    def _base_view(self) -> Mapping:
        if not self._source:
            raise RuntimeError(f"{type(self).__name__}: built from an in-memory state_dict; pass `base_state_dict=` "
                               "(the reference-format weights are not kept after packing)")
        return LazyCheckpoint([Path(f) for f in self._source["files"]])

    def _repack_in_place(self, view: Mapping) -> None:
        """Pack ``view`` into a fresh CPU skeleton of this model and copy every packed tensor over the existing one."""
        fresh = type(self)(**{k: v for k, v in dict(self.config).items()})
        fresh.load_state_dict(view, device="cpu")
        old, new = packed_cache.packed_tensors(self), packed_cache.packed_tensors(fresh)
        if list(old) != list(new):
            raise RuntimeError("re-pack produced a different tensor inventory")
        for k, t in old.items():
            if tuple(t.shape) != tuple(new[k].shape):
                raise RuntimeError(f"re-pack changed the shape of {k}")
            t.copy_(new[k])
        # derived device copies that captured graphs / plans hold by raw pointer follow the edit at once (ops.pad_thin_out: the
        # 16-channel padded conv_out weight) -- a replayed graph never passes through the Python call that would refresh them
        from . import ops
        ops.refresh_thin_out(old.values())
        if hasattr(self, "_cond_cache"):
            self._cond_cache = None

    def fuse_lora(self, lora_state_dict: Mapping, lora_scale: float = 1.0, base_state_dict: Optional[Mapping] = None):
        """Fuse a LoRA adapter (PEFT / legacy diffusers / converted kohya naming) into the packed weights IN PLACE: the
        model is re-packed from ``W + lora_scale * (alpha / r) * B A`` and the result copied over the tensors the kernels
        (and any captured HIP graph) already point at.  Adapters do not stack: fusing replaces a previously fused one."""
        base = base_state_dict if base_state_dict is not None else self._base_view()
        view = LoraFusedView(base, lora_state_dict, lora_scale)
        if not view.pairs:
            return self            # a file for other components only (warned about above): nothing to fuse, as in the reference
        self._repack_in_place(view)
        self._lora = {"scale": float(lora_scale), "modules": len(view.pairs)}
        return self

    def unfuse_lora(self, base_state_dict: Optional[Mapping] = None):
        """Back to the base weights (re-packed from the checkpoint: exact, no subtraction error)."""
        self._repack_in_place(base_state_dict if base_state_dict is not None else self._base_view())
        self._lora = None
        return self

    def load_lora_adapter(self, pretrained_model_name_or_path_or_dict, lora_scale: float = 1.0, **kw):
        """``PeftAdapterMixin.load_lora_adapter`` entry point (loaders/peft.py:79): a state dict or a ``.safetensors`` file."""
        sd = pretrained_model_name_or_path_or_dict
        if not isinstance(sd, Mapping):
            sd = LazyCheckpoint([Path(sd)])
        return self.fuse_lora(sd, lora_scale, **kw)


def save_reference_checkpoint(state_dict: Mapping, config: Mapping, directory, variant: Optional[str] = None) -> Path:
    """Write a reference-layout model directory (``config.json`` + ``diffusion_pytorch_model[.variant].safetensors``), i.e.
    what the reference's ``save_pretrained`` writes -- used by tests and by tools that materialise seeded checkpoints."""
    from safetensors.torch import save_file
    d = Path(directory)
    d.mkdir(parents=True, exist_ok=True)
    cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in dict(config).items()}
    (d / CONFIG_NAME).write_text(json.dumps(cfg, indent=2, default=str))
    stem = WEIGHTS_NAME + (f".{variant}" if variant else "")
    save_file({k: v.detach().contiguous().cpu() for k, v in state_dict.items()}, str(d / f"{stem}.safetensors"))
    return d
