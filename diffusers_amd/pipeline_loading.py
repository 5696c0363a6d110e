"""Pipeline-level ``from_pretrained`` for the engine's pipelines: a local reference pipeline directory -> an engine pipeline.

The reference assembles a pipeline from ``model_index.json`` (``pipelines/pipeline_utils.py:739-1140``: one ``[library, class]``
pair per component slot, each component in its own sub-folder, components passed to the call taking precedence over the files).
The same contract here, over a LOCAL directory (no hub, no download):

* ``["diffusers", "<Model>"]`` slots load through the engine class of the same name (``Model.from_pretrained(dir, subfolder=slot)``:
  config.json + safetensors, packed once, cached -- ``loading.PretrainedMixin``);
* ``["diffusers", "<Scheduler>"]`` slots are built with ``Scheduler.from_config`` from ``<slot>/scheduler_config.json``;
* ``["transformers", ...]`` slots (text encoders, tokenizers) load through ``transformers`` itself -- the engine takes the caller's
  modules there (``text_encoding.py``) -- or, with ``text_encoders="engine"``, the text encoders run on the engine's kernels
  (``text_encoders.py``) from the same files;
* a slot whose class the engine does not have (a sampler outside the path: SD1.5's shipped ``PNDMScheduler``, a safety checker ...)
  must be passed in, set to ``None``, or -- for the safety checker / feature extractor -- is dropped; nothing is substituted silently.
"""
from __future__ import annotations

import importlib
import inspect
import json
from pathlib import Path
from typing import Optional

import torch

MODEL_INDEX = "model_index.json"
SCHEDULER_CONFIG = "scheduler_config.json"
_DROPPED = ("safety_checker", "feature_extractor", "image_encoder")     # outside the engine (pipelines.py refuses a safety checker)


def _engine_class(name: str):
    import diffusers_amd
    try:
        return getattr(diffusers_amd, name)
    except AttributeError:
        return None


class PipelineLoadingMixin:
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=None, variant: Optional[str] = None, device="cuda",
                        text_encoders: str = "transformers", cache_packed: bool = True, **passed):
        """Build this pipeline from a local reference pipeline directory.  ``passed`` components (``scheduler=``, ``unet=``,
        ``text_encoder=None`` ...) replace what the directory holds, as in the reference; ``text_encoders`` = "transformers" (the
        caller-side modules the directory names), "engine" (the same weights on the engine's kernels) or "none" (pipelines are then
        called with ``prompt_embeds``)."""
        if torch_dtype not in (None, torch.bfloat16):
            raise ValueError(f"{cls.__name__}.from_pretrained: torch_dtype={torch_dtype} -- the HIP engine computes in bfloat16")
        if text_encoders not in ("transformers", "engine", "none"):
            raise ValueError('text_encoders: "transformers", "engine" or "none"')
        root = Path(pretrained_model_name_or_path)
        if not (root / MODEL_INDEX).is_file():
            raise FileNotFoundError(f"{root}: no {MODEL_INDEX} (a local pipeline directory is needed; there is no hub access here)")
        index = json.loads((root / MODEL_INDEX).read_text())
        want = index.get("_class_name")
        if want and want != cls.__name__:
            raise ValueError(f"{root} holds a {want}; load it with diffusers_amd.{want}" if _engine_class(want) is not None else
                             f"{root} holds a {want}, which this engine does not implement")
        params = inspect.signature(cls.__init__).parameters
        kwargs = {}
        for slot, spec in index.items():
            if slot.startswith("_"):
                continue
            if slot in passed:
                continue
            if slot not in params:
                if isinstance(spec, (list, tuple)) and slot not in _DROPPED and spec[0] is not None:
                    raise ValueError(f"{cls.__name__} has no component slot {slot!r} ({spec})")
                continue
            if not isinstance(spec, (list, tuple)):            # a plain config value (force_zeros_for_empty_prompt, ...)
                kwargs[slot] = spec
                continue
            library, class_name = spec
            if library is None or class_name is None or slot in _DROPPED:
                kwargs[slot] = None
                continue
            kwargs[slot] = cls._load_component(root, slot, library, class_name, device, variant, text_encoders, cache_packed)
        unknown = [k for k in passed if k not in params]
        if unknown:
            raise TypeError(f"{cls.__name__}.from_pretrained: unexpected components {unknown}")
        kwargs.update(passed)
        pipe = cls(**{k: v for k, v in kwargs.items() if k in params})
        pipe._name_or_path = str(root)
        return pipe

    @staticmethod
    def _load_component(root: Path, slot: str, library: str, class_name: str, device, variant, text_encoders: str, cache_packed: bool):
        sub = root / slot
        if library == "diffusers":
            klass = _engine_class(class_name)
            if klass is None:
                raise NotImplementedError(
                    f"{slot}: the engine has no {class_name} (its samplers: DDIM, DDPM, EulerDiscrete, FlowMatchEulerDiscrete, "
                    f"UniPCMultistep; its models: the U-Nets, DiTs and VAEs of SURVEY.md 8a) -- pass `{slot}=` yourself")
            if (sub / SCHEDULER_CONFIG).is_file():
                return klass.from_config(json.loads((sub / SCHEDULER_CONFIG).read_text()))
            return klass.from_pretrained(root, subfolder=slot, variant=variant, device=device, cache_packed=cache_packed)
        if library == "transformers":
            is_tokenizer = "Tokenizer" in class_name
            if not is_tokenizer and text_encoders == "none":
                return None
            transformers = importlib.import_module("transformers")
            if is_tokenizer:
                return getattr(transformers, class_name).from_pretrained(str(sub))
            if text_encoders == "engine":
                klass = _engine_class(class_name)
                if klass is None:
                    raise NotImplementedError(f"{slot}: no engine-kernel {class_name}; use text_encoders='transformers'")
                from .loading import LazyCheckpoint, _weight_files
                model = klass(json.loads((sub / "config.json").read_text()))
                model.load_state_dict(LazyCheckpoint(_weight_files(sub, variant, stem="model")), device=device)
                return model
            # (`variant`: a directory that ships only model.fp16.safetensors must load its text encoders too, pipeline_utils.py:1004)
            return getattr(transformers, class_name).from_pretrained(str(sub), torch_dtype=torch.bfloat16, variant=variant).to(device)
        raise NotImplementedError(f"{slot}: components of library {library!r} are not loaded by the engine; pass `{slot}=`")

    # ---- the small surface callers of DiffusionPipeline objects rely on ----------------------------------------------------
    @property
    def components(self) -> dict:
        """``DiffusionPipeline.components`` (pipeline_utils.py:1882-1916): the constructor's component slots and what fills them."""
        params = inspect.signature(type(self).__init__).parameters
        return {k: getattr(self, k) for k in params if k != "self" and hasattr(self, k)}

    def to(self, *args, **kwargs):
        """``pipe.to("cuda")`` of reference scripts: the engine's models live where they were loaded; a move to their own device /
        dtype is a no-op, anything else is refused (models are re-loaded onto another device, not copied)."""
        movable = [m for m in self.components.values() if hasattr(m, "to") and (hasattr(m, "config") or isinstance(m, torch.nn.Module))]
        # the engine components first: they refuse what they cannot honour (another dtype / device) BEFORE a transformers module of the
        # same pipeline has been converted -- a refused call leaves the pipeline as it was
        for m in sorted(movable, key=lambda m: isinstance(m, torch.nn.Module)):
            m.to(*args, **kwargs)
        return self
