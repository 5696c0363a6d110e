"""TEST / BASELINE INFRASTRUCTURE ONLY -- the reference package at run time, wherever this repository runs.

`load_reference()` imports huggingface/diffusers v0.40.0.dev0 from `oracle/_ref/diffusers_ref.zip` (built by
`oracle/build_ref.py` from `/root/reference/src` in the build container; git-ignored, ships to the GPU box next to the built
`.so`) and returns the module, or None when the archive is absent.  Nothing in `diffusers_amd/` imports this file; users are
`bench.py`'s baseline / parity legs and `tests/` (enforced by tests/test_bench_contract.py).

The builders below construct the REAL reference objects for BASELINE config 3 from the same seeded
reference-format state dicts the engine packs, so "stock diffusers on PyTorch-ROCm" runs on identical weights."""
from __future__ import annotations

import sys
from pathlib import Path
from typing import Optional

import torch

ARCHIVE = Path(__file__).resolve().parent / "_ref" / "diffusers_ref.zip"
EXPECT_VERSION = "0.40.0.dev0"
_mod = None


def available() -> bool:
    return ARCHIVE.exists()


def load_reference():
    """The reference package (module `diffusers`), imported from the shipped archive; None if it did not ship."""
    global _mod
    if _mod is not None:
        return _mod
    if not ARCHIVE.exists():
        return None
    if "diffusers" in sys.modules and not str(getattr(sys.modules["diffusers"], "__file__", "")).startswith(str(ARCHIVE)):
        mod = sys.modules["diffusers"]          # the build container's live checkout (tests put /root/reference/src on the path)
    else:
        sys.path.insert(0, str(ARCHIVE))
        try:
            import diffusers as mod  # noqa: F401
        finally:
            sys.path.remove(str(ARCHIVE))
    if mod.__version__ != EXPECT_VERSION:
        raise RuntimeError(f"reference archive is version {mod.__version__}, expected {EXPECT_VERSION}")
    _mod = mod
    return mod


def _lists(cfg: dict) -> dict:
    return {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}


def build_unet(ref, cfg: dict, state_dict: dict, device, dtype) -> "torch.nn.Module":
    """`UNet2DConditionModel(**cfg)` of the reference with the given reference-format weights (models/unets/unet_2d_condition.py)."""
    with torch.device("meta"):
        m = ref.UNet2DConditionModel(**_lists(cfg))
    m = m.to_empty(device=device)
    missing, unexpected = m.load_state_dict({k: v.to(device=device, dtype=dtype) for k, v in state_dict.items()}, strict=False, assign=True)
    if missing or unexpected:
        raise RuntimeError(f"reference U-Net state_dict mismatch: missing {missing[:3]} unexpected {unexpected[:3]}")
    return m.to(dtype).eval()


def build_vae(ref, cfg: dict, decoder_state_dict: dict, device, dtype) -> "torch.nn.Module":
    """`AutoencoderKL(**cfg)`; the engine's state dicts hold the DECODER half (decoder.*, post_quant_conv.*): the encoder keeps its
    default initialisation -- it is not on the path."""
    m = ref.AutoencoderKL(**_lists(cfg)).to(device=device, dtype=dtype)
    own = m.state_dict()
    sd = {k: v.to(device=device, dtype=dtype) for k, v in decoder_state_dict.items()}
    unknown = [k for k in sd if k not in own]
    if unknown:
        raise RuntimeError(f"reference VAE has no parameters {unknown[:3]}")
    m.load_state_dict(sd, strict=False)
    return m.eval()


def build_sdxl_pipeline(ref, ucfg: dict, vcfg: dict, unet_sd: dict, vae_sd: dict, scheduler_kwargs: dict, device, dtype):
    """The unchanged `StableDiffusionXLPipeline` (pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl.py:823-1308) over
    reference components; no text encoders (the harness passes prompt embeddings)."""
    unet = build_unet(ref, ucfg, unet_sd, device, dtype)
    vae = build_vae(ref, vcfg, vae_sd, device, dtype)
    sch = ref.EulerDiscreteScheduler(**scheduler_kwargs)
    pipe = ref.StableDiffusionXLPipeline(vae=vae, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None,
                                         unet=unet, scheduler=sch)
    pipe.set_progress_bar_config(disable=True)
    return pipe


def run_sdxl(pipe, inp: dict, steps: int, guidance: float, hw: int, dtype, output_type: str = "pt", want_latents: bool = False):
    """One `__call__` on the harness inputs (bf16 tensors of bench.synth_inputs).  Returns (images in [0, 1] NCHW, final latents
    or None).  The final latents are captured with the pipeline's own `callback_on_step_end` hook."""
    last = {}

    def grab(p, i, t, kw):
        last["latents"] = kw["latents"]
        return {}
    kw = dict(prompt_embeds=inp["prompt_embeds"].to(dtype), negative_prompt_embeds=inp["negative_prompt_embeds"].to(dtype),
              pooled_prompt_embeds=inp["pooled"].to(dtype), negative_pooled_prompt_embeds=inp["negative_pooled"].to(dtype),
              latents=inp["latents"].to(dtype).clone(), num_inference_steps=steps, guidance_scale=guidance, height=hw, width=hw,
              output_type=output_type)
    if want_latents:
        kw.update(callback_on_step_end=grab, callback_on_step_end_tensor_inputs=["latents"])
    with torch.no_grad():
        img = pipe(**kw).images
    return img, (last.get("latents") if want_latents else None)
