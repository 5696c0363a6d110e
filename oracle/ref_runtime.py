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


def build_model(ref, cls_name: str, cfg: dict, state_dict: dict, device, dtype, keep_fp32=()) -> "torch.nn.Module":
    """Any reference model class (`FluxTransformer2DModel`, `WanTransformer3DModel`, `UNet2DModel`, ...) built on ``device`` and
    filled with the given reference-format weights (strict).  ``keep_fp32``: parameter-name fragments the reference keeps in fp32
    whatever ``dtype`` is (`_keep_in_fp32_modules`, e.g. Wan's scale_shift_table / norms)."""
    with torch.device(device):            # (not the meta device: non-persistent buffers such as Wan's rope tables must be built)
        m = getattr(ref, cls_name)(**_lists(cfg))

    def cast(k, v):
        return v.to(device=device, dtype=torch.float32 if any(f in k for f in keep_fp32) else dtype)
    missing, unexpected = m.load_state_dict({k: cast(k, v) for k, v in state_dict.items()}, strict=False, assign=True)
    if missing or unexpected:
        raise RuntimeError(f"reference {cls_name} state_dict mismatch: missing {missing[:3]} unexpected {unexpected[:3]}")
    return m.eval()


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


# ---- the unchanged reference pipeline over ENGINE components (drop-in boundary B1 / B2 / B5 of SURVEY.md 8b) -------------------
def dummy_text_stack(device, dtype, proj_dim: int):
    """A tokenizer and a tiny random CLIP pair that are never called (the harness passes prompt embeddings): the reference
    pipeline derives `_execution_device` / the embedding dtype from its nn.Module components (pipeline_utils.py:1152) and
    `text_encoder_2.config.projection_dim` from the second encoder (pipeline_stable_diffusion_xl.py:737)."""
    from tokenizers import pre_tokenizers
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection, CLIPTokenizer
    alpha = sorted(pre_tokenizers.ByteLevel.alphabet())
    vocab = {}
    for c in alpha:
        vocab[c] = len(vocab)
    for c in alpha:
        vocab[c + "</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    tok = CLIPTokenizer(vocab=vocab, merges=[], model_max_length=16)
    nv = len(vocab)

    def clip(proj, seed):
        torch.manual_seed(seed)
        cfg = CLIPTextConfig(vocab_size=nv, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                             max_position_embeddings=16, projection_dim=proj or 32, bos_token_id=nv - 2, eos_token_id=nv - 1,
                             pad_token_id=nv - 1)
        return (CLIPTextModelWithProjection(cfg) if proj else CLIPTextModel(cfg)).eval().to(device, dtype)
    return tok, clip(None, 1), clip(proj_dim, 2)


def engine_under_reference_sdxl(ref, unet, vae, scheduler, device, dtype, proj_dim: int):
    """`StableDiffusionXLPipeline` of the reference, UNCHANGED, with engine `unet` / `vae` / `scheduler` objects in its
    component slots (DiffusionPipeline.register_modules, pipelines/pipeline_utils.py:224-252): its own `__call__`
    (pipeline_stable_diffusion_xl.py:823-1308) then drives the HIP kernels step by step, eagerly."""
    tok, te, te2 = dummy_text_stack(device, dtype, proj_dim)
    pipe = ref.StableDiffusionXLPipeline(vae=vae, text_encoder=te, text_encoder_2=te2, tokenizer=tok, tokenizer_2=tok,
                                         unet=unet, scheduler=scheduler)
    pipe.set_progress_bar_config(disable=True)
    return pipe


# ---- reference classes of the OTHER BASELINE configs, random weights, for the CPU legs (timing only) ---------------------------
def cpu_step_seconds(ref, config: str, cfgs: dict, threads: int):
    """Seconds of ONE bounded reference forward on the host cores in fp32 and its algorithmic TFLOP, per BASELINE config:
      sd15  UNet2DConditionModel.forward, CFG batch 2, 64x64 latents (1.6065 TFLOP)  [+ AutoencoderKL.decode 512x512, 2.5145]
      ddpm  UNet2DModel.forward, batch 1, 256x256 (0.4970 TFLOP)
      flux  FluxTransformer2DModel.forward at full width and sequence (4096 + 512 tokens) with 2 double + 2 single blocks of the
            19 + 38 (every block is 1.305 TFLOP: 5.22 TFLOP)
      wan   WanTransformer3DModel.forward at full width with TWO of the 30 blocks on 9 of the 21 latent frames (14 040 tokens)
    Returns (seconds, tflop, description)."""
    import time
    torch.set_num_threads(threads)
    g = torch.Generator("cpu").manual_seed(3)
    with torch.no_grad():
        if config == "sd15":
            m = ref.UNet2DConditionModel(**_lists(cfgs["unet"])).eval()
            x, ehs = torch.randn((2, 4, 64, 64), generator=g), torch.randn((2, 77, 768), generator=g)
            t0 = time.perf_counter()
            m(x, torch.tensor(500.0), encoder_hidden_states=ehs, return_dict=False)
            ts = time.perf_counter() - t0
            del m
            v = ref.AutoencoderKL(**_lists(cfgs["vae"])).eval()
            t0 = time.perf_counter()
            v.decode(torch.randn((1, 4, 64, 64), generator=g), return_dict=False)
            td = time.perf_counter() - t0
            return (ts, 1.6065, td, 2.5145,
                    f"1 CFG-batched reference UNet2DConditionModel step at 64x64 latents ({ts:.1f} s) + 1 AutoencoderKL.decode to "
                    f"512x512 ({td:.1f} s); image = 50 x step + decode")
        if config == "ddpm":
            m = ref.UNet2DModel(**_lists(cfgs["unet"])).eval()
            x = torch.randn((1, 3, 256, 256), generator=g)
            m(x, torch.tensor(10), return_dict=False)
            t0 = time.perf_counter()
            m(x, torch.tensor(500), return_dict=False)
            ts = time.perf_counter() - t0
            return ts, 0.4970, 0.0, 0.0, f"1 reference UNet2DModel step at 256x256 ({ts:.1f} s); image = 50 x step"
        if config == "flux":
            c = dict(_lists(cfgs["transformer"]), num_layers=2, num_single_layers=2)
            m = ref.FluxTransformer2DModel(**c).eval()
            hs, ehs = torch.randn((1, 4096, 64), generator=g), torch.randn((1, 512, 4096), generator=g)
            pooled = torch.randn((1, 768), generator=g)
            img_ids, txt_ids = torch.zeros((4096, 3)), torch.zeros((512, 3))
            img_ids[:, 1] = torch.arange(4096) // 64
            img_ids[:, 2] = torch.arange(4096) % 64
            t0 = time.perf_counter()
            m(hidden_states=hs, encoder_hidden_states=ehs, pooled_projections=pooled, timestep=torch.tensor([0.5]), img_ids=img_ids,
              txt_ids=txt_ids, guidance=None, return_dict=False)
            ts = time.perf_counter() - t0
            return (ts, 4 * 1.305, 0.0, 0.0,
                    f"reference FluxTransformer2DModel.forward, full width, 4096 + 512 tokens, 2 double + 2 single blocks of the 19 + 38 "
                    f"(5.22 TFLOP, {ts:.1f} s); EXTRAPOLATED by algorithmic FLOPs to the 308 TFLOP of an image")
        if config == "wan":
            c = dict(_lists(cfgs["transformer"]), num_layers=2)
            m = ref.WanTransformer3DModel(**c).eval()
            frames = 9
            tokens = frames * 30 * 52
            hs = torch.randn((1, 16, frames, 60, 104), generator=g)
            ehs = torch.randn((1, 512, 4096), generator=g)
            t0 = time.perf_counter()
            m(hidden_states=hs, timestep=torch.tensor([500]), encoder_hidden_states=ehs, return_dict=False)
            ts = time.perf_counter() - t0
            tfl = 2 * (2.73 * tokens / 32760 + 6.594 * (tokens / 32760) ** 2)
            return (ts, tfl, 0.0, 0.0,
                    f"reference WanTransformer3DModel.forward, full width, TWO of the 30 blocks on {frames} of the 21 latent frames "
                    f"({tokens} tokens, {tfl:.2f} TFLOP, {ts:.1f} s); EXTRAPOLATED by algorithmic FLOPs to the 28 300 TFLOP of a video")
    raise ValueError(config)
