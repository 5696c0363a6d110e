"""The REAL reference package on the GPU box (oracle/_ref/diffusers_ref.zip, built by oracle/build_ref.py; skipped when it did
not ship): boundary B1 / B2 / B5 of SURVEY.md 8b on hardware.

1. engine `UNet2DConditionModel` + `AutoencoderKL` + `EulerDiscreteScheduler` registered into the UNCHANGED reference
   `StableDiffusionXLPipeline` (pipelines/pipeline_utils.py:224-252), whose own `__call__`
   (pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl.py:823-1308) then drives the HIP kernels: cat([latents] * 2),
   scale_model_input, unet(...), torch CFG combine, scheduler.step, vae.decode, postprocess -- PSNR >= 40 dB against the
   all-reference fp32 run on the same weights / embeddings / latents;
2. the engine's own pipeline against the same all-reference run (what bench.py's `parity` leg does at full size)."""
import numpy as np
import pytest
import torch

from oracle import ref_runtime as RR

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not RR.available(), reason="reference archive oracle/_ref/diffusers_ref.zip did not ship")]
bf16 = torch.bfloat16
DEV = "cuda"


def _psnr(a, b):
    return 10 * np.log10(1.0 / max(float((a.float() - b.float()).pow(2).mean()), 1e-12))


def _setup(tiny=True):
    import diffusers_amd as da
    from diffusers_amd import factory, init as dinit
    ref = RR.load_reference()
    ucfg, vcfg = (dinit.TINY_SDXL_UNET, dinit.TINY_VAE) if tiny else (dinit.SDXL_UNET, dinit.SDXL_VAE)
    unet, usd = factory.build_unet(ucfg, seed=0, device=DEV, init_device="cpu" if tiny else DEV)
    vae, vsd = factory.build_vae(vcfg, seed=1, device=DEV, init_device="cpu" if tiny else DEV)
    cd, pd, lat = (64, 64, 16) if tiny else (2048, 1280, 128)
    g = torch.Generator("cpu").manual_seed(1234)
    inp = {"prompt_embeds": torch.randn((1, 77, cd), generator=g), "negative_prompt_embeds": torch.randn((1, 77, cd), generator=g),
           "pooled": torch.randn((1, pd), generator=g), "negative_pooled": torch.randn((1, pd), generator=g),
           "latents": torch.randn((1, 4, lat, lat), generator=g)}
    inp = {k: v.to(bf16).to(DEV) for k, v in inp.items()}
    return da, factory, ref, ucfg, vcfg, unet, usd, vae, vsd, inp


@pytest.mark.parametrize("steps", [4])
def test_engine_components_under_the_unchanged_reference_pipeline_on_hardware(steps):
    da, factory, ref, ucfg, vcfg, unet, usd, vae, vsd, inp = _setup(tiny=True)
    hw = 16 * 2 ** (len(vcfg["block_out_channels"]) - 1)
    rpipe = RR.build_sdxl_pipeline(ref, ucfg, vcfg, usd, vsd, factory.SDXL_SCHEDULER, DEV, torch.float32)
    want, want_lat = RR.run_sdxl(rpipe, inp, steps, 5.0, hw, torch.float32, want_latents=True)        # all-reference, fp32
    # the reference pipeline object, engine components in its slots
    rpipe.to(bf16)
    rpipe.register_modules(unet=unet, vae=vae, scheduler=da.EulerDiscreteScheduler.from_config(rpipe.scheduler.config))
    # DiffusionPipeline._execution_device is the device of the pipeline's nn.Module components (pipeline_utils.py:1152): in
    # real use the text encoders; here (prompt embeddings are passed) a tiny CLIP pair that is never called
    from test_text_encoding import _clip, _tokenizer
    tok, nv = _tokenizer()
    rpipe.register_modules(tokenizer=tok, tokenizer_2=tok, text_encoder=_clip(nv, 32, seed=1).to(DEV, bf16),
                           text_encoder_2=_clip(nv, 32, proj=64, seed=2).to(DEV, bf16))
    assert str(rpipe._execution_device).startswith("cuda")
    got, got_lat = RR.run_sdxl(rpipe, inp, steps, 5.0, hw, bf16, want_latents=True)
    ps = _psnr(got, want)
    rr = float((got_lat.float() - want_lat.float()).pow(2).mean().sqrt() / want_lat.float().pow(2).mean().sqrt())
    print(f"[drop-in] engine under the reference SDXL __call__ on the GPU: PSNR {ps:.1f} dB, latents rel-rms {rr:.3e} vs the all-reference fp32 run")
    assert got.shape == want.shape and ps >= 40.0
    # the engine's own pipeline (fused CFG + Euler step, HIP graph) lands on the same image
    epipe = factory.build_sdxl_pipeline(device=DEV, tiny=True, seed=0)
    img = epipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"],
                pooled_prompt_embeds=inp["pooled"], negative_pooled_prompt_embeds=inp["negative_pooled"],
                latents=inp["latents"].clone(), num_inference_steps=steps, guidance_scale=5.0, height=hw, width=hw,
                output_type="pt").images
    ps2 = _psnr(img, want)
    print(f"[drop-in] engine pipeline vs the all-reference fp32 run: PSNR {ps2:.1f} dB; vs engine-under-reference: {_psnr(img, got):.1f} dB")
    assert ps2 >= 40.0


def test_reference_classes_accept_the_engine_state_dicts_at_full_size():
    """The seeded reference-format state dicts the engine packs load into the REAL reference classes (strictly for the U-Net,
    decoder half for the VAE), and one full-size U-Net forward agrees: engine (bf16, HIP) vs reference (fp32, PyTorch-ROCm)."""
    da, factory, ref, ucfg, vcfg, unet, usd, vae, vsd, inp = _setup(tiny=False)
    runet = RR.build_unet(ref, ucfg, usd, DEV, torch.float32)
    g = torch.Generator("cpu").manual_seed(5)
    sample = torch.randn((2, 4, 64, 64), generator=g).to(DEV)
    ehs = torch.randn((2, 77, 2048), generator=g).to(DEV)
    added = {"text_embeds": torch.randn((2, 1280), generator=g).to(DEV),
             "time_ids": torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]], device=DEV).repeat(2, 1)}
    with torch.no_grad():
        want = runet(sample, torch.tensor(481.0, device=DEV), encoder_hidden_states=ehs, added_cond_kwargs=added, return_dict=False)[0]
    got = unet(sample.to(bf16), torch.tensor(481.0), ehs.to(bf16),
               added_cond_kwargs={"text_embeds": added["text_embeds"].to(bf16), "time_ids": added["time_ids"]}).sample
    rr = float((got.float() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
    print(f"[parity] full SDXL U-Net (64x64 latents) engine vs the REAL reference class in fp32: rel-rms {rr:.3e}")
    assert rr < 2.5e-2



def test_engine_components_under_the_reference_pipeline_at_full_size():
    """The drop-in at BASELINE size: SDXL-base U-Net + VAE engine objects in the component slots of the
    reference `StableDiffusionXLPipeline`, its own `__call__` at 1024 x 1024 (8 steps here; bench.py's `dropin` leg times the 50-step
    call): image PSNR against the all-reference fp32 run within 1 dB of the all-reference bf16 run's, and the engine's own graphed
    pipeline on the same inputs lands on the same image."""
    import diffusers_amd as da
    da_, factory, ref, ucfg, vcfg, unet, usd, vae, vsd, inp = _setup(tiny=False)
    steps, hw = 8, 1024
    rpipe = RR.build_sdxl_pipeline(ref, ucfg, vcfg, usd, vsd, factory.SDXL_SCHEDULER, DEV, torch.float32)
    want, _ = RR.run_sdxl(rpipe, inp, steps, 5.0, hw, torch.float32)
    rpipe.to(bf16)
    floor, _ = RR.run_sdxl(rpipe, inp, steps, 5.0, hw, bf16)
    del rpipe
    torch.cuda.empty_cache()
    dpipe = RR.engine_under_reference_sdxl(ref, unet, vae, da.EulerDiscreteScheduler(**factory.SDXL_SCHEDULER), DEV, bf16, 1280)
    assert str(dpipe._execution_device).startswith("cuda")
    got, _ = RR.run_sdxl(dpipe, inp, steps, 5.0, hw, bf16)
    ps, pf = _psnr(got, want), _psnr(floor, want)
    print(f"[drop-in] full size, {steps} steps: engine under the reference SDXL __call__ vs the all-reference fp32 run: PSNR {ps:.1f} dB "
          f"(all-reference bf16 run: {pf:.1f} dB)")
    assert got.shape == want.shape == (1, 3, hw, hw) and ps >= 40.0 and ps >= pf - 1.0
    from diffusers_amd.pipelines import StableDiffusionXLPipeline
    epipe = StableDiffusionXLPipeline(vae=vae, unet=unet, scheduler=da.EulerDiscreteScheduler(**factory.SDXL_SCHEDULER))
    img = epipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"],
                pooled_prompt_embeds=inp["pooled"], negative_pooled_prompt_embeds=inp["negative_pooled"],
                latents=inp["latents"].clone(), num_inference_steps=steps, guidance_scale=5.0, height=hw, width=hw, output_type="pt").images
    ps2 = _psnr(img, want)
    print(f"[drop-in] full size: engine pipeline vs the all-reference fp32 run: PSNR {ps2:.1f} dB; vs engine-under-reference: {_psnr(img, got):.1f} dB")
    assert ps2 >= 40.0 and ps2 >= pf - 1.0


# ---- the other three pipeline families on hardware: the unchanged reference `__call__` over engine components ----
def _families():
    import diffusers_amd as da
    from diffusers_amd import factory, init as dinit
    from test_text_encoding import _as_lists, _clip, _t5, _t5_tokenizer, _tokenizer
    return da, factory, dinit, _as_lists, _clip, _t5, _t5_tokenizer, _tokenizer


def test_engine_under_the_reference_flux_pipeline_on_hardware():
    """`FluxPipeline.__call__` (pipelines/flux/pipeline_flux.py:886-960) with engine FluxTransformer2DModel + AutoencoderKL +
    FlowMatchEulerDiscreteScheduler in its component slots: latent packing, image ids, timestep / 1000, sigmas= schedule."""
    da, factory, dinit, _as_lists, _clip, _t5, _t5_tokenizer, _tokenizer = _families()
    ref = RR.load_reference()
    tok, nv = _tokenizer()
    tok2, nv2 = _t5_tokenizer()
    torch.manual_seed(0)
    rtr = ref.FluxTransformer2DModel(**_as_lists(dinit.TINY_FLUX)).eval().to(DEV)
    rvae = ref.AutoencoderKL(**_as_lists(dinit.TINY_FLUX_VAE)).eval().to(DEV)
    rs = ref.FlowMatchEulerDiscreteScheduler(shift=1.0, use_dynamic_shifting=False)
    pipe = ref.FluxPipeline(scheduler=rs, vae=rvae, text_encoder=_clip(nv, 64, seed=3).to(DEV), tokenizer=tok,
                            text_encoder_2=_t5(nv2, d=64, seed=4).to(DEV), tokenizer_2=tok2, transformer=rtr)
    pipe.set_progress_bar_config(disable=True)
    lat = torch.randn(1, 64, 64, generator=torch.Generator().manual_seed(1)).to(DEV)
    kw = dict(prompt="hello a cat", num_inference_steps=3, guidance_scale=0.0, height=32, width=32, output_type="pt", max_sequence_length=16)
    with torch.no_grad():
        want = pipe(latents=lat.clone(), **kw).images                          # all-reference, fp32
        pipe.to(bf16)
        floor = pipe(latents=lat.clone().to(bf16), **kw).images                # all-reference, bf16
    tr = da.from_reference_config(da.FluxTransformer2DModel, rtr.config)
    tr.load_state_dict(rtr.state_dict(), device=DEV)
    vae = da.from_reference_config(da.AutoencoderKL, rvae.config)
    vae.load_state_dict(rvae.state_dict(), device=DEV)
    pipe.register_modules(transformer=tr, vae=vae, scheduler=da.FlowMatchEulerDiscreteScheduler.from_config(rs.config))
    with torch.no_grad():
        got = pipe(latents=lat.clone().to(bf16), **kw).images
    ps, pf = _psnr(got, want), _psnr(floor, want)
    print(f"[drop-in] engine under the reference FluxPipeline.__call__ on the GPU: PSNR {ps:.1f} dB vs the all-reference fp32 run "
          f"(all-reference bf16: {pf:.1f} dB)")
    assert got.shape == want.shape and ps >= 40.0


def test_engine_under_the_reference_wan_pipeline_on_hardware():
    """`WanPipeline.__call__` (pipelines/wan/pipeline_wan.py:560-661) as shipped -- UniPC (flow, order 2), two transformer calls per
    step, fp32 latents, latent de-normalisation, AutoencoderKLWan.decode, video post-processing -- over engine components."""
    da, factory, dinit, _as_lists, _clip, _t5, _t5_tokenizer, _tokenizer = _families()
    ref = RR.load_reference()
    tok, nv = _t5_tokenizer()
    torch.manual_seed(0)
    rtr = ref.WanTransformer3DModel(**_as_lists(dinit.TINY_WAN)).eval().to(DEV)
    rvae = ref.AutoencoderKLWan(**_as_lists(dinit.TINY_WAN_VAE)).eval().to(DEV)
    rs = ref.UniPCMultistepScheduler(prediction_type="flow_prediction", use_flow_sigmas=True, flow_shift=3.0)
    pipe = ref.WanPipeline(tokenizer=tok, text_encoder=_t5(nv, d=64, seed=5, umt5=True).to(DEV), vae=rvae, scheduler=rs, transformer=rtr)
    pipe.set_progress_bar_config(disable=True)
    lat = torch.randn(1, 16, 3, 8, 8, generator=torch.Generator().manual_seed(1)).to(DEV)
    kw = dict(prompt="a cat on the mat", negative_prompt="red", num_inference_steps=3, guidance_scale=5.0, height=64, width=64,
              num_frames=9, output_type="pt", max_sequence_length=16)
    with torch.no_grad():
        want = pipe(latents=lat.clone(), **kw).frames
        pipe.to(bf16)
        floor = pipe(latents=lat.clone(), **kw).frames
    tr = da.from_reference_config(da.WanTransformer3DModel, rtr.config)
    tr.load_state_dict(rtr.state_dict(), device=DEV)
    vae = da.from_reference_config(da.AutoencoderKLWan, rvae.config)
    vae.load_state_dict(rvae.state_dict(), device=DEV)
    pipe.register_modules(transformer=tr, vae=vae, scheduler=da.UniPCMultistepScheduler.from_config(rs.config))
    with torch.no_grad():
        got = pipe(latents=lat.clone(), **kw).frames
    ps, pf = _psnr(got, want), _psnr(floor, want)
    print(f"[drop-in] engine under the reference WanPipeline.__call__ (UniPC + AutoencoderKLWan) on the GPU: PSNR {ps:.1f} dB vs the "
          f"all-reference fp32 run (all-reference bf16: {pf:.1f} dB)")
    assert got.shape == want.shape == (1, 9, 3, 64, 64) and ps >= 40.0


def test_engine_under_the_reference_sd_pipeline_on_hardware():
    """`StableDiffusionPipeline.__call__` (pipelines/stable_diffusion/pipeline_stable_diffusion.py:1030-1060) with DDIM over engine
    UNet2DConditionModel (SD1.5 block layout) + AutoencoderKL + DDIMScheduler."""
    da, factory, dinit, _as_lists, _clip, _t5, _t5_tokenizer, _tokenizer = _families()
    ref = RR.load_reference()
    tok, nv = _tokenizer()
    torch.manual_seed(0)
    runet = ref.UNet2DConditionModel(**_as_lists(dinit.TINY_SD15_UNET)).eval().to(DEV)
    rvae = ref.AutoencoderKL(**_as_lists(dinit.TINY_VAE)).eval().to(DEV)
    rs = ref.DDIMScheduler(**factory.SD15_SCHEDULER)
    pipe = ref.StableDiffusionPipeline(vae=rvae, text_encoder=_clip(nv, 64, seed=3).to(DEV), tokenizer=tok, unet=runet, scheduler=rs,
                                       safety_checker=None, feature_extractor=None, requires_safety_checker=False)
    pipe.set_progress_bar_config(disable=True)
    lat = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(2)).to(DEV)
    kw = dict(prompt="hello a cat", negative_prompt="cat", num_inference_steps=4, guidance_scale=7.5, height=32, width=32, output_type="pt")
    with torch.no_grad():
        want = pipe(latents=lat.clone(), **kw).images
        pipe.to(bf16)
        floor = pipe(latents=lat.clone().to(bf16), **kw).images
    unet = da.from_reference_config(da.UNet2DConditionModel, runet.config)
    unet.load_state_dict(runet.state_dict(), device=DEV)
    vae = da.from_reference_config(da.AutoencoderKL, rvae.config)
    vae.load_state_dict(rvae.state_dict(), device=DEV)
    pipe.register_modules(unet=unet, vae=vae, scheduler=da.DDIMScheduler.from_config(rs.config))
    with torch.no_grad():
        got = pipe(latents=lat.clone().to(bf16), **kw).images
    ps, pf = _psnr(got, want), _psnr(floor, want)
    print(f"[drop-in] engine under the reference StableDiffusionPipeline.__call__ (DDIM) on the GPU: PSNR {ps:.1f} dB vs the all-reference "
          f"fp32 run (all-reference bf16: {pf:.1f} dB)")
    assert got.shape == want.shape and ps >= 40.0
