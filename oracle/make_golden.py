"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by running the REAL reference (huggingface/diffusers
imported from /root/reference/src) on CPU with seeded weights and inputs.  Run in the build container only:

    PYTHONPATH=/root/reference/src python oracle/make_golden.py

The GPU box has no /root/reference; tests there consume the committed fixtures.  Weights are not stored: they are
regenerated from ``diffusers_amd.init.random_state_dict`` (deterministic per tensor name + seed), and this script
asserts that inventory equals the reference modules' own ``state_dict()`` key/shape set (strict load).
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
REF_SRC = "/root/reference/src"
if REF_SRC not in sys.path:
    sys.path.insert(0, REF_SRC)

import diffusers  # noqa: E402  (the reference)
from diffusers import (AutoencoderKL, DDIMScheduler, DDPMScheduler, EulerDiscreteScheduler,  # noqa: E402
                       FlowMatchEulerDiscreteScheduler, StableDiffusionPipeline, StableDiffusionXLPipeline,
                       UNet2DConditionModel)

from diffusers_amd import init as dinit  # noqa: E402

GOLD = ROOT / "tests" / "golden"
GOLD.mkdir(parents=True, exist_ok=True)


def gen(seed):
    return torch.Generator("cpu").manual_seed(seed)


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def build_unet(cfg, seed=0):
    ref = UNet2DConditionModel(**cfg).eval()
    shapes = dinit.unet_param_shapes(dict(ref.config))
    ref_shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert ref_shapes == dict(shapes), (set(ref_shapes) ^ set(shapes))
    sd = dinit.random_state_dict(shapes, seed=seed)
    ref.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    return ref


def build_vae(cfg, seed=1):
    ref = AutoencoderKL(**cfg).eval()
    shapes = dinit.vae_decoder_param_shapes(dict(ref.config))
    ref_sd = ref.state_dict()
    ref_shapes = {k: tuple(v.shape) for k, v in ref_sd.items() if k.startswith(("decoder.", "post_quant_conv."))}
    assert ref_shapes == dict(shapes), (set(ref_shapes) ^ set(shapes))
    sd = dinit.random_state_dict(shapes, seed=seed)
    full = {k: v.clone() for k, v in ref_sd.items()}
    full.update({k: v.float() for k, v in sd.items()})
    ref.load_state_dict(full, strict=True)
    return ref


def unet_case(name, cfg, with_added):
    ref = build_unet(cfg)
    g = gen(1234)
    B, S = 2, 7
    hw = cfg["sample_size"]
    sample = bf16_round(torch.randn((B, 4, hw, hw), generator=g))
    ehs = bf16_round(torch.randn((B, S, cfg["cross_attention_dim"]), generator=g))
    t = torch.tensor(801.0)
    kw = {}
    out = {"sample": sample.numpy(), "ehs": ehs.numpy(), "t": np.float32(801.0)}
    if with_added:
        te = bf16_round(torch.randn((B, 64), generator=g))
        ids = torch.tensor([[16., 16., 0., 0., 16., 16.]]).repeat(B, 1)
        kw["added_cond_kwargs"] = {"text_embeds": te, "time_ids": ids}
        out["text_embeds"] = te.numpy()
        out["time_ids"] = ids.numpy()
    with torch.no_grad():
        y = ref(sample, t, ehs, **kw).sample
    out["out"] = y.numpy()
    np.savez_compressed(GOLD / f"{name}.npz", **out)
    print(name, "out rms", float(y.pow(2).mean().sqrt()))
    return ref


def vae_case(name, cfg):
    ref = build_vae(cfg)
    g = gen(77)
    z = bf16_round(torch.randn((1, cfg["latent_channels"], 16, 16), generator=g))
    with torch.no_grad():
        y = ref.decode(z).sample
    np.savez_compressed(GOLD / f"{name}.npz", z=z.numpy(), out=y.numpy())
    print(name, "out rms", float(y.pow(2).mean().sqrt()))
    return ref


def scheduler_cases():
    out = {}
    g = gen(5)
    shape = (2, 4, 8, 8)
    # --- schedule tables at the BASELINE configs (50 steps SDXL Euler / SD1.5 DDIM, 4 steps Flux) ---
    e = EulerDiscreteScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1,
                               timestep_spacing="leading")
    e.set_timesteps(50)
    out["euler50_timesteps"] = e.timesteps.numpy()
    out["euler50_sigmas"] = e.sigmas.numpy()
    out["euler50_init_sigma"] = np.float32(float(e.init_noise_sigma))
    d = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                      set_alpha_to_one=False, steps_offset=1)
    d.set_timesteps(50)
    out["ddim50_timesteps"] = d.timesteps.numpy()
    f = FlowMatchEulerDiscreteScheduler(shift=1.0)
    sig = np.linspace(1.0, 1 / 4, 4)
    f.set_timesteps(sigmas=sig)
    out["flow4_timesteps"] = f.timesteps.numpy()
    out["flow4_sigmas"] = f.sigmas.numpy()
    fd = FlowMatchEulerDiscreteScheduler(shift=3.0, use_dynamic_shifting=True)
    fd.set_timesteps(sigmas=np.linspace(1.0, 1 / 28, 28), mu=1.15)
    out["flowdyn28_sigmas"] = fd.sigmas.numpy()

    # --- step trajectories on random tensors, fp32 and bf16, 5 steps each ---
    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        x0 = torch.randn(shape, generator=g).to(dt)
        eps = [torch.randn(shape, generator=g).to(dt) for _ in range(5)]
        out[f"x0_{dt_name}"] = x0.float().numpy()
        out[f"eps_{dt_name}"] = torch.stack(eps).float().numpy()
        # Euler
        e.set_timesteps(5)
        x = x0 * e.init_noise_sigma
        if dt == torch.bfloat16:
            x = x.to(dt)
        traj, scaled = [], []
        for i, t in enumerate(e.timesteps):
            scaled.append(e.scale_model_input(x, t).float())
            x = e.step(eps[i], t, x).prev_sample
            traj.append(x.float())
        out[f"euler_start_{dt_name}"] = (x0 * e.init_noise_sigma).to(dt).float().numpy()
        out[f"euler_traj_{dt_name}"] = torch.stack(traj).numpy()
        out[f"euler_scaled_{dt_name}"] = torch.stack(scaled).numpy()
        # DDIM
        d.set_timesteps(5)
        x = x0.clone()
        traj = []
        for i, t in enumerate(d.timesteps):
            x = d.step(eps[i], t, x).prev_sample
            traj.append(x.float())
        out[f"ddim_traj_{dt_name}"] = torch.stack(traj).numpy()
        # DDPM (noise from a CPU generator seeded 9)
        p = DDPMScheduler(beta_start=0.0001, beta_end=0.02, beta_schedule="linear", variance_type="fixed_small",
                          clip_sample=True)
        p.set_timesteps(5)
        gg = gen(9)
        x = x0.clone()
        traj = []
        for i, t in enumerate(p.timesteps):
            x = p.step(eps[i], t, x, generator=gg).prev_sample
            traj.append(x.float())
        out[f"ddpm_traj_{dt_name}"] = torch.stack(traj).numpy()
        # FlowMatch
        f.set_timesteps(sigmas=np.linspace(1.0, 1 / 5, 5))
        x = x0.clone()
        traj = []
        for i, t in enumerate(f.timesteps):
            x = f.step(eps[i], t, x).prev_sample
            traj.append(x.float())
        out[f"flow_traj_{dt_name}"] = torch.stack(traj).numpy()
        # CFG combine (pipeline_stable_diffusion.py:1054-1055)
        u, c = eps[0], eps[1]
        out[f"cfg_{dt_name}"] = (u + 7.5 * (c - u)).float().numpy()
    np.savez_compressed(GOLD / "schedulers.npz", **out)
    print("schedulers ok")


def pipeline_case(unet, vae):
    """Tiny SDXL pipeline, 4 Euler steps, fp32 CPU reference (pipeline_stable_diffusion_xl.py:823-1308)."""
    sch = EulerDiscreteScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1,
                                 timestep_spacing="leading")
    pipe = StableDiffusionXLPipeline(vae=vae, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None,
                                     unet=unet, scheduler=sch, force_zeros_for_empty_prompt=True)
    pipe.set_progress_bar_config(disable=True)
    g = gen(4321)
    pe = bf16_round(torch.randn((1, 7, 64), generator=g))
    ne = bf16_round(torch.randn((1, 7, 64), generator=g))
    pp = bf16_round(torch.randn((1, 64), generator=g))
    npp = bf16_round(torch.randn((1, 64), generator=g))
    lat = bf16_round(torch.randn((1, 4, 16, 16), generator=g))
    with torch.no_grad():
        res = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp,
                   negative_pooled_prompt_embeds=npp, latents=lat.clone(), num_inference_steps=4, guidance_scale=5.0,
                   height=128, width=128, original_size=(128, 128), target_size=(128, 128), output_type="latent")
        latents = res.images
        img = vae.decode(latents / vae.config.scaling_factor).sample
    np.savez_compressed(GOLD / "tiny_sdxl_pipeline.npz", prompt_embeds=pe.numpy(), negative_prompt_embeds=ne.numpy(),
                        pooled=pp.numpy(), negative_pooled=npp.numpy(), latents=lat.numpy(),
                        final_latents=latents.numpy(), image=img.numpy())
    print("pipeline ok; latents rms", float(latents.pow(2).mean().sqrt()), "img rms", float(img.pow(2).mean().sqrt()))


def main():
    torch.set_grad_enabled(False)
    print("reference diffusers", diffusers.__version__, "torch", torch.__version__)
    unet_xl = unet_case("tiny_unet_sdxl", dinit.TINY_SDXL_UNET, with_added=True)
    unet_case("tiny_unet_sd15", dinit.TINY_SD15_UNET, with_added=False)
    vae = vae_case("tiny_vae", dinit.TINY_VAE)
    scheduler_cases()
    pipeline_case(unet_xl, vae)
    # structural check of the full-size inventories against the reference constructors (meta device: no memory)
    with torch.device("meta"):
        for cfg in (dinit.SDXL_UNET, dinit.SD15_UNET):
            m = UNet2DConditionModel(**cfg)
            assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == dict(dinit.unet_param_shapes(dict(m.config)))
        for cfg in (dinit.SDXL_VAE,):
            m = AutoencoderKL(**cfg)
            ref_shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()
                          if k.startswith(("decoder.", "post_quant_conv."))}
            assert ref_shapes == dict(dinit.vae_decoder_param_shapes(dict(m.config)))
    print("full-size inventories match the reference constructors")


if __name__ == "__main__":
    main()
