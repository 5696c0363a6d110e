"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Golden vectors for the Flux rows from the REAL reference
(huggingface/diffusers imported from /root/reference/src), CPU fp32, seeded weights / inputs.  Build container only:

    PYTHONPATH=/root/reference/src python oracle/make_golden_flux.py

The reference has no offline known-answer test for FluxTransformer2DModel / FlowMatchEuler ("parity unpinned" in
SURVEY.md 8c), so these live-reference outputs ARE the pin for the Flux oracle and engine."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, "/root/reference/src")

from diffusers import AutoencoderKL, FlowMatchEulerDiscreteScheduler, FluxPipeline, FluxTransformer2DModel  # noqa: E402

from diffusers_amd import init as dinit  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def image_ids(h, w):
    ids = torch.zeros(h, w, 3)
    ids[..., 1] += torch.arange(h)[:, None]
    ids[..., 2] += torch.arange(w)[None, :]
    return ids.reshape(h * w, 3)


def main():
    torch.set_grad_enabled(False)
    cfg = dinit.TINY_FLUX
    tr = FluxTransformer2DModel(**cfg).eval()
    shapes = dinit.flux_param_shapes(cfg)
    assert {k: tuple(v.shape) for k, v in tr.state_dict().items()} == dict(shapes)
    sd = dinit.random_state_dict(shapes, seed=5)
    tr.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)

    g = torch.Generator().manual_seed(2024)
    B, St, hh = 2, 16, 8
    hs = bf16_round(torch.randn((B, hh * hh, 64), generator=g))
    ehs = bf16_round(torch.randn((B, St, 64), generator=g))
    pooled = bf16_round(torch.randn((B, 64), generator=g))
    t = torch.tensor([0.75, 0.25])
    img_ids, txt_ids = image_ids(hh, hh), torch.zeros(St, 3)
    y = tr(hidden_states=hs, encoder_hidden_states=ehs, pooled_projections=pooled, timestep=t, img_ids=img_ids,
           txt_ids=txt_ids, return_dict=False)[0]
    np.savez_compressed(GOLD / "tiny_flux.npz", hidden_states=hs.numpy(), encoder_hidden_states=ehs.numpy(),
                        pooled=pooled.numpy(), timestep=t.numpy(), img_ids=img_ids.numpy(), txt_ids=txt_ids.numpy(),
                        out=y.numpy())
    print("tiny_flux out rms", float(y.pow(2).mean().sqrt()))

    # ---- pipeline: 4 FlowMatch-Euler steps (schnell protocol: guidance 0, static shift 1.0) + 16-channel VAE decode ----
    vcfg = dinit.TINY_FLUX_VAE
    vae = AutoencoderKL(**vcfg).eval()
    vshapes = dinit.vae_decoder_param_shapes(dict(vae.config))
    vsd = dinit.random_state_dict(vshapes, seed=6)
    full = {k: v.clone() for k, v in vae.state_dict().items()}
    full.update({k: v.float() for k, v in vsd.items()})
    vae.load_state_dict(full, strict=True)
    sch = FlowMatchEulerDiscreteScheduler(shift=1.0, use_dynamic_shifting=False)
    pipe = FluxPipeline(scheduler=sch, vae=vae, text_encoder=None, tokenizer=None, text_encoder_2=None,
                        tokenizer_2=None, transformer=tr)
    pipe.set_progress_bar_config(disable=True)
    lat = bf16_round(torch.randn((1, hh * hh, 64), generator=g))
    pe = ehs[:1].clone()
    pp = pooled[:1].clone()
    size = 2 * hh * pipe.vae_scale_factor
    res = pipe(prompt_embeds=pe, pooled_prompt_embeds=pp, latents=lat.clone(), num_inference_steps=4, guidance_scale=0.0,
               height=size, width=size, output_type="latent", max_sequence_length=St)
    latents = res.images
    unp = pipe._unpack_latents(latents, size, size, pipe.vae_scale_factor)
    img = vae.decode(unp / vae.config.scaling_factor + vae.config.shift_factor).sample
    np.savez_compressed(GOLD / "tiny_flux_pipeline.npz", prompt_embeds=pe.numpy(), pooled=pp.numpy(), latents=lat.numpy(),
                        final_latents=latents.numpy(), image=img.numpy(), height=np.int32(size),
                        timesteps=sch.timesteps.numpy(), sigmas=sch.sigmas.numpy())
    print("tiny_flux_pipeline latents rms", float(latents.pow(2).mean().sqrt()), "image", tuple(img.shape),
          "rms", float(img.pow(2).mean().sqrt()), "timesteps", sch.timesteps.tolist())
    with torch.device("meta"):
        m = FluxTransformer2DModel(**dinit.FLUX_SCHNELL)
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == dict(dinit.flux_param_shapes(dinit.FLUX_SCHNELL))
        v = AutoencoderKL(**dinit.FLUX_VAE)
        ref_shapes = {k: tuple(x.shape) for k, x in v.state_dict().items() if k.startswith(("decoder.", "post_quant_conv."))}
        assert ref_shapes == dict(dinit.vae_decoder_param_shapes(dict(v.config)))
    print("full-size Flux inventories match the reference constructors")


if __name__ == "__main__":
    main()
