"""The pipeline surface added in round 4 on hardware: pipeline-level from_pretrained over a local pipeline directory, custom Euler
schedules / denoising_end / num_images_per_prompt of the SDXL pipeline -- each through the captured HIP graph against the eager loop
(bit-identical: the same launches either way)."""
import json

import pytest
import torch

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16


def _rand(shape, seed):
    return torch.randn(shape, generator=torch.Generator("cpu").manual_seed(seed)).to(bf16).to("cuda")


def _kw():
    return dict(prompt_embeds=_rand((1, 77, 64), 1), negative_prompt_embeds=_rand((1, 77, 64), 2), pooled_prompt_embeds=_rand((1, 64), 3),
                negative_pooled_prompt_embeds=_rand((1, 64), 4), guidance_scale=5.0, height=128, width=128, output_type="latent")


def test_pipeline_from_pretrained_directory_on_the_gpu(tmp_path):
    from diffusers_amd import factory, init as dinit, loading
    from diffusers_amd.autoencoder_kl import AutoencoderKL
    from diffusers_amd.pipelines import StableDiffusionXLPipeline
    from diffusers_amd.unet_2d_condition import UNet2DConditionModel
    root = tmp_path / "pipe"
    usd = dinit.random_state_dict(dinit.unet_param_shapes(UNet2DConditionModel(**dinit.TINY_SDXL_UNET).config), seed=0)
    loading.save_reference_checkpoint(usd, dict(dinit.TINY_SDXL_UNET, _class_name="UNet2DConditionModel"), root / "unet")
    vsd = dinit.random_state_dict(dinit.vae_decoder_param_shapes(AutoencoderKL(**dinit.TINY_VAE).config), seed=1)
    loading.save_reference_checkpoint(vsd, dict(dinit.TINY_VAE, _class_name="AutoencoderKL"), root / "vae")
    (root / "scheduler").mkdir(parents=True)
    (root / "scheduler" / "scheduler_config.json").write_text(json.dumps(dict(factory.SDXL_SCHEDULER, _class_name="EulerDiscreteScheduler")))
    (root / "model_index.json").write_text(json.dumps({
        "_class_name": "StableDiffusionXLPipeline", "force_zeros_for_empty_prompt": True, "unet": ["diffusers", "UNet2DConditionModel"],
        "vae": ["diffusers", "AutoencoderKL"], "scheduler": ["diffusers", "EulerDiscreteScheduler"], "text_encoder": [None, None],
        "text_encoder_2": [None, None], "tokenizer": [None, None], "tokenizer_2": [None, None]}))
    pipe = StableDiffusionXLPipeline.from_pretrained(root, torch_dtype=torch.bfloat16).to("cuda")
    want = factory.build_sdxl_pipeline(device="cuda", tiny=True, seed=0)
    lat = _rand((1, 4, 16, 16), 5)
    kw = dict(_kw(), num_inference_steps=3, output_type="pt")
    a = pipe(latents=lat.clone(), **kw).images
    b = want(latents=lat.clone(), **kw).images
    assert torch.equal(a, b)
    again = StableDiffusionXLPipeline.from_pretrained(root)                      # second load: from the packed cache
    assert torch.equal(again(latents=lat.clone(), **kw).images, a)


def test_sdxl_custom_schedules_denoising_end_and_batching_through_the_graph():
    from diffusers_amd import factory
    pipe = factory.build_sdxl_pipeline(device="cuda", tiny=True, seed=0)
    lat = _rand((1, 4, 16, 16), 5)

    def both(latents=None, **extra):
        x = lat if latents is None else latents
        outs = [pipe(latents=x.clone(), use_graph=g, **dict(_kw(), **extra)).images.clone() for g in (False, True)]
        assert torch.equal(outs[0], outs[1]), extra
        return outs[0]
    full = both(num_inference_steps=4)
    ts = both(timesteps=[901.0, 601.0, 301.0, 1.0])
    assert pipe.scheduler.timesteps.tolist() == [901.0, 601.0, 301.0, 1.0] and not torch.equal(ts, full)
    sg = both(sigmas=[14.6, 5.0, 1.5, 0.4, 0.0])
    assert torch.isfinite(sg.float()).all() and not torch.equal(sg, ts)
    seen = []
    pipe(latents=lat.clone(), use_graph=True, num_inference_steps=4,
         callback_on_step_end=lambda p, i, t, k: seen.append(k["latents"].clone()) or k, **_kw())
    cut = both(num_inference_steps=4, denoising_end=0.5)
    assert pipe.scheduler.step_index == 2 and torch.equal(cut, seen[1])
    two = both(num_inference_steps=4, num_images_per_prompt=2, latents=torch.cat([lat, lat * 0.5]))
    assert two.shape[0] == 2 and torch.equal(two[:1], full)
