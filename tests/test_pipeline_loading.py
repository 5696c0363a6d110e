"""Pipeline-level from_pretrained (diffusers_amd/pipeline_loading.py): a local reference pipeline directory (model_index.json + one
sub-folder per component, pipelines/pipeline_utils.py:739-1140) -> an engine pipeline.  CPU: the models run on the torch stand-ins
of the kernels (tests/ops_emulation.py), which is enough to check what is loaded from where and that the pipeline runs."""
import json

import pytest
import torch

from diffusers_amd import factory, init as dinit, loading
from diffusers_amd.pipelines import DDPMPipeline, StableDiffusionPipeline, StableDiffusionXLPipeline
from diffusers_amd.schedulers import DDIMScheduler, EulerDiscreteScheduler
from diffusers_amd.unet_2d_condition import UNet2DConditionModel

import ops_emulation
from diffusers_amd import ops

bf16 = torch.bfloat16


@pytest.fixture(autouse=True)
def _emulated_kernels(monkeypatch):
    ops_emulation.install(monkeypatch, ops)
    monkeypatch.setattr(ops, "TUNING", False)


def _sdxl_dir(tmp_path, scheduler_class="EulerDiscreteScheduler", class_name="StableDiffusionXLPipeline", extra=None):
    root = tmp_path / "pipe"
    usd = dinit.random_state_dict(dinit.unet_param_shapes(UNet2DConditionModel(**dinit.TINY_SDXL_UNET).config), seed=0)
    loading.save_reference_checkpoint(usd, dict(dinit.TINY_SDXL_UNET, _class_name="UNet2DConditionModel"), root / "unet")
    from diffusers_amd.autoencoder_kl import AutoencoderKL
    vsd = dinit.random_state_dict(dinit.vae_decoder_param_shapes(AutoencoderKL(**dinit.TINY_VAE).config), seed=1)
    loading.save_reference_checkpoint(vsd, dict(dinit.TINY_VAE, _class_name="AutoencoderKL"), root / "vae")
    (root / "scheduler").mkdir(parents=True)
    (root / "scheduler" / "scheduler_config.json").write_text(json.dumps(dict(factory.SDXL_SCHEDULER, _class_name=scheduler_class,
                                                                               _diffusers_version="0.40.0")))
    index = {"_class_name": class_name, "_diffusers_version": "0.40.0", "force_zeros_for_empty_prompt": True,
             "unet": ["diffusers", "UNet2DConditionModel"], "vae": ["diffusers", "AutoencoderKL"],
             "scheduler": ["diffusers", scheduler_class], "text_encoder": [None, None], "text_encoder_2": [None, None],
             "tokenizer": [None, None], "tokenizer_2": [None, None], "feature_extractor": [None, None], "image_encoder": [None, None]}
    index.update(extra or {})
    (root / "model_index.json").write_text(json.dumps(index))
    return root


def _inputs():
    g = torch.Generator().manual_seed(3)
    return dict(prompt_embeds=torch.randn((1, 77, 64), generator=g).to(bf16), negative_prompt_embeds=torch.randn((1, 77, 64), generator=g).to(bf16),
                pooled_prompt_embeds=torch.randn((1, 64), generator=g).to(bf16),
                negative_pooled_prompt_embeds=torch.randn((1, 64), generator=g).to(bf16),
                latents=torch.randn((1, 4, 16, 16), generator=g).to(bf16), num_inference_steps=3, guidance_scale=5.0, height=128,
                width=128, use_graph=False, output_type="latent")


def test_sdxl_pipeline_from_a_local_pipeline_directory(tmp_path):
    root = _sdxl_dir(tmp_path)
    pipe = StableDiffusionXLPipeline.from_pretrained(root, device="cpu", torch_dtype=torch.bfloat16)
    assert isinstance(pipe.unet, UNet2DConditionModel) and isinstance(pipe.scheduler, EulerDiscreteScheduler)
    assert pipe.scheduler.config.beta_schedule == "scaled_linear" and pipe.text_encoder is None and pipe.tokenizer_2 is None
    assert set(pipe.components) >= {"vae", "unet", "scheduler", "text_encoder", "tokenizer"} and pipe.to(torch.bfloat16) is pipe
    with pytest.raises(ValueError, match="no CPU path"):        # pipe.to() goes to the models' own .to(): bf16 on a HIP device only
        pipe.to("cpu")
    want = factory.build_sdxl_pipeline(device="cpu", tiny=True, seed=0)          # the same seeded weights, assembled by hand
    kw = _inputs()
    a = pipe(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()}).images
    b = want(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()}).images
    assert torch.equal(a, b)
    # the packed cache of each model was written next to its checkpoint and is what a second load reads
    assert list((root / "unet" / loading.PACKED_DIR).glob("*.safetensors")) and list((root / "vae" / loading.PACKED_DIR).glob("*.safetensors"))
    again = StableDiffusionXLPipeline.from_pretrained(root, device="cpu")
    assert torch.equal(again(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()}).images, a)
    # a component handed in replaces the directory's, as in the reference
    sch = DDIMScheduler(**factory.SD15_SCHEDULER)
    assert StableDiffusionXLPipeline.from_pretrained(root, device="cpu", scheduler=sch).scheduler is sch
    with pytest.raises(TypeError, match="unexpected components"):
        StableDiffusionXLPipeline.from_pretrained(root, device="cpu", controlnet=object())
    with pytest.raises(ValueError, match="bfloat16"):
        StableDiffusionXLPipeline.from_pretrained(root, device="cpu", torch_dtype=torch.float16)


def test_pipeline_directory_refusals(tmp_path):
    # a sampler the engine does not have must be passed in: nothing is substituted silently
    root = _sdxl_dir(tmp_path, scheduler_class="PNDMScheduler")
    with pytest.raises(NotImplementedError, match="PNDMScheduler"):
        StableDiffusionXLPipeline.from_pretrained(root, device="cpu")
    pipe = StableDiffusionXLPipeline.from_pretrained(root, device="cpu", scheduler=EulerDiscreteScheduler(**factory.SDXL_SCHEDULER))
    assert isinstance(pipe.scheduler, EulerDiscreteScheduler)
    # the directory names another pipeline class
    with pytest.raises(ValueError, match="StableDiffusionXLPipeline"):
        StableDiffusionPipeline.from_pretrained(root, device="cpu")
    with pytest.raises(ValueError, match="does not implement"):
        _sdxl_dir(tmp_path / "x", class_name="KandinskyPipeline") and StableDiffusionXLPipeline.from_pretrained(tmp_path / "x" / "pipe", device="cpu")
    with pytest.raises(FileNotFoundError, match="model_index.json"):
        StableDiffusionXLPipeline.from_pretrained(tmp_path / "nowhere", device="cpu")
    # a component slot the pipeline class does not have
    root2 = _sdxl_dir(tmp_path / "y", extra={"controlnet": ["diffusers", "ControlNetModel"]})
    with pytest.raises(ValueError, match="no component slot 'controlnet'"):
        StableDiffusionXLPipeline.from_pretrained(root2, device="cpu")


def test_ddpm_pipeline_from_a_local_pipeline_directory(tmp_path):
    from diffusers_amd.unet_2d import UNet2DModel
    root = tmp_path / "ddpm"
    sd = dinit.random_state_dict(dinit.unet2d_param_shapes(UNet2DModel(**dinit.TINY_DDPM).config), seed=11)
    loading.save_reference_checkpoint(sd, dict(dinit.TINY_DDPM, _class_name="UNet2DModel"), root / "unet")
    (root / "scheduler").mkdir(parents=True)
    (root / "scheduler" / "scheduler_config.json").write_text(json.dumps({"_class_name": "DDPMScheduler", "num_train_timesteps": 1000,
                                                                          "beta_schedule": "linear", "variance_type": "fixed_small"}))
    (root / "model_index.json").write_text(json.dumps({"_class_name": "DDPMPipeline", "unet": ["diffusers", "UNet2DModel"],
                                                       "scheduler": ["diffusers", "DDPMScheduler"]}))
    pipe = DDPMPipeline.from_pretrained(root, device="cpu")
    want = factory.build_ddpm_pipeline(device="cpu", tiny=True, seed=11)
    a = pipe(batch_size=1, generator=torch.Generator().manual_seed(0), num_inference_steps=3, output_type="np", use_graph=False).images
    b = want(batch_size=1, generator=torch.Generator().manual_seed(0), num_inference_steps=3, output_type="np", use_graph=False).images
    assert (a == b).all()


def test_sd_pipeline_directory_with_transformers_text_encoder(tmp_path):
    """["transformers", ...] slots: the tokenizer and the text encoder the directory names are loaded through transformers itself (the
    engine takes the caller's modules there), `prompt=` then works as in the reference; text_encoders="none" leaves the slot empty."""
    transformers = pytest.importorskip("transformers")
    from test_text_encoding import _clip, _tokenizer
    from diffusers_amd.autoencoder_kl import AutoencoderKL
    root = tmp_path / "sd"
    cfg = dict(dinit.TINY_SD15_UNET)
    usd = dinit.random_state_dict(dinit.unet_param_shapes(UNet2DConditionModel(**cfg).config), seed=0)
    loading.save_reference_checkpoint(usd, dict(cfg, _class_name="UNet2DConditionModel"), root / "unet")
    vsd = dinit.random_state_dict(dinit.vae_decoder_param_shapes(AutoencoderKL(**dinit.TINY_VAE).config), seed=1)
    loading.save_reference_checkpoint(vsd, dict(dinit.TINY_VAE, _class_name="AutoencoderKL"), root / "vae")
    (root / "scheduler").mkdir(parents=True)
    (root / "scheduler" / "scheduler_config.json").write_text(json.dumps(dict(factory.SD15_SCHEDULER, _class_name="DDIMScheduler")))
    tok, nv = _tokenizer()
    tok.save_pretrained(str(root / "tokenizer"))
    _clip(nv, 64, seed=1).save_pretrained(str(root / "text_encoder"))
    (root / "model_index.json").write_text(json.dumps({
        "_class_name": "StableDiffusionPipeline", "unet": ["diffusers", "UNet2DConditionModel"], "vae": ["diffusers", "AutoencoderKL"],
        "scheduler": ["diffusers", "DDIMScheduler"], "text_encoder": ["transformers", "CLIPTextModel"],
        "tokenizer": ["transformers", "CLIPTokenizer"], "safety_checker": ["stable_diffusion", "StableDiffusionSafetyChecker"],
        "feature_extractor": ["transformers", "CLIPImageProcessor"], "requires_safety_checker": True}))
    pipe = StableDiffusionPipeline.from_pretrained(root, device="cpu")
    assert isinstance(pipe.text_encoder, transformers.CLIPTextModel) and isinstance(pipe.tokenizer, transformers.CLIPTokenizer)
    assert isinstance(pipe.scheduler, DDIMScheduler) and pipe.text_encoder.dtype == torch.bfloat16
    g = torch.Generator().manual_seed(5)
    lat = torch.randn((1, 4, 16, 16), generator=g).to(bf16)
    img = pipe(prompt="hello cat", negative_prompt="", latents=lat.clone(), num_inference_steps=2, guidance_scale=7.5, height=32, width=32,
               output_type="raw", use_graph=False).images
    assert img.shape == (1, 3, 32, 32) and torch.isfinite(img.float()).all()
    bare = StableDiffusionPipeline.from_pretrained(root, device="cpu", text_encoders="none")
    assert bare.text_encoder is None and isinstance(bare.tokenizer, transformers.CLIPTokenizer)
    with pytest.raises(ValueError, match="prompt_embeds"):
        bare(prompt="hello", latents=lat.clone(), num_inference_steps=2, height=32, width=32, use_graph=False)
