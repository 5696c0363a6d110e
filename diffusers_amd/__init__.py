"""diffusers_amd -- MI355X (gfx950) native denoising engine behind the huggingface/diffusers operator surface.

The public names mirror the reference's (same constructor kwargs, ``state_dict`` keys and call signatures); every tensor
op underneath is a hand-written HIP kernel reached through the C ABI of ``include/diffusers_amd.h``.  Names resolve
lazily so that ``import diffusers_amd`` stays cheap and works on a host without the built library (the host-side
schedule/packing logic is importable everywhere; any compute call without ``libdiffusers_amd.so`` raises).
"""
from __future__ import annotations

import importlib

__version__ = "0.1.0"

_EXPORTS = {
    "UNet2DConditionModel": "unet_2d_condition",
    "AutoencoderKL": "autoencoder_kl",
    "AutoencoderKLWan": "autoencoder_kl_wan",
    "FluxTransformer2DModel": "transformer_flux",
    "FluxPipeline": "pipelines",
    "WanTransformer3DModel": "transformer_wan",
    "WanPipeline": "pipelines",
    "UNet2DModel": "unet_2d",
    "DDPMPipeline": "pipelines",
    "EulerDiscreteScheduler": "schedulers",
    "DDIMScheduler": "schedulers",
    "DDPMScheduler": "schedulers",
    "FlowMatchEulerDiscreteScheduler": "schedulers",
    "UniPCMultistepScheduler": "schedulers",
    "StableDiffusionPipeline": "pipelines",
    "StableDiffusionXLPipeline": "pipelines",
    "from_reference_config": "config_utils",
    "CLIPTextModel": "text_encoders",
    "CLIPTextModelWithProjection": "text_encoders",
    "T5EncoderModel": "text_encoders",
    "UMT5EncoderModel": "text_encoders",
}

__all__ = sorted(_EXPORTS)


def __getattr__(name):
    mod = _EXPORTS.get(name)
    if mod is None:
        raise AttributeError(f"module 'diffusers_amd' has no attribute {name!r}")
    return getattr(importlib.import_module(f".{mod}", __name__), name)
