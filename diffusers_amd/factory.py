"""Build engine models / pipelines from canonical configs with seeded random weights (no checkpoints exist offline)."""
from __future__ import annotations

from typing import Optional

from . import init as dinit
from .autoencoder_kl import AutoencoderKL
from .autoencoder_kl_wan import AutoencoderKLWan
from .pipelines import DDPMPipeline, FluxPipeline, StableDiffusionPipeline, StableDiffusionXLPipeline, WanPipeline
from .schedulers import DDPMScheduler
from .unet_2d import UNet2DModel
from .transformer_wan import WanTransformer3DModel
from .schedulers import DDIMScheduler, EulerDiscreteScheduler, FlowMatchEulerDiscreteScheduler
from .transformer_flux import FluxTransformer2DModel
from .unet_2d_condition import UNet2DConditionModel

SDXL_SCHEDULER = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1,
                      timestep_spacing="leading")
SD15_SCHEDULER = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                      set_alpha_to_one=False, steps_offset=1)


def build_unet(cfg: dict, seed: int = 0, device="cuda", init_device: Optional[str] = None, state_dict=None):
    unet = UNet2DConditionModel(**cfg)
    if state_dict is None:
        shapes = dinit.unet_param_shapes(unet.config)
        state_dict = dinit.random_state_dict(shapes, seed=seed, device=init_device or "cpu")
    unet.load_state_dict(state_dict, device=device)
    return unet, state_dict


def build_vae(cfg: dict, seed: int = 1, device="cuda", init_device: Optional[str] = None, state_dict=None):
    vae = AutoencoderKL(**cfg)
    if state_dict is None:
        shapes = dinit.vae_decoder_param_shapes(vae.config)
        state_dict = dinit.random_state_dict(shapes, seed=seed, device=init_device or "cpu")
    vae.load_state_dict(state_dict, device=device)
    return vae, state_dict


def build_wan_vae(cfg: dict, seed: int = 21, device="cuda", init_device: Optional[str] = None, state_dict=None):
    vae = AutoencoderKLWan(**cfg)
    if state_dict is None:
        shapes = dinit.wan_vae_decoder_param_shapes(vae.config)
        state_dict = dinit.random_state_dict(shapes, seed=seed, device=init_device or "cpu")
    vae.load_state_dict(state_dict, device=device)
    return vae, state_dict


def build_sdxl_pipeline(device="cuda", tiny: bool = False, seed: int = 0, init_device: Optional[str] = None):
    """SDXL-base (BASELINE config 3) or its tiny sibling.  Full size: weights are generated on ``init_device``
    (default: the HIP device, ~2.6 B parameters in seconds) and freed after packing."""
    ucfg = dinit.TINY_SDXL_UNET if tiny else dinit.SDXL_UNET
    vcfg = dinit.TINY_VAE if tiny else dinit.SDXL_VAE
    idev = init_device or ("cpu" if tiny else str(device))
    unet, _ = build_unet(ucfg, seed=seed, device=device, init_device=idev)
    vae, _ = build_vae(vcfg, seed=seed + 1, device=device, init_device=idev)
    sch = EulerDiscreteScheduler(**SDXL_SCHEDULER)
    return StableDiffusionXLPipeline(vae=vae, unet=unet, scheduler=sch)


def build_sd15_pipeline(device="cuda", tiny: bool = False, seed: int = 0, init_device: Optional[str] = None):
    ucfg = dinit.TINY_SD15_UNET if tiny else dinit.SD15_UNET
    vcfg = dinit.TINY_VAE if tiny else dinit.SD_VAE
    idev = init_device or ("cpu" if tiny else str(device))
    unet, _ = build_unet(ucfg, seed=seed, device=device, init_device=idev)
    vae, _ = build_vae(vcfg, seed=seed + 1, device=device, init_device=idev)
    return StableDiffusionPipeline(vae=vae, unet=unet, scheduler=DDIMScheduler(**SD15_SCHEDULER))


def build_flux_transformer(cfg: dict, seed: int = 5, device="cuda", init_device: Optional[str] = None, state_dict=None):
    tr = FluxTransformer2DModel(**cfg)
    if state_dict is None:
        state_dict = dinit.random_state_dict(dinit.flux_param_shapes(tr.config), seed=seed, device=init_device or "cpu")
    tr.load_state_dict(state_dict, device=device)
    return tr, state_dict


def build_flux_pipeline(device="cuda", tiny: bool = False, seed: int = 5, init_device: Optional[str] = None):
    """FLUX.1-schnell (BASELINE config 4) or its tiny sibling: transformer + 16-channel VAE + FlowMatch-Euler (shift 1)."""
    tcfg = dinit.TINY_FLUX if tiny else dinit.FLUX_SCHNELL
    vcfg = dinit.TINY_FLUX_VAE if tiny else dinit.FLUX_VAE
    idev = init_device or ("cpu" if tiny else str(device))
    tr, _ = build_flux_transformer(tcfg, seed=seed, device=device, init_device=idev)
    vae, _ = build_vae(vcfg, seed=seed + 1, device=device, init_device=idev)
    sch = FlowMatchEulerDiscreteScheduler(shift=1.0, use_dynamic_shifting=False)
    return FluxPipeline(scheduler=sch, vae=vae, transformer=tr)


def build_wan_transformer(cfg: dict, seed: int = 9, device="cuda", init_device: Optional[str] = None, state_dict=None):
    tr = WanTransformer3DModel(**cfg)
    if state_dict is None:
        state_dict = dinit.random_state_dict(dinit.wan_param_shapes(tr.config), seed=seed, device=init_device or "cpu")
    tr.load_state_dict(state_dict, device=device)
    return tr, state_dict


def build_wan_pipeline(device="cuda", tiny: bool = False, seed: int = 9, init_device: Optional[str] = None,
                       flow_shift: float = 3.0, with_vae: bool = False, scheduler: str = "flowmatch"):
    """Wan2.1-T2V-1.3B (BASELINE config 5) or its tiny sibling.  ``scheduler="unipc"``: the sampler the checkpoint ships
    (pipeline_wan.py:52-59: UniPCMultistepScheduler, flow_prediction, use_flow_sigmas, flow_shift 3.0 for 480p), else the
    FlowMatch-Euler scheduler of SURVEY.md 8d; ``with_vae`` adds AutoencoderKLWan (seed + 12) so that
    ``output_type="raw" / "pt"`` decodes the video."""
    cfg = dinit.TINY_WAN if tiny else dinit.WAN_1_3B
    idev = init_device or ("cpu" if tiny else str(device))
    tr, _ = build_wan_transformer(cfg, seed=seed, device=device, init_device=idev)
    if scheduler == "unipc":
        from .schedulers import UniPCMultistepScheduler
        sch = UniPCMultistepScheduler(prediction_type="flow_prediction", use_flow_sigmas=True, flow_shift=flow_shift)
    elif scheduler == "flowmatch":
        sch = FlowMatchEulerDiscreteScheduler(shift=flow_shift, use_dynamic_shifting=False)
    else:
        raise ValueError("scheduler must be 'unipc' or 'flowmatch'")
    vae = None
    if with_vae:
        vae, _ = build_wan_vae(dinit.TINY_WAN_VAE if tiny else dinit.WAN_VAE, seed=seed + 12, device=device, init_device=idev)
    return WanPipeline(scheduler=sch, transformer=tr, vae=vae)


def build_unet2d(cfg: dict, seed: int = 0, device="cuda", init_device: Optional[str] = None, state_dict=None):
    unet = UNet2DModel(**cfg)
    if state_dict is None:
        full = dict(unet.config)
        state_dict = dinit.random_state_dict(dinit.unet2d_param_shapes(full), seed=seed, device=init_device or "cpu")
    unet.load_state_dict(state_dict, device=device)
    return unet, state_dict


def build_ddpm_pipeline(device="cuda", tiny: bool = False, seed: int = 0, init_device: Optional[str] = None):
    """google/ddpm-cat-256 (BASELINE config 1) or its tiny sibling."""
    cfg = dinit.TINY_DDPM if tiny else dinit.DDPM_CAT
    unet, _ = build_unet2d(cfg, seed=seed, device=device, init_device=init_device or ("cpu" if tiny else str(device)))
    return DDPMPipeline(unet=unet, scheduler=DDPMScheduler(**dinit.DDPM_SCHEDULER))
