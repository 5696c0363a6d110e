"""Denoising pipelines on the HIP engine.

``StableDiffusionXLPipeline`` / ``StableDiffusionPipeline`` keep the reference ``__call__`` surface for the path the
BASELINE measures (pre-computed prompt embeddings + latents in, image out; reference:
pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl.py:823-1308 and
pipelines/stable_diffusion/pipeline_stable_diffusion.py:772-1107).  Text encoders / tokenizers are out of scope
(SURVEY.md 8f rank 3), so ``prompt=`` raises and ``prompt_embeds=`` is required.

The denoising loop body -- scale_model_input + CFG batch doubling, UNet forward, CFG combine + scheduler.step -- is
captured ONCE into a HIP graph (torch.cuda.CUDAGraph records the kernels our C ABI launches on the current stream) and
replayed for every step: per-step scalars come from the scheduler's device table indexed by a device step counter.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import ops
from .autoencoder_kl import AutoencoderKL
from .schedulers import DDIMScheduler, EulerDiscreteScheduler
from .unet_2d_condition import UNet2DConditionModel

bf16 = torch.bfloat16


@dataclass
class PipelineOutput:
    images: torch.Tensor


class _LatentDiffusionBase:
    def __init__(self, vae: AutoencoderKL, unet: UNet2DConditionModel, scheduler):
        self.vae, self.unet, self.scheduler = vae, unet, scheduler
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1) if vae is not None else 8
        self._graph = None
        self._graph_key = None
        self._static = {}

    @property
    def device(self):
        return self.unet.device

    def set_progress_bar_config(self, **kw):
        pass

    # ---- the captured step -------------------------------------------------------------------------------------
    def _step(self, latents, cond, guidance_scale, do_cfg):
        sch = self.scheduler
        rep = 2 if do_cfg else 1
        if isinstance(sch, EulerDiscreteScheduler):
            x_in = sch.scale_model_input(latents, sch.timesteps[0], rep=rep)
        else:
            x_in = torch.cat([latents] * rep) if rep > 1 else latents  # DDIM: scale_model_input is the identity
        eps = self.unet(x_in, None, None, conditioning=cond, sampler_table=sch.device_table,
                        step_idx=sch.device_step, return_dict=False)[0]
        if do_cfg:
            sch.step_cfg(eps, latents, guidance_scale, out=latents)  # in place: same buffer every replay
        else:
            raise NotImplementedError("guidance_scale <= 1 (no CFG) path")
        return latents

    def _denoise(self, latents, cond, num_steps, guidance_scale, do_cfg, use_graph):
        sch = self.scheduler
        sch.reset(0)
        if not use_graph:
            for _ in range(num_steps):
                self._step(latents, cond, guidance_scale, do_cfg)
            return latents
        key = (tuple(latents.shape), float(guidance_scale), cond["kvs"][0][0].skv if cond["kvs"] else 0,
               sch.device_table.data_ptr(), sch.device_step.data_ptr())
        if self._graph is None or self._graph_key != key:
            # warm-up on a side stream (lazy one-time driver calls must not happen during capture), then capture
            saved = latents.clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._step(latents, cond, guidance_scale, do_cfg)
            torch.cuda.current_stream().wait_stream(s)
            latents.copy_(saved)
            sch.reset(0)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step(latents, cond, guidance_scale, do_cfg)
            self._graph, self._graph_key = g, key
            self._static = {"latents": latents, "cond": cond}
            latents.copy_(saved)
            sch.reset(0)
        else:
            # same shapes: refresh the graph's static inputs (device-to-device copies, no re-capture)
            self._static["latents"].copy_(latents)
            latents = self._static["latents"]
            old = self._static["cond"]
            for kv_old, kv_new in zip(old["kvs"], cond["kvs"]):
                for a, b in zip(kv_old, kv_new):
                    a.k.copy_(b.k)
                    a.vt.copy_(b.vt)
            if old["aug_emb"] is not None:
                old["aug_emb"].copy_(cond["aug_emb"])
        for _ in range(num_steps):
            self._graph.replay()
        sch._step_index = num_steps
        return latents

    def _decode(self, latents, output_type):
        if output_type == "latent":
            return latents
        img = self.vae.decode(latents, return_dict=False, latents_div=float(self.vae.config.scaling_factor))[0]
        if output_type == "raw":
            return img
        if output_type == "pt":
            # VaeImageProcessor.postprocess (image_processor.py:738; out of hot-path scope): denormalise to [0,1]
            return (img.float() * 0.5 + 0.5).clamp(0, 1)
        raise ValueError(f"output_type={output_type!r}: use 'pt', 'raw' or 'latent'")


class StableDiffusionXLPipeline(_LatentDiffusionBase):
    def __init__(self, vae, unet, scheduler, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None,
                 force_zeros_for_empty_prompt: bool = True):
        super().__init__(vae, unet, scheduler)
        self.default_sample_size = unet.config.sample_size

    def _get_add_time_ids(self, original_size, crops_coords_top_left, target_size, text_encoder_projection_dim):
        add_time_ids = list(original_size + crops_coords_top_left + target_size)
        c = self.unet.config
        passed = c.addition_time_embed_dim * len(add_time_ids) + text_encoder_projection_dim
        expected = c.projection_class_embeddings_input_dim
        if expected != passed:
            raise ValueError(f"Model expects an added time embedding vector of length {expected}, but a vector of "
                             f"{passed} was created. The model has an incorrect config. Please check "
                             "`unet.config.time_embedding_type` and `text_encoder_2.config.projection_dim`.")
        return torch.tensor([add_time_ids], dtype=torch.float32)

    @torch.no_grad()
    def __call__(self, prompt=None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 5.0, latents: Optional[torch.Tensor] = None,
                 prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None,
                 negative_pooled_prompt_embeds=None, output_type: str = "pt", return_dict: bool = True,
                 original_size: Optional[Tuple[int, int]] = None, crops_coords_top_left: Tuple[int, int] = (0, 0),
                 target_size: Optional[Tuple[int, int]] = None, generator=None, use_graph: bool = True):
        if prompt is not None:
            raise ValueError("text encoders are outside the HIP hot path: pass `prompt_embeds`/`pooled_prompt_embeds`")
        if prompt_embeds is None or pooled_prompt_embeds is None:
            raise ValueError("Provide `prompt_embeds` and `pooled_prompt_embeds`.")
        do_cfg = guidance_scale > 1.0
        if do_cfg and (negative_prompt_embeds is None or negative_pooled_prompt_embeds is None):
            raise ValueError("classifier-free guidance needs `negative_prompt_embeds` and "
                             "`negative_pooled_prompt_embeds`")
        dev = self.device
        height = height or self.default_sample_size * self.vae_scale_factor
        width = width or self.default_sample_size * self.vae_scale_factor
        original_size = original_size or (height, width)
        target_size = target_size or (height, width)
        B = prompt_embeds.shape[0]
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        shape = (B, self.unet.config.in_channels, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            gdev = generator.device if generator is not None else torch.device("cpu")
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=bf16)
        # (like the reference's prepare_latents, user-supplied latents are taken as they are: no shape check,
        #  pipeline_stable_diffusion_xl.py:707-727)
        latents = latents.to(device=dev, dtype=bf16).contiguous()
        latents = ops.mul_scalar(latents, float(self.scheduler.init_noise_sigma))

        pe = prompt_embeds.to(device=dev, dtype=bf16)
        te = pooled_prompt_embeds.to(device=dev, dtype=bf16)
        ids = self._get_add_time_ids(original_size, crops_coords_top_left, target_size, te.shape[-1]).to(dev)
        ids = ids.repeat(B, 1)
        if do_cfg:
            pe = torch.cat([negative_prompt_embeds.to(device=dev, dtype=bf16), pe], dim=0)
            te = torch.cat([negative_pooled_prompt_embeds.to(device=dev, dtype=bf16), te], dim=0)
            ids = torch.cat([ids, ids], dim=0)
        cond = self.unet.precompute_conditioning(pe.contiguous(), {"text_embeds": te, "time_ids": ids})
        latents = self._denoise(latents, cond, num_inference_steps, guidance_scale, do_cfg, use_graph)
        images = self._decode(latents, output_type)
        if not return_dict:
            return (images,)
        return PipelineOutput(images=images)


class StableDiffusionPipeline(_LatentDiffusionBase):
    def __init__(self, vae, unet, scheduler, text_encoder=None, tokenizer=None, safety_checker=None,
                 feature_extractor=None, requires_safety_checker: bool = False):
        super().__init__(vae, unet, scheduler)

    @torch.no_grad()
    def __call__(self, prompt=None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, eta: float = 0.0,
                 latents: Optional[torch.Tensor] = None, prompt_embeds=None, negative_prompt_embeds=None,
                 output_type: str = "pt", return_dict: bool = True, generator=None, use_graph: bool = True):
        if prompt is not None:
            raise ValueError("text encoders are outside the HIP hot path: pass `prompt_embeds`")
        if prompt_embeds is None:
            raise ValueError("Provide `prompt_embeds`.")
        if eta != 0.0:
            raise NotImplementedError("eta > 0 in the fused pipeline loop")
        do_cfg = guidance_scale > 1.0
        if do_cfg and negative_prompt_embeds is None:
            raise ValueError("classifier-free guidance needs `negative_prompt_embeds`")
        dev = self.device
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        B = prompt_embeds.shape[0]
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        shape = (B, self.unet.config.in_channels, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            gdev = generator.device if generator is not None else torch.device("cpu")
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=bf16)
        latents = latents.to(device=dev, dtype=bf16).contiguous()
        latents = ops.mul_scalar(latents, float(self.scheduler.init_noise_sigma))
        pe = prompt_embeds.to(device=dev, dtype=bf16)
        if do_cfg:
            pe = torch.cat([negative_prompt_embeds.to(device=dev, dtype=bf16), pe], dim=0)
        cond = self.unet.precompute_conditioning(pe.contiguous(), None)
        latents = self._denoise(latents, cond, num_inference_steps, guidance_scale, do_cfg, use_graph)
        images = self._decode(latents, output_type)
        if not return_dict:
            return (images,)
        return PipelineOutput(images=images)
