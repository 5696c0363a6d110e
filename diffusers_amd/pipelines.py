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

import threading

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from . import ops
from .autoencoder_kl import AutoencoderKL
from .pipeline_loading import PipelineLoadingMixin
from .schedulers import (DDIMScheduler, DDPMScheduler, EulerDiscreteScheduler, FlowMatchEulerDiscreteScheduler,
                         UniPCMultistepScheduler)
from .transformer_flux import FluxTransformer2DModel
from .transformer_wan import WanTransformer3DModel
from .unet_2d import UNet2DModel
from .unet_2d_condition import UNet2DConditionModel

bf16 = torch.bfloat16

# Capture mode of the denoising-step graphs.  "thread_local": only THIS thread's calls are checked against the capture,
# so the RCCL watchdog thread of a multi-GPU job (which polls events while we capture) cannot invalidate it.
GRAPH_CAPTURE_MODE = "thread_local"


def postprocess_images(img: torch.Tensor, output_type: str):
    """VaeImageProcessor.postprocess / VideoProcessor.postprocess_video (image_processor.py:738-786) fused into one kernel
    pass over the decoded tensor: "raw" = the decoder output in [-1, 1]; "pt" = [0, 1] fp32, same layout; "np" = channels
    last numpy fp32; "pil" = list of PIL images (stills only) built from the kernel's uint8 bytes."""
    if output_type == "raw":
        return img
    if output_type == "pt":
        return ops.image_postprocess(img, "pt")
    if output_type == "np":
        return ops.image_postprocess(img, "np").cpu().numpy()
    if output_type == "pil":
        if img.dim() != 4:
            raise ValueError("output_type='pil' is for stills; use 'np' for video")
        from PIL import Image
        arr = ops.image_postprocess(img, "uint8").cpu().numpy()
        return [Image.fromarray(a.squeeze(-1), mode="L") if a.shape[-1] == 1 else Image.fromarray(a) for a in arr]
    raise ValueError(f"output_type={output_type!r}: use 'pil', 'np', 'pt', 'raw' or 'latent'")


_DECODE_LOCK = threading.Lock()


class _exclusive_decode:
    """Round 6, several pipelines in flight on one GPU (``STREAM_DOMAIN``).  The step graphs of several pipelines replay concurrently on
    several streams without touching each other (final latents bit-identical over every round tried, tools/debug_inflight.py --latent).
    Two EAGER VAE decodes running at the same time on two streams do NOT reproduce their one-at-a-time bits (tools/debug_decode_concurrent.py:
    most concurrent decodes differ, the first differing launch usually a conv / nn.Linear of the 1024 x 1024 level).  What was ruled out,
    each by its own run under tools/: out-of-bounds writes (64 KiB canaries around every allocation of a decode and of a U-Net step:
    intact), the caching allocator (per-thread bump arenas: same result), a host-side launch race (a lock around every C-ABI call: same),
    split-K / one-launch GroupNorm / prefetch hints / the four-pixel conv_in (switched off: same), any single op looped next to a
    decode (the decode stays intact), LDS writes outside a workgroup's allocation (tools/debug_lds_canary.py: canary workgroups of another
    kernel kept resident on every CU through eager decodes, U-Net steps, the eight-phase tiles and D = 128 attention stay intact -- and a
    positive control shows the hardware bounds plain ds_write AND LDS-DMA to the issuing workgroup's allocation in the first place,
    profiles/r06f_lds_canary.jsonl), scratch memory (three kernel instantiations use any, none of them in a decode);
    one hardware queue (GPU_MAX_HW_QUEUES=1) makes the difference disappear, and decodes replayed
    from their own HIP graphs reproduce their bits (48 of 48) -- but a graph-captured decode next to another pipeline's step graphs
    made things worse, not better.  The cause is NOT isolated.  Mitigation: a thread that set ``STREAM_DOMAIN.tag`` takes this lock around
    its decode and holds it until its stream has drained, so two decodes never overlap; with it 5 of 6 concurrent images reproduce their
    bits and the sixth differs by <= 2e-2 (a decode overlapping the OTHER pipeline's step replays).  Running several pipelines
    concurrently is therefore a MEASUREMENT in this repository (bench.py's informational `serving_two_in_flight` leg says whether its
    images were bit-identical), not a supported mode.  The default single-domain path takes no lock and no synchronisation."""

    def __enter__(self):
        self.on = getattr(STREAM_DOMAIN, "tag", 0) != 0
        if self.on:
            _DECODE_LOCK.acquire()
        return self

    def __exit__(self, *exc):
        if self.on:
            try:
                torch.cuda.current_stream().synchronize()
            finally:
                _DECODE_LOCK.release()
        return False


def decode_postprocessed(vae, latents: torch.Tensor, output_type: str, **decode_kw):
    """``vae.decode(latents / scaling_factor).sample`` followed by ``image_processor.postprocess(..., output_type)``
    (pipeline_stable_diffusion_xl.py:1283-1299) with the postprocess fused into the decoder's last pass."""
    with _exclusive_decode():
        return _decode_postprocessed(vae, latents, output_type, **decode_kw)


def _decode_postprocessed(vae, latents: torch.Tensor, output_type: str, **decode_kw):
    mode = {"pt": "pt", "np": "np", "pil": "uint8"}.get(output_type)
    if mode is None:
        return postprocess_images(vae.decode(latents, return_dict=False, **decode_kw)[0], output_type)
    out = vae.decode(latents, return_dict=False, postprocess=mode, **decode_kw)[0]
    if output_type == "pt":
        return out
    arr = out.cpu().numpy()
    if output_type == "np":
        return arr
    from PIL import Image
    return [Image.fromarray(a.squeeze(-1), mode="L") if a.shape[-1] == 1 else Image.fromarray(a) for a in arr]


def _per_prompt(t: Optional[torch.Tensor], n: int) -> Optional[torch.Tensor]:
    """``num_images_per_prompt`` copies of each row of caller-supplied embeddings, the copies of one prompt adjacent -- what the
    reference's SD / SDXL ``encode_prompt`` does to ``prompt_embeds`` it is handed (pipeline_stable_diffusion_xl.py:488-516:
    ``repeat(1, n, 1).view(bs * n, seq, -1)``; pooled: ``repeat(1, n).view(bs * n, -1)``).  The reference's Flux and Wan
    ``encode_prompt`` return supplied embeddings as they are and only multiply the latent batch (pipeline_flux.py:358-375,
    pipeline_wan.py:225-262) -- a combination that fails inside their transformer for n > 1; repeating them there as well is an
    ENGINE EXTENSION of those two pipelines, not reference behaviour."""
    if t is None or n == 1:
        return t
    if n < 1:
        raise ValueError("num_images_per_prompt must be >= 1")
    return t.repeat_interleave(n, dim=0)


def denoising_end_steps(scheduler, denoising_end) -> int:
    """Steps of the scheduler's current schedule that run when the loop stops at the fraction ``denoising_end`` of the training
    timesteps (pipeline_stable_diffusion_xl.py:1164-1183: base + refiner workflows): those whose timestep is at or above the cut-off."""
    n = len(scheduler.timesteps)
    if denoising_end is None or not isinstance(denoising_end, float) or not 0.0 < denoising_end < 1.0:
        return n
    n_train = scheduler.config.num_train_timesteps
    cutoff = int(round(n_train - denoising_end * n_train))
    return len([t for t in scheduler.timesteps.tolist() if t >= cutoff])


def _capture_step(pipe, step, mode):
    """What `_denoise` replays once per step: the captured HIP graph (``use_graph=True``), or -- ``use_graph="plan"`` -- the
    step's launch list owned by the C library (diffusers_amd/plan.py, include/diffusers_amd.h "launch plans": the same launches
    in the same order, issued by `da_plan_launch` instead of the graph executor; what a host without Python replays).  `step` has
    run once already (warm-up); the caller restores the latents and the step counter afterwards, as it does after a capture."""
    if mode == "plan":
        from . import plan as P, tuning
        if any(v[3] > 1 for v in tuning.table().values()):
            # split-K launches take their workspace per (device, stream); the warm-up ran on a side stream, and allocating (and
            # zeroing) this stream's inside the recording would be a torch operator the plan cannot replay
            ops.splitk_workspace(pipe.device, torch.cuda.current_stream().cuda_stream)
        if ops.ATTN_SPLIT:
            ops.attn_split_workspace(pipe.device, torch.cuda.current_stream().cuda_stream)
        if ops.GN_MULTI:
            ops.gn_sync_workspace(pipe.device, torch.cuda.current_stream().cuda_stream)
        with ops.weight_prefetch(_pf(pipe), "apply"):
            pl, _ = P.record(step)
        return pl
    g = torch.cuda.CUDAGraph()
    cs = _side_stream("capture")
    from . import tuning
    if ops.ATTN_SPLIT:     # the flash kernel's key-split workspace of the capture stream: allocated (counters zeroed) BEFORE the capture
        ops.attn_split_workspace(pipe.device, cs.cuda_stream)
    if ops.GN_MULTI:       # likewise the multi-workgroup GroupNorm's arrival counters
        ops.gn_sync_workspace(pipe.device, cs.cuda_stream)
    if any(len(v) > 3 and v[3] > 1 for v in tuning.table().values()):
        # split-K launches take their workspace per (device, stream): allocated (and its flags zeroed) for the capture stream BEFORE
        # the capture, so that neither the allocation lands in the graph's private pool nor the zero-fill becomes a graph node
        ops.splitk_workspace(pipe.device, cs.cuda_stream)
    with torch.cuda.graph(g, stream=cs, capture_error_mode=GRAPH_CAPTURE_MODE), ops.weight_prefetch(_pf(pipe), "apply"):
        step()
    return g


_side_streams = {}


STREAM_DOMAIN = threading.local()


def _side_stream(kind: str) -> "torch.cuda.Stream":
    """ONE warm-up stream and ONE capture stream per device for every pipeline of the process: per-stream resources (the 64 MiB
    split-K workspace of ops.splitk_workspace, the flash kernel's key-split workspace, the GroupNorm sync buffer) are then allocated
    twice, not once per re-capture.  Graphs captured on one stream bake in the SAME workspaces and must not replay concurrently; a host
    that replays several pipelines' graphs at the same time (one thread and stream per pipeline) sets ``STREAM_DOMAIN.tag`` to a
    distinct value in each thread BEFORE the pipeline's first call: each domain gets side streams -- hence workspaces -- of its own."""
    key = (torch.cuda.current_device(), kind, getattr(STREAM_DOMAIN, "tag", 0))
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream()
    return st


def _pf(pipe) -> "ops.WeightPrefetch":
    """The pipeline's weight-prefetch trace (ops.weight_prefetch): recorded by one eager step, applied to every later one."""
    pf = getattr(pipe, "_weight_prefetch", None)
    if pf is None:
        pf = pipe._weight_prefetch = ops.WeightPrefetch(owner=pipe, slots=("unet", "transformer"))
    return pf


class _StepCallbacks:
    """``callback_on_step_end`` of the reference pipelines (pipeline_stable_diffusion_xl.py:857-858, :1239-1247; pipeline_stable_diffusion.py:1064-1071,
    pipeline_flux.py:938-945, pipeline_wan.py:637-644): after every denoising step -- a replayed graph, a replayed plan or an eager step alike -- the
    callable gets ``(pipe, step_index, timestep, {"latents": latents})``; a ``"latents"`` entry in the dict it returns replaces the
    loop's latents (copied into the buffer the captured step reads), and setting ``pipe._interrupt = True`` ends the loop after
    the current step (``interrupt`` property, :817-819 / :1198).  ``latents`` is the only tensor the engine's loop can hand out or
    take back: the text conditioning is packed once before the loop (K / V^T projections hoisted out of it), so the reference's
    other names (``prompt_embeds``, ``add_text_embeds``, ...) are refused by name."""
    _callback_tensor_inputs = ["latents"]
    _interrupt = False
    _step_callback = None

    @property
    def interrupt(self):
        return self._interrupt

    def _arm_callback(self, callback_on_step_end, callback_on_step_end_tensor_inputs):
        self._interrupt = False
        names = callback_on_step_end_tensor_inputs
        if hasattr(callback_on_step_end, "tensor_inputs"):            # PipelineCallback / MultiPipelineCallbacks objects (:1041-1042)
            names = callback_on_step_end.tensor_inputs
        names = ["latents"] if names is None else list(names)
        bad = [k for k in names if k not in self._callback_tensor_inputs]
        if bad:
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}, but found {bad}")
        self._step_callback = (callback_on_step_end, names) if callback_on_step_end is not None else None

    def _after_step(self, i: int, latents: torch.Tensor) -> bool:
        """Runs the armed callback after step ``i``; False = the loop stops here."""
        if self._step_callback is None:
            return True
        fn, names = self._step_callback
        out = fn(self, i, self.scheduler.timesteps[i], {k: latents for k in names})
        if out is not None:
            new = out.pop("latents", latents)
            if new is not latents:
                latents.copy_(new.to(device=latents.device, dtype=latents.dtype))
        return not self._interrupt


@dataclass
class PipelineOutput:
    images: torch.Tensor


class _LatentDiffusionBase(_StepCallbacks, PipelineLoadingMixin):
    def __init__(self, vae: AutoencoderKL, unet: UNet2DConditionModel, scheduler):
        self.vae, self.unet, self.scheduler = vae, unet, scheduler
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1) if vae is not None else 8
        self._graph = None
        self._graph_key = None
        self._static = {}
        self._eta = 0.0            # DDIM only: eta > 0 adds pre-drawn variance noise (one row per step) in the fused step
        self._noise_table = None
        self._guidance_rescale = 0.0   # > 0: rescale_noise_cfg after the CFG combine (pipeline_stable_diffusion.py:69-92)

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
            x_in = ops.mul_scalar(latents, 1.0, rep=rep) if rep > 1 else latents  # DDIM: scale_model_input = identity
        eps = self.unet(x_in, None, None, conditioning=cond, sampler_table=sch.device_table,
                        step_idx=sch.device_step, return_dict=False)[0]
        # in place (same buffer every replay); without CFG (guidance_scale <= 1, pipeline_stable_diffusion_xl.py:1202,
        # :1223) the U-Net ran on the un-doubled batch and the same kernel skips the combine
        kw = {"eta": self._eta, "noise_table": self._noise_table} if self._eta > 0 else {}
        if do_cfg and self._guidance_rescale > 0.0:
            # pipeline_stable_diffusion_xl.py:1227-1229 / pipeline_stable_diffusion.py:1057-1059: combine, then rescale_noise_cfg
            # (per-sample std of the text and of the guided prediction: two small launches), then the step without its combine
            eps = ops.cfg_rescale(eps, guidance_scale, self._guidance_rescale)
            sch.step_cfg(eps, latents, guidance_scale, out=latents, cfg=False, **kw)
            return latents
        sch.step_cfg(eps, latents, guidance_scale, out=latents, cfg=do_cfg, **kw)
        return latents

    def _make_graph_key(self, latents, cond, guidance_scale, do_cfg):
        """Everything a captured step depends on besides the contents of its static buffers."""
        sch = self.scheduler
        if isinstance(sch, DDIMScheduler):
            # set_timesteps() invalidated the coefficient table; rebuild it for THIS call's eta before the key reads its
            # address (reading `device_table` first would rebuild it in place for eta = 0, and a replayed graph would
            # then run deterministic DDIM whatever eta the caller passed)
            sch._ensure(self._eta, sch.timesteps[0])
        return (tuple(latents.shape), float(guidance_scale), bool(do_cfg), cond["kvs"][0][0].skv if cond["kvs"] else 0,
                sch.device_table.data_ptr(), sch.device_step.data_ptr(),
                self._noise_table.data_ptr() if self._noise_table is not None else 0, float(self._eta),
                float(self._guidance_rescale), id(self.unet))       # (a captured step points into THIS model's packed weights)

    def _denoise(self, latents, cond, num_steps, guidance_scale, do_cfg, use_graph):
        sch = self.scheduler
        sch.reset(0)
        if not use_graph:
            for i in range(num_steps):
                with ops.weight_prefetch(_pf(self), "apply" if i else "record"):
                    self._step(latents, cond, guidance_scale, do_cfg)
                if not self._after_step(i, latents):
                    break
            return latents
        key = self._make_graph_key(latents, cond, guidance_scale, do_cfg) + (use_graph == "plan",)
        if self._graph is None or self._graph_key != key:
            # warm-up on a side stream (lazy one-time driver calls must not happen during capture), then capture
            saved = latents.clone()
            s = _side_stream("warm")
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                with ops.weight_prefetch(_pf(self), "record"):
                    self._step(latents, cond, guidance_scale, do_cfg)
            torch.cuda.current_stream().wait_stream(s)
            latents.copy_(saved)
            sch.reset(0)
            g = _capture_step(self, lambda: self._step(latents, cond, guidance_scale, do_cfg), use_graph)
            self._graph, self._graph_key = g, key
            self._static = {"latents": latents, "cond": cond, "noise_table": self._noise_table}
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
        done = 0
        for i in range(num_steps):
            self._graph.replay()
            done = i + 1
            if not self._after_step(i, latents):
                break
        sch._step_index = done
        return latents

    def _decode(self, latents, output_type):
        if output_type == "latent":
            return latents
        vc = self.vae.config
        if vc.get("latents_mean") is not None or vc.get("latents_std") is not None:
            # pipeline_stable_diffusion_xl.py:1267-1277 de-normalises per channel with these before decoding; the fused
            # decode entry takes one scalar divisor, so a VAE that ships them is refused rather than decoded wrongly
            raise NotImplementedError("AutoencoderKL configs with latents_mean / latents_std are not supported by the "
                                      "engine pipelines (decode the returned output_type='latent' tensor yourself)")
        return decode_postprocessed(self.vae, latents, output_type, latents_div=float(vc.scaling_factor))


class StableDiffusionXLPipeline(_LatentDiffusionBase):
    def __init__(self, vae, unet, scheduler, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None,
                 force_zeros_for_empty_prompt: bool = True):
        super().__init__(vae, unet, scheduler)
        self.default_sample_size = unet.config.sample_size
        # optional caller-side components (transformers modules): with them `prompt=` works as in the reference
        self.text_encoder, self.text_encoder_2 = text_encoder, text_encoder_2
        self.tokenizer, self.tokenizer_2 = tokenizer, tokenizer_2
        self.force_zeros_for_empty_prompt = force_zeros_for_empty_prompt

    def encode_prompt(self, prompt, prompt_2=None, device=None, num_images_per_prompt: int = 1,
                      do_classifier_free_guidance: bool = True, negative_prompt=None, negative_prompt_2=None,
                      clip_skip=None):
        """pipeline_stable_diffusion_xl.py:283-518 through the caller's CLIP encoders (text_encoding.encode_prompt_sdxl)."""
        from .text_encoding import encode_prompt_sdxl
        if self.tokenizer_2 is None or self.text_encoder_2 is None:
            raise ValueError("`prompt=` needs tokenizer_2 / text_encoder_2 (and optionally tokenizer / text_encoder); "
                             "without them pass `prompt_embeds` and `pooled_prompt_embeds`")
        toks = [self.tokenizer, self.tokenizer_2] if self.tokenizer is not None else [self.tokenizer_2]
        encs = [self.text_encoder, self.text_encoder_2] if self.text_encoder is not None else [self.text_encoder_2]
        return encode_prompt_sdxl(toks, encs, prompt, prompt_2, device or self.device, num_images_per_prompt,
                                  do_classifier_free_guidance, negative_prompt, negative_prompt_2,
                                  self.force_zeros_for_empty_prompt, clip_skip)

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
                 target_size: Optional[Tuple[int, int]] = None, generator=None, use_graph: bool = True,
                 prompt_2=None, negative_prompt=None, negative_prompt_2=None, num_images_per_prompt: int = 1,
                 clip_skip=None, guidance_rescale: float = 0.0, callback_on_step_end=None,
                 callback_on_step_end_tensor_inputs=None, timesteps=None, sigmas=None, denoising_end: Optional[float] = None):
        do_cfg = guidance_scale > 1.0
        self._guidance_rescale = float(guidance_rescale)    # pipeline_stable_diffusion_xl.py:849, :1227-1229
        if timesteps is not None and sigmas is not None:    # retrieve_timesteps, :142-143
            raise ValueError("Only one of `timesteps` or `sigmas` can be passed. Please choose one to set custom values")
        self._arm_callback(callback_on_step_end, callback_on_step_end_tensor_inputs)
        if prompt is not None:
            if prompt_embeds is not None:
                raise ValueError("Cannot forward both `prompt` and `prompt_embeds`. Please make sure to only forward one "
                                 "of the two.")
            prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds = \
                self.encode_prompt(prompt, prompt_2, self.device, num_images_per_prompt, do_cfg, negative_prompt,
                                   negative_prompt_2, clip_skip)
        else:
            prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds = (
                _per_prompt(t, num_images_per_prompt) for t in (prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds,
                                                                negative_pooled_prompt_embeds))
        if prompt_embeds is None or pooled_prompt_embeds is None:
            raise ValueError("Provide either `prompt` (with the text encoders given to the pipeline) or `prompt_embeds` "
                             "and `pooled_prompt_embeds`.")
        if do_cfg and (negative_prompt_embeds is None or negative_pooled_prompt_embeds is None):
            raise ValueError("classifier-free guidance needs `negative_prompt_embeds` and "
                             "`negative_pooled_prompt_embeds`")
        dev = self.device
        height = height or self.default_sample_size * self.vae_scale_factor
        width = width or self.default_sample_size * self.vae_scale_factor
        original_size = original_size or (height, width)
        target_size = target_size or (height, width)
        B = prompt_embeds.shape[0]
        # retrieve_timesteps (:144-167): a custom timestep / sigma schedule replaces the step count; denoising_end (:1164-1183) then
        # keeps the leading steps down to its cut-off (the loop below simply runs fewer replays of the same captured step)
        if timesteps is not None:
            self.scheduler.set_timesteps(timesteps=timesteps, device=dev)
        elif sigmas is not None:
            self.scheduler.set_timesteps(sigmas=sigmas, device=dev)
        else:
            self.scheduler.set_timesteps(num_inference_steps, device=dev)
        num_inference_steps = denoising_end_steps(self.scheduler, denoising_end)
        shape = (B, self.unet.config.in_channels, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            gdev = generator.device if generator is not None else torch.device("cpu")
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=bf16)
        # (like the reference's prepare_latents, user-supplied latents are taken as they are: no shape check,
        #  pipeline_stable_diffusion_xl.py:707-727)
        latents = latents.to(device=dev, dtype=bf16).contiguous()
        if latents.shape[0] != B:
            # (the reference's prepare_latents takes supplied latents unchecked and fails inside the U-Net; here a stale batch would
            #  meet a captured step of another size)
            raise ValueError(f"`latents` holds {latents.shape[0]} samples, the prompt embeddings (x num_images_per_prompt) {B}")
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
        if safety_checker is not None:
            raise NotImplementedError("the safety checker is outside this engine: post-filter the returned images")
        self.text_encoder, self.tokenizer = text_encoder, tokenizer   # optional caller-side transformers modules

    def encode_prompt(self, prompt, device=None, num_images_per_prompt: int = 1, do_classifier_free_guidance: bool = True,
                      negative_prompt=None, clip_skip=None):
        """pipeline_stable_diffusion.py:332-513 through the caller's CLIP encoder (text_encoding.encode_prompt_sd)."""
        from .text_encoding import encode_prompt_sd
        if self.tokenizer is None or self.text_encoder is None:
            raise ValueError("`prompt=` needs the pipeline's tokenizer / text_encoder; without them pass `prompt_embeds`")
        return encode_prompt_sd(self.tokenizer, self.text_encoder, prompt, device or self.device, num_images_per_prompt,
                                do_classifier_free_guidance, negative_prompt, clip_skip)

    @torch.no_grad()
    def __call__(self, prompt=None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, eta: float = 0.0,
                 latents: Optional[torch.Tensor] = None, prompt_embeds=None, negative_prompt_embeds=None,
                 output_type: str = "pt", return_dict: bool = True, generator=None, use_graph: bool = True,
                 negative_prompt=None, num_images_per_prompt: int = 1, clip_skip=None, guidance_rescale: float = 0.0,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=None):
        self._guidance_rescale = float(guidance_rescale)    # pipeline_stable_diffusion.py:1057-1059
        self._arm_callback(callback_on_step_end, callback_on_step_end_tensor_inputs)
        if eta < 0.0 or eta > 1.0:
            raise ValueError("eta (DDIM) must be in [0, 1]")
        do_cfg = guidance_scale > 1.0
        if prompt is not None:
            if prompt_embeds is not None:
                raise ValueError("Cannot forward both `prompt` and `prompt_embeds`. Please make sure to only forward one "
                                 "of the two.")
            prompt_embeds, negative_prompt_embeds = self.encode_prompt(prompt, self.device, num_images_per_prompt, do_cfg,
                                                                        negative_prompt, clip_skip)
        else:
            prompt_embeds, negative_prompt_embeds = (_per_prompt(t, num_images_per_prompt) for t in (prompt_embeds, negative_prompt_embeds))
        if prompt_embeds is None:
            raise ValueError("Provide either `prompt` (with the text encoder given to the pipeline) or `prompt_embeds`.")
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
        if latents.shape[0] != B:
            # (the reference's prepare_latents takes supplied latents unchecked and fails inside the U-Net; here a stale batch would
            #  meet a captured step of another size)
            raise ValueError(f"`latents` holds {latents.shape[0]} samples, the prompt embeddings (x num_images_per_prompt) {B}")
        latents = ops.mul_scalar(latents, float(self.scheduler.init_noise_sigma))
        pe = prompt_embeds.to(device=dev, dtype=bf16)
        if do_cfg:
            pe = torch.cat([negative_prompt_embeds.to(device=dev, dtype=bf16), pe], dim=0)
        cond = self.unet.precompute_conditioning(pe.contiguous(), None)
        self._eta, self._noise_table = float(eta), None
        if eta > 0:
            if not hasattr(self.scheduler, "_get_variance"):
                raise ValueError("eta > 0 is a DDIMScheduler option (pipeline_stable_diffusion.py:608-625 passes it only "
                                 "to schedulers whose step() accepts it)")
            # the reference draws one randn per step inside scheduler.step (scheduling_ddim.py:500-507), from `generator`
            # in the latents dtype: same draws, in the same order, made up front so the step stays graph-replayable
            gdev = generator.device if generator is not None else dev
            draws = [torch.randn(latents.shape, generator=generator, device=gdev, dtype=bf16)
                     for _ in range(num_inference_steps)]
            table = torch.stack(draws).to(dev).contiguous()
            if self._static.get("noise_table") is not None and self._static["noise_table"].shape == table.shape:
                self._static["noise_table"].copy_(table)       # keep the address the captured graph reads
                table = self._static["noise_table"]
            self._static["noise_table"] = table
            self._noise_table = table
        latents = self._denoise(latents, cond, num_inference_steps, guidance_scale, do_cfg, use_graph)
        images = self._decode(latents, output_type)
        if not return_dict:
            return (images,)
        return PipelineOutput(images=images)


def calculate_shift(image_seq_len, base_seq_len: int = 256, max_seq_len: int = 4096, base_shift: float = 0.5,
                    max_shift: float = 1.15):
    """pipelines/flux/pipeline_flux.py:73-84."""
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


class FluxPipeline(_StepCallbacks, PipelineLoadingMixin):
    """pipelines/flux/pipeline_flux.py:654-980 for pre-computed prompt embeddings (FLUX.1-schnell protocol: no true-CFG,
    no guidance embedding).  The step body -- transformer forward + FlowMatch-Euler update -- is captured once in a HIP
    graph and replayed; latents stay in the packed (B, (h/2)(w/2), 64) layout of the reference throughout the loop."""

    def __init__(self, scheduler: FlowMatchEulerDiscreteScheduler, vae: AutoencoderKL, text_encoder=None, tokenizer=None,
                 text_encoder_2=None, tokenizer_2=None, transformer: FluxTransformer2DModel = None, image_encoder=None,
                 feature_extractor=None):
        self.scheduler, self.vae, self.transformer = scheduler, vae, transformer
        self.text_encoder, self.tokenizer = text_encoder, tokenizer          # optional caller-side transformers modules
        self.text_encoder_2, self.tokenizer_2 = text_encoder_2, tokenizer_2
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1) if vae is not None else 8
        self.default_sample_size = 128
        self._graph = None
        self._graph_key = None
        self._static = {}

    @property
    def device(self):
        return self.transformer.device

    def set_progress_bar_config(self, **kw):
        pass

    @staticmethod
    def _prepare_latent_image_ids(height, width):
        ids = torch.zeros(height, width, 3)
        ids[..., 1] = ids[..., 1] + torch.arange(height)[:, None]
        ids[..., 2] = ids[..., 2] + torch.arange(width)[None, :]
        return ids.reshape(height * width, 3)

    @staticmethod
    def _pack_latents(latents, batch_size, num_channels_latents, height, width):
        latents = latents.view(batch_size, num_channels_latents, height // 2, 2, width // 2, 2)
        latents = latents.permute(0, 2, 4, 1, 3, 5)
        return latents.reshape(batch_size, (height // 2) * (width // 2), num_channels_latents * 4)

    @staticmethod
    def _unpack_latents(latents, height, width, vae_scale_factor):
        batch_size, num_patches, channels = latents.shape
        height = 2 * (int(height) // (vae_scale_factor * 2))
        width = 2 * (int(width) // (vae_scale_factor * 2))
        latents = latents.view(batch_size, height // 2, width // 2, channels // 4, 2, 2)
        latents = latents.permute(0, 3, 1, 4, 2, 5)
        return latents.reshape(batch_size, channels // (2 * 2), height, width)

    def _step(self, latents, pe, cond):
        sch = self.scheduler
        v = self.transformer(latents, encoder_hidden_states=pe, conditioning=cond, sampler_table=sch.device_table,
                             step_idx=sch.device_step, return_dict=False)[0]
        sch.step_inplace(v, latents)
        return latents

    def _denoise(self, latents, pe, cond, num_steps, use_graph):
        sch = self.scheduler
        sch.reset(0)
        if not use_graph:
            for i in range(num_steps):
                with ops.weight_prefetch(_pf(self), "apply" if i else "record"):
                    self._step(latents, pe, cond)
                if not self._after_step(i, latents):
                    break
            return latents
        key = (tuple(latents.shape), tuple(pe.shape), sch.device_table.data_ptr(), sch.device_step.data_ptr(), use_graph == "plan",
               id(self.transformer))                        # (a captured step points into THIS model's packed weights)
        if self._graph is None or self._graph_key != key:
            saved = latents.clone()
            s = _side_stream("warm")
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                with ops.weight_prefetch(_pf(self), "record"):
                    self._step(latents, pe, cond)      # warm-up: variant tuning + lazy one-time driver calls
            torch.cuda.current_stream().wait_stream(s)
            latents.copy_(saved)
            sch.reset(0)
            g = _capture_step(self, lambda: self._step(latents, pe, cond), use_graph)
            self._graph, self._graph_key = g, key
            self._static = {"latents": latents, "pe": pe, "cond": cond}
            latents.copy_(saved)
            sch.reset(0)
        else:
            st = self._static
            st["latents"].copy_(latents)
            st["pe"].copy_(pe)
            st["cond"]["pooled_emb"].copy_(cond["pooled_emb"])
            if st["cond"]["cos"] is not cond["cos"]:     # a different id grid of the same size (e.g. HxW after WxH)
                st["cond"]["cos"].copy_(cond["cos"])
                st["cond"]["sin"].copy_(cond["sin"])
            latents = st["latents"]
        done = 0
        for i in range(num_steps):
            self._graph.replay()
            done = i + 1
            if not self._after_step(i, latents):
                break
        sch._step_index = done
        return latents

    @torch.no_grad()
    def __call__(self, prompt=None, prompt_2=None, negative_prompt=None, negative_prompt_2=None, true_cfg_scale: float = 1.0,
                 height: Optional[int] = None, width: Optional[int] = None, num_inference_steps: int = 28, sigmas=None,
                 guidance_scale: float = 3.5, num_images_per_prompt: int = 1, generator=None,
                 latents: Optional[torch.Tensor] = None, prompt_embeds=None, pooled_prompt_embeds=None,
                 negative_prompt_embeds=None, negative_pooled_prompt_embeds=None, output_type: str = "pt",
                 return_dict: bool = True, max_sequence_length: int = 512, use_graph: bool = True,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=None):
        self._arm_callback(callback_on_step_end, callback_on_step_end_tensor_inputs)      # pipeline_flux.py:626-627, :938-945
        if prompt is not None:
            if prompt_embeds is not None:
                raise ValueError("Cannot forward both `prompt` and `prompt_embeds`. Please make sure to only forward one "
                                 "of the two.")
            if None in (self.tokenizer, self.text_encoder, self.tokenizer_2, self.text_encoder_2):
                raise ValueError("`prompt=` needs the pipeline's CLIP and T5 tokenizers / encoders; without them pass "
                                 "`prompt_embeds` and `pooled_prompt_embeds`")
            from .text_encoding import encode_prompt_flux
            prompt_embeds, pooled_prompt_embeds, _ = encode_prompt_flux(
                self.tokenizer, self.text_encoder, self.tokenizer_2, self.text_encoder_2, prompt, prompt_2, self.device,
                num_images_per_prompt, max_sequence_length)
        else:
            prompt_embeds, pooled_prompt_embeds = (_per_prompt(t, num_images_per_prompt) for t in (prompt_embeds, pooled_prompt_embeds))
        if prompt_embeds is None or pooled_prompt_embeds is None:
            raise ValueError("Provide either `prompt` (with the text encoders given to the pipeline) or `prompt_embeds` "
                             "and `pooled_prompt_embeds`.")
        if negative_prompt_embeds is not None or negative_pooled_prompt_embeds is not None or negative_prompt is not None:
            raise NotImplementedError("true-CFG (negative prompts) is not on the FLUX.1-schnell hot path")
        dev = self.device
        height = height or self.default_sample_size * self.vae_scale_factor
        width = width or self.default_sample_size * self.vae_scale_factor
        if height % (self.vae_scale_factor * 2) or width % (self.vae_scale_factor * 2):
            raise ValueError(f"`height` and `width` have to be divisible by {self.vae_scale_factor * 2}")
        B = prompt_embeds.shape[0]
        lh = 2 * (int(height) // (self.vae_scale_factor * 2))
        lw = 2 * (int(width) // (self.vae_scale_factor * 2))
        nch = self.transformer.config.in_channels // 4
        if latents is None:
            gdev = generator.device if generator is not None else torch.device("cpu")
            raw = torch.randn((B, nch, lh, lw), generator=generator, device=gdev, dtype=bf16)
            latents = self._pack_latents(raw, B, nch, lh, lw)
        if latents.shape[0] != B:
            # (as in the SD / SDXL pipelines: a stale batch would meet a captured step of another size)
            raise ValueError(f"`latents` holds {latents.shape[0]} samples, the prompt embeddings (x num_images_per_prompt) {B}")
        latents = latents.to(device=dev, dtype=bf16).contiguous().clone()
        img_ids = self._prepare_latent_image_ids(lh // 2, lw // 2)
        txt_ids = torch.zeros(prompt_embeds.shape[1], 3)

        n = num_inference_steps
        sig = np.linspace(1.0, 1 / n, n) if sigmas is None else sigmas
        sc = self.scheduler.config
        mu = calculate_shift(latents.shape[1], sc.get("base_image_seq_len", 256), sc.get("max_image_seq_len", 4096),
                             sc.get("base_shift", 0.5), sc.get("max_shift", 1.15))
        self.scheduler.set_timesteps(sigmas=sig, device=dev, mu=mu)
        # what the transformer's sinusoid sees: timestep -> latents dtype, / 1000 (pipeline), * 1000 (model), all in bf16
        t_model = ((self.scheduler.timesteps.to("cpu", bf16) / 1000).to(bf16) * 1000).float()
        self.scheduler.set_model_timesteps(t_model)
        self.scheduler.set_begin_index(0)

        pe = prompt_embeds.to(device=dev, dtype=bf16).contiguous()
        cond = self.transformer.precompute_conditioning(pooled_prompt_embeds.to(device=dev, dtype=bf16), img_ids, txt_ids)
        latents = self._denoise(latents, pe, cond, len(self.scheduler.timesteps), use_graph)
        if output_type == "latent":
            images = latents
        else:
            unp = self._unpack_latents(latents, height, width, self.vae_scale_factor).contiguous()
            vc = self.vae.config
            images = decode_postprocessed(self.vae, unp, output_type, latents_div=float(vc.scaling_factor),
                                          latents_add=float(vc.shift_factor or 0.0))
        if not return_dict:
            return (images,)
        return PipelineOutput(images=images)


class WanPipeline(_StepCallbacks, PipelineLoadingMixin):
    """pipelines/wan/pipeline_wan.py:380-700 (Wan 2.1 T2V) for pre-computed prompt embeddings: the denoising loop.
    The reference runs the transformer twice per step (cond / uncond, :613-632); here the two are one batch-2 call
    (identical arithmetic per sample, twice the GEMM M) and ``uncond + g (cond - uncond)`` is fused into the FlowMatch
    update.  ``output_type="latent"`` returns the latents; "raw" / "pt" / "np" decode them with AutoencoderKLWan (the
    latent de-normalisation of :653-661 is folded into its first conv): "raw" = the clamped decoder output
    [B][3][F][H][W] in [-1, 1], "pt" / "np" = VideoProcessor.postprocess_video of it."""

    def __init__(self, tokenizer=None, text_encoder=None, vae=None, scheduler: FlowMatchEulerDiscreteScheduler = None,
                 transformer: WanTransformer3DModel = None, transformer_2=None, boundary_ratio=None,
                 expand_timesteps: bool = False):
        if transformer_2 is not None or boundary_ratio is not None or expand_timesteps:
            raise NotImplementedError("Wan 2.2 two-stage / TI2V options are not on the BASELINE hot path")
        self.scheduler, self.transformer, self.vae = scheduler, transformer, vae
        self.text_encoder, self.tokenizer = text_encoder, tokenizer          # optional caller-side transformers modules
        self.vae_scale_factor_temporal, self.vae_scale_factor_spatial = 4, 8
        self._graph = None
        self._graph_key = None
        self._static = {}

    @property
    def device(self):
        return self.transformer.device

    def set_progress_bar_config(self, **kw):
        pass

    def _step(self, latents, cond, guidance_scale, do_cfg):
        sch = self.scheduler
        rep = 2 if do_cfg else 1                                              # cond / uncond share the latents
        if latents.dtype == torch.float32:                                    # UniPC keeps fp32 latents (pipeline_wan.py:568)
            x_in = ops.cast_f32_bf16(latents, rep=rep)
        else:
            x_in = ops.mul_scalar(latents, 1.0, rep=rep) if do_cfg else latents
        v = self.transformer(x_in, conditioning=cond, sampler_table=sch.device_table, step_idx=sch.device_step,
                             return_dict=False)[0]
        if do_cfg:
            sch.step_cfg(v, latents, guidance_scale, out=latents)
        else:
            sch.step_inplace(v, latents)
        return latents

    def _denoise(self, latents, cond, num_steps, guidance_scale, do_cfg, use_graph):
        sch = self.scheduler
        sch.reset(0)
        if not use_graph:
            for i in range(num_steps):
                with ops.weight_prefetch(_pf(self), "apply" if i else "record"):
                    self._step(latents, cond, guidance_scale, do_cfg)
                if not self._after_step(i, latents):
                    break
            return latents
        key = (tuple(latents.shape), float(guidance_scale), do_cfg, cond["St"], sch.device_table.data_ptr(), use_graph == "plan",
               id(self.transformer))
        if self._graph is None or self._graph_key != key:
            saved = latents.clone()
            s = _side_stream("warm")
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                with ops.weight_prefetch(_pf(self), "record"):
                    self._step(latents, cond, guidance_scale, do_cfg)
            torch.cuda.current_stream().wait_stream(s)
            latents.copy_(saved)
            sch.reset(0)
            g = _capture_step(self, lambda: self._step(latents, cond, guidance_scale, do_cfg), use_graph)
            self._graph, self._graph_key = g, key
            self._static = {"latents": latents, "cond": cond}
            latents.copy_(saved)
            sch.reset(0)
        else:
            st = self._static
            st["latents"].copy_(latents)
            for (k0, v0), (k1, v1) in zip(st["cond"]["kvs"], cond["kvs"]):
                k0.copy_(k1)
                v0.copy_(v1)
            latents = st["latents"]
        done = 0
        for i in range(num_steps):
            self._graph.replay()
            done = i + 1
            if not self._after_step(i, latents):
                break
        sch._step_index = done
        return latents

    @torch.no_grad()
    def __call__(self, prompt=None, negative_prompt=None, height: int = 480, width: int = 832, num_frames: int = 81,
                 num_inference_steps: int = 50, guidance_scale: float = 5.0, num_videos_per_prompt: int = 1,
                 generator=None, latents: Optional[torch.Tensor] = None, prompt_embeds=None,
                 negative_prompt_embeds=None, output_type: str = "latent", return_dict: bool = True,
                 use_graph: bool = True, max_sequence_length: int = 512, callback_on_step_end=None,
                 callback_on_step_end_tensor_inputs=None):
        self._arm_callback(callback_on_step_end, callback_on_step_end_tensor_inputs)      # pipeline_wan.py:401-402, :637-644
        if prompt is not None:
            if prompt_embeds is not None:
                raise ValueError("Cannot forward both `prompt` and `prompt_embeds`. Please make sure to only forward one "
                                 "of the two.")
            if self.tokenizer is None or self.text_encoder is None:
                raise ValueError("`prompt=` needs the pipeline's tokenizer / text_encoder (UMT5); without them pass "
                                 "`prompt_embeds`")
            from .text_encoding import encode_prompt_wan
            prompt_embeds, negative_prompt_embeds = encode_prompt_wan(
                self.tokenizer, self.text_encoder, prompt, negative_prompt, guidance_scale > 1.0, num_videos_per_prompt,
                max_sequence_length, self.device)
        else:
            prompt_embeds, negative_prompt_embeds = (_per_prompt(t, num_videos_per_prompt) for t in (prompt_embeds, negative_prompt_embeds))
        if prompt_embeds is None:
            raise ValueError("Provide either `prompt` (with the text encoder given to the pipeline) or `prompt_embeds`.")
        if output_type not in ("latent", "pt", "raw", "np"):
            raise ValueError("output_type must be 'latent', 'raw', 'pt' or 'np'")
        if output_type != "latent" and self.vae is None:
            raise ValueError("decoding needs `vae` (diffusers_amd.AutoencoderKLWan)")
        do_cfg = guidance_scale > 1.0
        if do_cfg and negative_prompt_embeds is None:
            raise ValueError("classifier-free guidance needs `negative_prompt_embeds`")
        if num_frames % self.vae_scale_factor_temporal != 1:
            raise ValueError("`num_frames - 1` has to be divisible by 4")
        dev = self.device
        B = prompt_embeds.shape[0]
        if B != 1:
            raise NotImplementedError("one prompt per call (its cond / uncond pair forms the batch)")
        c = self.transformer.config
        shape = (B, c.in_channels, (num_frames - 1) // self.vae_scale_factor_temporal + 1,
                 height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial)
        if latents is None:
            gdev = generator.device if generator is not None else torch.device("cpu")
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32)
        # FlowMatchEuler returns the model dtype after its first step, so the loop runs on bf16 latents; UniPC (the
        # scheduler Wan 2.1 ships) keeps the pipeline's fp32 latents and history (pipeline_wan.py:562-571)
        lat_dtype = torch.float32 if isinstance(self.scheduler, UniPCMultistepScheduler) else bf16
        latents = latents.to(device=dev, dtype=lat_dtype).contiguous().clone()
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        self.scheduler.set_begin_index(0)
        pe = prompt_embeds.to(device=dev, dtype=bf16)
        if do_cfg:
            pe = torch.cat([negative_prompt_embeds.to(device=dev, dtype=bf16), pe], dim=0)   # [uncond ; cond]
        cond = self.transformer.precompute_conditioning(pe.contiguous())
        latents = self._denoise(latents, cond, len(self.scheduler.timesteps), guidance_scale, do_cfg, use_graph)
        if output_type != "latent":
            with _exclusive_decode():
                video = self.vae.decode(latents, return_dict=False, denormalize=True)[0]        # [B][3][F][H][W]
            # VideoProcessor.postprocess_video (video_processor.py): "np" [B][F][H][W][C], "pt" [B][F][C][H][W], in [0, 1]
            latents = postprocess_images(video, output_type)
            if output_type == "pt":
                latents = latents.permute(0, 2, 1, 3, 4)
        if not return_dict:
            return (latents,)
        return PipelineOutput(images=latents)


class DDPMPipeline(PipelineLoadingMixin):
    """pipelines/ddpm/pipeline_ddpm.py:40-130: unconditional ancestral sampling.  The initial image and the per-step
    variance noise are drawn on the host from ``generator`` in fp32 in the reference's order (initial image first, then
    one draw per step with t > 0) and rounded to bf16, so a seeded run consumes the same random stream as the
    reference; each step is the U-Net forward plus ONE fused update kernel (da_x0_linear_step)."""

    def __init__(self, unet: UNet2DModel, scheduler: DDPMScheduler):
        self.unet, self.scheduler = unet, scheduler

    @property
    def device(self):
        return self.unet.device

    def set_progress_bar_config(self, **kw):
        pass

    def _step(self, image, noise_table):
        sch = self.scheduler
        eps = self.unet(image, None, sampler_table=sch.device_table, step_idx=sch.device_step, return_dict=False)[0]
        sch.step_inplace(eps, image, noise_table)

    @torch.no_grad()
    def __call__(self, batch_size: int = 1, generator=None, num_inference_steps: int = 1000, output_type: str = "np",
                 return_dict: bool = True, use_graph: bool = True):
        c = self.unet.config
        ss = c.sample_size
        shape = (batch_size, c.in_channels, ss, ss) if isinstance(ss, int) else (batch_size, c.in_channels, *ss)
        dev = self.device
        sch = self.scheduler
        gdev = generator.device if generator is not None else torch.device("cpu")
        sch.set_timesteps(num_inference_steps, device=dev)
        ts = sch.timesteps.tolist()
        # the reference's random stream (pipeline_ddpm.py:104-121): the initial image, then one draw per step with t > 0,
        # all fp32 (its pipeline is fp32); drawn up front so the loop has no host work, then rounded to bf16
        image = torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32).to(device=dev, dtype=bf16)
        draws = [torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32) if t > 0 else
                 torch.zeros(shape, dtype=torch.float32, device=gdev) for t in ts]
        noise_table = torch.stack(draws).to(device=dev, dtype=bf16).contiguous()
        sch.reset(0)
        if not use_graph:
            for i, _ in enumerate(ts):
                with ops.weight_prefetch(_pf(self), "apply" if i else "record"):
                    self._step(image, noise_table)
        else:
            key = (tuple(shape), len(ts), sch.device_table.data_ptr(), use_graph == "plan", id(self.unet))
            if getattr(self, "_graph_key", None) != key:
                saved = image.clone()
                s = _side_stream("warm")
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    with ops.weight_prefetch(_pf(self), "record"):
                        self._step(image, noise_table)          # warm-up: variant tuning, lazy driver calls
                torch.cuda.current_stream().wait_stream(s)
                image.copy_(saved)
                sch.reset(0)
                g = _capture_step(self, lambda: self._step(image, noise_table), use_graph)
                self._graph, self._graph_key = g, key
                self._static = {"image": image, "noise": noise_table}
                image.copy_(saved)
                sch.reset(0)
            else:
                self._static["image"].copy_(image)
                self._static["noise"].copy_(noise_table)
                image = self._static["image"]
            for _ in ts:
                self._graph.replay()
            sch._step_index = len(ts)
        out = postprocess_images(image.contiguous(), output_type)    # pipeline_ddpm.py:118-121
        if not return_dict:
            return (out,)
        return PipelineOutput(images=out)
