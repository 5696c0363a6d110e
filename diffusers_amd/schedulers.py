"""Schedulers with the reference's SchedulerMixin surface (set_timesteps / scale_model_input / step / timesteps / sigmas /
init_noise_sigma / order / config) whose per-step tensor update is ONE fused HIP kernel (csrc/sampler.hip).

Host side (schedule construction) is numpy / torch-CPU scalar math written to follow the reference formulas exactly:
  EulerDiscreteScheduler          schedulers/scheduling_euler_discrete.py:203-276, :350-481, :326-348, :685-800
  DDIMScheduler                   schedulers/scheduling_ddim.py:212-236, :334-382, :384-514
  DDPMScheduler                   schedulers/scheduling_ddpm.py:348-416, :461-567
  FlowMatchEulerDiscreteScheduler schedulers/scheduling_flow_match_euler_discrete.py:283-382, :423-523
Per-step scalars are evaluated once with fp32 torch scalar ops (same ops, same order as the reference evaluates them
every step) and stored in a device table of 8 floats per step; a device-resident step counter selects the row, so a
denoising step is HIP-graph replayable.  ``step_cfg`` additionally fuses the classifier-free-guidance combine.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from . import _lib as L
from . import ops
from .unet_2d_condition import FrozenConfig


@dataclass
class SchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


def _betas(beta_schedule, beta_start, beta_end, n, trained_betas=None):
    if trained_betas is not None:
        return torch.tensor(trained_betas, dtype=torch.float32)
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    raise NotImplementedError(f"{beta_schedule} is not implemented")


class _SchedulerBase:
    order = 1
    _defaults: dict = {}

    def __init__(self, **kwargs):
        unknown = set(kwargs) - set(self._defaults)
        if unknown:
            raise TypeError(f"{type(self).__name__}: unexpected config keys {sorted(unknown)}")
        cfg = dict(self._defaults)
        cfg.update(kwargs)
        self.config = FrozenConfig(cfg)
        self._step_index = None
        self._begin_index = None
        self._table = None          # device [n_steps][8] fp32
        self._step_dev = None       # device int32 scalar
        self._device = None
        self.num_inference_steps = None

    @classmethod
    def from_config(cls, config, **kwargs):
        """``SchedulerMixin.from_config`` (configuration_utils.py:179-260): build from another scheduler's ``config``
        (a mapping; the reference's FrozenDict or this package's FrozenConfig).  Private entries (``_class_name``,
        ``_diffusers_version`` ...) and options this class does not have are dropped, as the reference does when a config
        is handed to a compatible scheduler class; ``kwargs`` override."""
        src = dict(config)
        src.update(kwargs)
        known = {k: (tuple(v) if isinstance(v, list) else v) for k, v in src.items() if k in cls._defaults}
        return cls(**known)

    # -- reference-compatible index bookkeeping (scheduling_euler_discrete.py:293-324, :640-683) --
    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        ts = self._timesteps_host if schedule_timesteps is None else np.asarray(schedule_timesteps.cpu())
        tv = float(timestep) if not torch.is_tensor(timestep) else float(timestep.item())
        idx = np.nonzero(ts == np.float32(tv))[0] if ts.dtype == np.float32 else np.nonzero(ts == tv)[0]
        if len(idx) == 0:
            raise ValueError(f"timestep {tv} is not in the schedule")
        return int(idx[1] if len(idx) > 1 else idx[0])

    def _init_step_index(self, timestep):
        self._step_index = self.index_for_timestep(timestep) if self._begin_index is None else self._begin_index
        self._sync_device_step()

    def _sync_device_step(self):
        if self._step_dev is not None:
            self._step_dev.fill_(int(self._step_index))

    def _upload(self, rows: np.ndarray, device):
        dev = torch.device(device) if device is not None else torch.device("cuda")
        host = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32))
        if (self._table is not None and self._device == dev and tuple(self._table.shape) == tuple(host.shape)):
            # same schedule length as before: refresh IN PLACE so that HIP graphs that captured these two device
            # addresses (the pipelines' per-step graph) stay valid across set_timesteps() calls
            self._table.copy_(host)
            self._step_dev.zero_()
            return
        self._device = dev
        self._table = host.to(dev)
        self._step_dev = torch.zeros((), dtype=torch.int32, device=dev)

    def _advance(self):
        self._step_index += 1
        ops.advance_step(self._step_dev)

    # engine extension: handles for graph capture
    @property
    def device_table(self):
        return self._table

    @property
    def device_step(self):
        return self._step_dev

    def set_model_timesteps(self, values) -> None:
        """Engine extension: overwrite column 7 of the device table (what the denoiser's sinusoidal embedding reads in
        the graph-replayed step) with the pipeline's own transform of ``timesteps`` -- e.g. Flux feeds
        ``bf16(bf16(t) / 1000) * 1000`` (pipeline_flux.py:902-907, transformer_flux.py:725)."""
        v = torch.as_tensor(values, dtype=torch.float32).reshape(-1)
        if self._table is None or v.numel() != self._table.shape[0]:
            raise ValueError("set_model_timesteps: call set_timesteps() first; one value per step")
        self._table[:, 7].copy_(v.to(self._table.device))

    def reset(self, index: int = 0):
        """Rewind to step ``index`` (host mirror + device counter)."""
        self._step_index = index
        self._sync_device_step()

    def scale_model_input(self, sample, timestep=None):
        return sample


# ----------------------------------------------------------------------------------------------------------------------
def _sigma_to_t(sigma, log_sigmas):
    """Fractional training timestep whose log sigma equals ``log(sigma)`` (piecewise-linear inverse,
    scheduling_euler_discrete.py:483-517)."""
    log_sigma = np.log(np.maximum(sigma, 1e-10))
    dists = log_sigma - log_sigmas[:, np.newaxis]
    low_idx = np.cumsum((dists >= 0), axis=0).argmax(axis=0).clip(max=log_sigmas.shape[0] - 2)
    high_idx = low_idx + 1
    low, high = log_sigmas[low_idx], log_sigmas[high_idx]
    w = np.clip((low - log_sigma) / (low - high), 0, 1)
    return ((1 - w) * low_idx + w * high_idx).reshape(np.shape(sigma))


class EulerDiscreteScheduler(_SchedulerBase):
    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, prediction_type="epsilon", interpolation_type="linear",
                     use_karras_sigmas=False, use_exponential_sigmas=False, use_beta_sigmas=False, sigma_min=None,
                     sigma_max=None, timestep_spacing="linspace", timestep_type="discrete", steps_offset=0,
                     rescale_betas_zero_snr=False, final_sigmas_type="zero")

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        c = self.config
        if c.use_beta_sigmas or c.rescale_betas_zero_snr:
            raise NotImplementedError("EulerDiscreteScheduler: beta sigmas / zero-SNR rescaling are not on the hot path")
        if sum([bool(c.use_karras_sigmas), bool(c.use_exponential_sigmas)]) > 1:
            raise ValueError("Only one of `config.use_beta_sigmas`, `config.use_exponential_sigmas`, "
                             "`config.use_karras_sigmas` can be used.")
        if c.timestep_type != "discrete" or c.interpolation_type != "linear":
            raise NotImplementedError("EulerDiscreteScheduler: only discrete timesteps / linear interpolation")
        if c.prediction_type == "original_sample":       # backwards-compatible alias (scheduling_euler_discrete.py:762)
            self._pred = L.PRED_SAMPLE
        elif c.prediction_type in L.PRED_TYPES:
            self._pred = L.PRED_TYPES[c.prediction_type]
        else:
            raise ValueError(f"prediction_type given as {c.prediction_type} must be one of `epsilon`, or `v_prediction`")
        self.betas = _betas(c.beta_schedule, c.beta_start, c.beta_end, c.num_train_timesteps, c.trained_betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).flip(0)
        ts = np.linspace(0, c.num_train_timesteps - 1, c.num_train_timesteps, dtype=float)[::-1].copy()
        self.timesteps = torch.from_numpy(ts).to(dtype=torch.float32)
        self._timesteps_host = self.timesteps.numpy()
        self.sigmas = torch.cat([sig, torch.zeros(1)])
        self.is_scale_input_called = False

    @property
    def init_noise_sigma(self):
        max_sigma = self.sigmas.max()
        if self.config.timestep_spacing in ("linspace", "trailing"):
            return max_sigma
        return (max_sigma ** 2 + 1) ** 0.5

    def set_timesteps(self, num_inference_steps: int = None, device=None, timesteps=None, sigmas=None):
        c = self.config
        # custom schedules (scheduling_euler_discrete.py:378-407): `timesteps` = the model timesteps, sigmas interpolated from the
        # training ladder; `sigmas` = the whole ladder INCLUDING its terminal value, timesteps found by inverting log sigma(t)
        if num_inference_steps is None and timesteps is None and sigmas is None:
            raise ValueError("Must pass exactly one of `num_inference_steps` or `timesteps` or `sigmas.")
        if num_inference_steps is not None and (timesteps is not None or sigmas is not None):
            raise ValueError("Can only pass one of `num_inference_steps` or `timesteps` or `sigmas`.")
        if timesteps is not None and sigmas is not None:          # scheduling_euler_discrete.py:381-382
            raise ValueError("Only one of `timesteps` or `sigmas` should be set.")
        if timesteps is not None and c.use_karras_sigmas:
            raise ValueError("Cannot set `timesteps` with `config.use_karras_sigmas = True`.")
        if timesteps is not None and c.use_exponential_sigmas:
            raise ValueError("Cannot set `timesteps` with `config.use_exponential_sigmas = True`.")
        if num_inference_steps is None:
            num_inference_steps = len(timesteps) if timesteps is not None else len(sigmas) - 1
        self.num_inference_steps = num_inference_steps
        n_train = c.num_train_timesteps
        base = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        if sigmas is not None:
            sig_full = np.array(sigmas).astype(np.float32)
            ts = np.array([_sigma_to_t(s_, np.log(base)) for s_ in sig_full[:-1]])
            self._finish_timesteps(sig_full, ts, num_inference_steps, device)
            return
        if timesteps is not None:
            ts = np.array(timesteps).astype(np.float32)
        elif c.timestep_spacing == "linspace":
            ts = np.linspace(0, n_train - 1, num_inference_steps, dtype=np.float32)[::-1].copy()
        elif c.timestep_spacing == "leading":
            ratio = n_train // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.float32)
            ts += c.steps_offset
        elif c.timestep_spacing == "trailing":
            ratio = n_train / num_inference_steps
            ts = (np.arange(n_train, 0, -ratio)).round().copy().astype(np.float32)
            ts -= 1
        else:
            raise ValueError(f"{c.timestep_spacing} is not supported. Please make sure to choose one of 'linspace', "
                             "'leading' or 'trailing'.")
        sig = np.interp(ts, np.arange(0, len(base)), base)
        if c.use_karras_sigmas or c.use_exponential_sigmas:
            # a re-spaced sigma ladder between the interpolated extremes (scheduling_euler_discrete.py:446-452, :520-585);
            # the model then sees fractional timesteps found by inverting log sigma(t) (:483-517).  Host tables only:
            # the step kernel reads sigma / dt / timestep from the same table as before.
            smin = c.sigma_min if c.sigma_min is not None else sig[-1].item()
            smax = c.sigma_max if c.sigma_max is not None else sig[0].item()
            if c.use_karras_sigmas:
                rho = 7.0
                ramp = np.linspace(0, 1, num_inference_steps)
                sig = (smax ** (1 / rho) + ramp * (smin ** (1 / rho) - smax ** (1 / rho))) ** rho
            else:
                sig = np.exp(np.linspace(math.log(smax), math.log(smin), num_inference_steps))
            ts = np.array([_sigma_to_t(s_, np.log(base)) for s_ in sig])
        if c.final_sigmas_type == "sigma_min":
            last = float(((1 - self.alphas_cumprod[0]) / self.alphas_cumprod[0]) ** 0.5)
        elif c.final_sigmas_type == "zero":
            last = 0
        else:
            raise ValueError(f"`final_sigmas_type` must be one of 'zero', or 'sigma_min', but got {c.final_sigmas_type}")
        self._finish_timesteps(np.concatenate([sig, [last]]).astype(np.float32), ts, num_inference_steps, device)

    def _finish_timesteps(self, sig, ts, num_inference_steps, device):
        """Sigma ladder (terminal value included) + model timesteps -> the scheduler's state and the device coefficient table."""
        self.sigmas = torch.from_numpy(sig).to(dtype=torch.float32)  # kept on CPU like the reference
        self.timesteps = torch.from_numpy(ts.astype(np.float32)).to(device=device)
        self._timesteps_host = ts.astype(np.float32)
        self._step_index = None
        self._begin_index = None
        rows = np.zeros((num_inference_steps, 8), dtype=np.float32)
        for i in range(num_inference_steps):
            s, s_next = self.sigmas[i], self.sigmas[i + 1]
            rows[i, 0] = float(s)
            rows[i, 1] = float(s_next)
            rows[i, 2] = float(s_next - s)                 # dt, fp32 torch scalar arithmetic as the reference
            rows[i, 3] = float((s ** 2 + 1) ** 0.5)        # scale_model_input denominator
            rows[i, 4] = float(-s / (s ** 2 + 1) ** 0.5)   # v_prediction: c_out (scheduling_euler_discrete.py:767)
            rows[i, 5] = float(s ** 2 + 1)                 # v_prediction: sample / (sigma^2 + 1)
            rows[i, 7] = float(ts[i])
        self._upload(rows, device)

    def scale_model_input(self, sample, timestep=None, rep: int = 1):
        if self._step_index is None:
            self._init_step_index(timestep)
        self.is_scale_input_called = True
        return ops.euler_scale_model_input(sample, self._table, self._step_dev, rep=rep)

    def step(self, model_output, timestep, sample, s_churn: float = 0.0, s_tmin: float = 0.0,
             s_tmax: float = float("inf"), s_noise: float = 1.0, generator=None, return_dict: bool = True):
        if isinstance(timestep, int) or (torch.is_tensor(timestep) and timestep.dtype in (torch.int32, torch.int64)):
            raise ValueError("Passing integer indices (e.g. from `enumerate(timesteps)`) as timesteps to "
                             "`EulerDiscreteScheduler.step()` is not supported. Make sure to pass one of the "
                             "`scheduler.timesteps` as a timestep.")
        if s_churn != 0.0:
            raise NotImplementedError("s_churn > 0 (stochastic Euler) is not on the hot path")
        if self._step_index is None:
            self._init_step_index(timestep)
        prev = ops.euler_step(model_output.contiguous(), sample.contiguous(), self._table, self._step_dev, cfg=False,
                              guidance=0.0, pred_type=self._pred)
        self._advance()
        if not return_dict:
            return (prev, None)
        return SchedulerOutput(prev_sample=prev)

    def step_cfg(self, model_output_2b, sample, guidance_scale: float, out=None, cfg: bool = True):
        """Engine extension: CFG combine + Euler step in one kernel; model_output_2b = [uncond ; cond] (``cfg=False``:
        a plain model output -- the ``guidance_scale <= 1`` loop of the pipelines -- with the same in-place ``out``)."""
        if self._step_index is None:
            self._init_step_index(self.timesteps[0])
        prev = ops.euler_step(model_output_2b, sample, self._table, self._step_dev, cfg=cfg,
                              guidance=float(guidance_scale), out=out, pred_type=self._pred)
        self._advance()
        return prev


# ----------------------------------------------------------------------------------------------------------------------
class DDIMScheduler(_SchedulerBase):
    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                     prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                     clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                     rescale_betas_zero_snr=False)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        c = self.config
        if c.thresholding or c.rescale_betas_zero_snr:
            raise NotImplementedError("DDIMScheduler: dynamic thresholding / zero-SNR rescaling are not on the hot path")
        if c.prediction_type not in L.PRED_TYPES:
            raise ValueError(f"prediction_type given as {c.prediction_type} must be one of `epsilon`, `sample`, or "
                             "`v_prediction`")
        self._pred = L.PRED_TYPES[c.prediction_type]
        self.betas = _betas(c.beta_schedule, c.beta_start, c.beta_end, c.num_train_timesteps, c.trained_betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if c.set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.timesteps = torch.from_numpy(np.arange(0, c.num_train_timesteps)[::-1].copy().astype(np.int64))
        self._timesteps_host = self.timesteps.numpy()
        self._eta = None

    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        return (b_prev / b_t) * (1 - a_t / a_prev)

    def set_timesteps(self, num_inference_steps: int, device=None):
        c = self.config
        if num_inference_steps > c.num_train_timesteps:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                             f"`self.config.train_timesteps`: {c.num_train_timesteps} as the unet model trained with "
                             f"this scheduler can only handle maximal {c.num_train_timesteps} timesteps.")
        self.num_inference_steps = num_inference_steps
        n_train = c.num_train_timesteps
        if c.timestep_spacing == "linspace":
            ts = np.linspace(0, n_train - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif c.timestep_spacing == "leading":
            ratio = n_train // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
            ts += c.steps_offset
        elif c.timestep_spacing == "trailing":
            ratio = n_train / num_inference_steps
            ts = np.round(np.arange(n_train, 0, -ratio)).astype(np.int64)
            ts -= 1
        else:
            raise ValueError(f"{c.timestep_spacing} is not supported. Please make sure to choose one of 'leading' or "
                             "'trailing'.")
        self.timesteps = torch.from_numpy(ts).to(device)
        self._timesteps_host = ts
        self._step_index = None
        self._device_req = device
        self._eta = None   # table rows are rebuilt lazily for the eta of the first step(); buffers are reused in place

    def _build(self, eta: float):
        c = self.config
        ts = self._timesteps_host
        n = len(ts)
        rows = np.zeros((n, 8), dtype=np.float32)
        for i, t in enumerate(ts):
            t = int(t)
            prev_t = t - c.num_train_timesteps // self.num_inference_steps
            a_t = self.alphas_cumprod[t]
            a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
            b_t = 1 - a_t
            variance = self._get_variance(t, prev_t)
            std = eta * variance ** 0.5
            rows[i, 0] = float(b_t ** 0.5)
            rows[i, 1] = float(a_t ** 0.5)
            rows[i, 2] = float(a_prev ** 0.5)
            rows[i, 3] = float((1 - a_prev - std ** 2) ** 0.5)
            rows[i, 4] = 0.0
            rows[i, 5] = float(std) if eta > 0 else 0.0
            rows[i, 6] = float(c.clip_sample_range) if c.clip_sample else 0.0
            rows[i, 7] = float(t)
        self._upload(rows, self._device_req)
        self._eta = eta

    def _ensure(self, eta, timestep):
        if self._table is None or self._eta != eta:
            keep = self._step_index
            self._build(eta)
            self._step_index = keep
            if keep is not None:
                self._sync_device_step()
        if self._step_index is None:
            self._init_step_index(timestep)

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
             generator=None, variance_noise=None, return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating "
                             "the scheduler")
        if use_clipped_model_output:
            raise NotImplementedError("use_clipped_model_output is not on the hot path")
        self._ensure(eta, timestep)
        # DDIM is stateless in the reference (indexed by timestep): honour out-of-order calls
        idx = self.index_for_timestep(timestep)
        if idx != self._step_index:
            self._step_index = idx
            self._sync_device_step()
        noise = None
        if eta > 0:
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either "
                                 "`generator` or `variance_noise` stays `None`.")
            if variance_noise is None:
                gdev = generator.device if generator is not None else model_output.device
                variance_noise = torch.randn(model_output.shape, generator=generator, device=gdev,
                                             dtype=model_output.dtype).to(model_output.device)
            noise = variance_noise
        prev = ops.x0_linear_step(model_output.contiguous(), sample.contiguous(), noise, self._table, self._step_dev,
                                  cfg=False, guidance=0.0, pred_type=self._pred)
        self._advance()
        if not return_dict:
            return (prev, None)
        return SchedulerOutput(prev_sample=prev)

    def step_cfg(self, model_output_2b, sample, guidance_scale: float, eta: float = 0.0, out=None, cfg: bool = True,
                 noise_table=None):
        """Engine extension: CFG combine (``cfg=False``: plain model output) + DDIM update in one kernel, in place when
        ``out`` is the sample.  ``eta > 0`` needs ``noise_table`` [steps][numel]: every step's variance noise drawn up
        front (the kernel picks the row of the device step counter, so the step stays HIP-graph replayable)."""
        self._ensure(eta, self.timesteps[0])
        if eta > 0 and noise_table is None:
            raise ValueError("DDIMScheduler.step_cfg: eta > 0 needs the pre-drawn `noise_table` [steps][numel]")
        prev = ops.x0_linear_step(model_output_2b, sample, noise_table if eta > 0 else None, self._table, self._step_dev,
                                  cfg=cfg, guidance=float(guidance_scale), out=out, pred_type=self._pred,
                                  noise_step_stride=sample.numel() if eta > 0 else 0)
        self._advance()
        return prev

    @property
    def device_table(self):
        if self._table is None or self._eta is None:
            self._build(0.0)
        return self._table

    @property
    def device_step(self):
        if self._table is None or self._eta is None:
            self._build(0.0)
        return self._step_dev


# ----------------------------------------------------------------------------------------------------------------------
class DDPMScheduler(_SchedulerBase):
    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, variance_type="fixed_small", clip_sample=True, prediction_type="epsilon",
                     thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0,
                     sample_max_value=1.0, timestep_spacing="leading", steps_offset=0, rescale_betas_zero_snr=False)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        c = self.config
        if c.thresholding or c.rescale_betas_zero_snr:
            raise NotImplementedError("DDPMScheduler: dynamic thresholding / zero-SNR rescaling are not on the hot path")
        if c.prediction_type not in L.PRED_TYPES:
            raise ValueError(f"prediction_type given as {c.prediction_type} must be one of `epsilon`, `sample` or "
                             "`v_prediction`  for the DDPMScheduler.")
        self._pred = L.PRED_TYPES[c.prediction_type]
        if c.variance_type not in ("fixed_small", "fixed_large"):
            raise NotImplementedError("DDPMScheduler: only fixed_small / fixed_large variance")
        self.betas = _betas(c.beta_schedule, c.beta_start, c.beta_end, c.num_train_timesteps, c.trained_betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.custom_timesteps = False
        self.timesteps = torch.from_numpy(np.arange(0, c.num_train_timesteps)[::-1].copy())
        self._timesteps_host = self.timesteps.numpy()
        self.variance_type = c.variance_type

    def set_timesteps(self, num_inference_steps: int = None, device=None, timesteps=None):
        c = self.config
        if timesteps is not None:
            raise NotImplementedError("custom timesteps are not supported by the HIP DDPMScheduler")
        if num_inference_steps > c.num_train_timesteps:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                             f"`self.config.train_timesteps`: {c.num_train_timesteps}")
        self.num_inference_steps = num_inference_steps
        n_train = c.num_train_timesteps
        if c.timestep_spacing == "linspace":
            ts = np.linspace(0, n_train - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif c.timestep_spacing == "leading":
            ratio = n_train // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
            ts += c.steps_offset
        elif c.timestep_spacing == "trailing":
            ratio = n_train / num_inference_steps
            ts = np.round(np.arange(n_train, 0, -ratio)).astype(np.int64)
            ts -= 1
        else:
            raise ValueError(f"{c.timestep_spacing} is not supported.")
        self.timesteps = torch.from_numpy(ts).to(device)
        self._timesteps_host = ts
        self._step_index = None
        rows = np.zeros((len(ts), 8), dtype=np.float32)
        for i, t in enumerate(ts):
            t = int(t)
            # previous_timestep() (scheduling_ddpm.py:648-668): the NEXT entry of the schedule, -1 after the last one --
            # not t - n_train // n_steps, which differs for 'linspace' / 'trailing' spacings
            prev_t = int(ts[i + 1]) if i + 1 < len(ts) else -1
            a_t = self.alphas_cumprod[t]
            a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
            b_t = 1 - a_t
            b_prev = 1 - a_prev
            cur_a = a_t / a_prev
            cur_b = 1 - cur_a
            k0 = (a_prev ** 0.5 * cur_b) / b_t
            kx = cur_a ** 0.5 * b_prev / b_t
            var = (1 - a_prev) / (1 - a_t) * cur_b
            var = torch.clamp(var, min=1e-20)
            if self.variance_type == "fixed_large":
                var = cur_b
            rows[i, 0] = float(b_t ** 0.5)
            rows[i, 1] = float(a_t ** 0.5)
            rows[i, 2] = float(k0)
            rows[i, 3] = 0.0
            rows[i, 4] = float(kx)
            rows[i, 5] = float(var ** 0.5) if t > 0 else 0.0
            rows[i, 6] = float(c.clip_sample_range) if c.clip_sample else 0.0
            rows[i, 7] = float(t)
        self._upload(rows, device)

    def step(self, model_output, timestep, sample, generator=None, return_dict: bool = True, noise=None):
        """scheduling_ddpm.py:461-567.  ``noise`` (engine extension): this step's variance noise instead of a draw from
        ``generator`` (which, like the reference's randn_tensor, draws in the dtype of ``model_output``)."""
        idx = self.index_for_timestep(timestep)
        if idx != self._step_index:
            self._step_index = idx
            self._sync_device_step()
        t = int(self._timesteps_host[idx])
        if t > 0 and noise is None:
            gdev = generator.device if generator is not None else model_output.device
            noise = torch.randn(model_output.shape, generator=generator, device=gdev,
                                dtype=model_output.dtype).to(model_output.device)
        prev = ops.x0_linear_step(model_output.contiguous(), sample.contiguous(), noise if t > 0 else None, self._table,
                                  self._step_dev, cfg=False, guidance=0.0, pred_type=self._pred)
        self._advance()
        if not return_dict:
            return (prev, None)
        return SchedulerOutput(prev_sample=prev)

    def step_inplace(self, model_output, sample, noise_table):
        """Engine extension for HIP-graph replay: the update written over ``sample``; ``noise_table`` [steps][numel]
        holds every step's pre-drawn variance noise, the kernel picks row ``step`` (kn = 0 on the last step)."""
        if self._step_index is None:
            self._step_index = 0
            self._sync_device_step()
        ops.x0_linear_step(model_output, sample, noise_table, self._table, self._step_dev, cfg=False, guidance=0.0,
                           out=sample, noise_step_stride=sample.numel(), pred_type=self._pred)
        self._advance()
        return sample


# ----------------------------------------------------------------------------------------------------------------------
class FlowMatchEulerDiscreteScheduler(_SchedulerBase):
    _defaults = dict(num_train_timesteps=1000, shift=1.0, use_dynamic_shifting=False, base_shift=0.5, max_shift=1.15,
                     base_image_seq_len=256, max_image_seq_len=4096, invert_sigmas=False, shift_terminal=None,
                     use_karras_sigmas=False, use_exponential_sigmas=False, use_beta_sigmas=False,
                     time_shift_type="exponential", stochastic_sampling=False)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        c = self.config
        if c.use_karras_sigmas or c.use_exponential_sigmas or c.use_beta_sigmas or c.invert_sigmas \
                or c.shift_terminal or c.stochastic_sampling:
            raise NotImplementedError("FlowMatchEulerDiscreteScheduler: option not on the hot path")
        if c.time_shift_type not in {"exponential", "linear"}:
            raise ValueError("`time_shift_type` must either be 'exponential' or 'linear'.")
        ts = np.linspace(1, c.num_train_timesteps, c.num_train_timesteps, dtype=np.float32)[::-1].copy()
        ts = torch.from_numpy(ts).to(dtype=torch.float32)
        sig = ts / c.num_train_timesteps
        if not c.use_dynamic_shifting:
            sig = c.shift * sig / (1 + (c.shift - 1) * sig)
        self.timesteps = sig * c.num_train_timesteps
        self._timesteps_host = self.timesteps.numpy()
        self._shift = c.shift
        self.sigmas = sig
        self.sigma_min = self.sigmas[-1].item()
        self.sigma_max = self.sigmas[0].item()
        self.init_noise_sigma = 1.0

    @property
    def shift(self):
        return self._shift

    def set_shift(self, shift: float):
        self._shift = shift

    def time_shift(self, mu: float, sigma: float, t):
        if self.config.time_shift_type == "exponential":
            return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)
        return mu / (mu + (1 / t - 1) ** sigma)

    def set_timesteps(self, num_inference_steps: int = None, device=None, sigmas: Optional[List[float]] = None,
                      mu: Optional[float] = None, timesteps: Optional[List[float]] = None):
        c = self.config
        if c.use_dynamic_shifting and mu is None:
            raise ValueError("`mu` must be passed when `use_dynamic_shifting` is set to be `True`")
        if timesteps is not None:
            raise NotImplementedError("custom timesteps are not supported by the HIP FlowMatch scheduler")
        if sigmas is not None and num_inference_steps is not None and len(sigmas) != num_inference_steps:
            raise ValueError("`sigmas` and `timesteps` should have the same length as num_inference_steps, if "
                             "`num_inference_steps` is provided")
        if num_inference_steps is None:
            num_inference_steps = len(sigmas)
        self.num_inference_steps = num_inference_steps
        if sigmas is None:
            ts = np.linspace(self.sigma_max * c.num_train_timesteps, self.sigma_min * c.num_train_timesteps,
                             num_inference_steps)
            sig = ts / c.num_train_timesteps
        else:
            sig = np.array(sigmas).astype(np.float32)
        if c.use_dynamic_shifting:
            sig = self.time_shift(mu, 1.0, sig)
        else:
            sig = self.shift * sig / (1 + (self.shift - 1) * sig)
        sig_t = torch.from_numpy(np.asarray(sig)).to(dtype=torch.float32)
        ts_t = sig_t * c.num_train_timesteps
        sig_t = torch.cat([sig_t, torch.zeros(1)])
        self.timesteps = ts_t.to(device)
        self._timesteps_host = ts_t.numpy()
        self.sigmas = sig_t.to(device)
        self._step_index = None
        self._begin_index = None
        rows = np.zeros((num_inference_steps, 8), dtype=np.float32)
        for i in range(num_inference_steps):
            s, s_next = sig_t[i], sig_t[i + 1]
            rows[i, 0] = float(s)
            rows[i, 1] = float(s_next)
            rows[i, 2] = float(s_next - s)
            rows[i, 7] = float(ts_t[i])
        self._upload(rows, device)

    def step(self, model_output, timestep, sample, s_churn: float = 0.0, s_tmin: float = 0.0,
             s_tmax: float = float("inf"), s_noise: float = 1.0, generator=None, per_token_timesteps=None,
             return_dict: bool = True):
        if isinstance(timestep, int) or (torch.is_tensor(timestep) and timestep.dtype in (torch.int32, torch.int64)):
            raise ValueError("Passing integer indices (e.g. from `enumerate(timesteps)`) as timesteps to "
                             "`FlowMatchEulerDiscreteScheduler.step()` is not supported. Make sure to pass one of the "
                             "`scheduler.timesteps` as a timestep.")
        if per_token_timesteps is not None:
            raise NotImplementedError("per_token_timesteps is not on the hot path")
        if self._step_index is None:
            self._init_step_index(timestep)
        prev = ops.flowmatch_step(model_output, sample, self._table, self._step_dev)
        self._advance()
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev)

    def step_inplace(self, model_output, sample):
        """Engine extension: the update written over ``sample`` (same buffer every HIP-graph replay)."""
        if self._step_index is None:
            self._init_step_index(self.timesteps[0])
        ops.flowmatch_step(model_output, sample, self._table, self._step_dev, out=sample)
        self._advance()
        return sample

    def step_cfg(self, model_output_2b, sample, guidance_scale: float, out=None):
        if self._step_index is None:
            self._init_step_index(self.timesteps[0])
        prev = ops.flowmatch_step(model_output_2b, sample, self._table, self._step_dev, cfg=True,
                                  guidance=float(guidance_scale), out=out)
        self._advance()
        return prev


# ----------------------------------------------------------------------------------------------------------------------
class UniPCMultistepScheduler(_SchedulerBase):
    """schedulers/scheduling_unipc_multistep.py in the configuration Wan 2.1 ships (pipeline_wan.py:52-59):
    ``prediction_type="flow_prediction"``, ``use_flow_sigmas=True``, ``flow_shift``, ``solver_order`` <= 2, B(h) solver,
    ``predict_x0``, ``lower_order_final``.  The order-2 predictor-corrector keeps two x0 predictions and the pre-predictor
    sample; the host evaluates the per-step coefficients with the reference's fp32 scalar ops (set_timesteps :428-466,
    _coeffs = :868-905 / :1030-1067) and ONE kernel per step (da_unipc_flow_step) applies them in the reference's
    operation order and rolls the history in place, so the step is HIP-graph replayable."""

    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, solver_order=2, prediction_type="epsilon", thresholding=False,
                     dynamic_thresholding_ratio=0.995, sample_max_value=1.0, predict_x0=True, solver_type="bh2",
                     lower_order_final=True, disable_corrector=[], solver_p=None, use_karras_sigmas=False,
                     use_exponential_sigmas=False, use_beta_sigmas=False, use_flow_sigmas=False, flow_shift=1.0,
                     timestep_spacing="linspace", steps_offset=0, final_sigmas_type="zero",
                     rescale_betas_zero_snr=False, use_dynamic_shifting=False, time_shift_type="exponential",
                     sigma_min=None, sigma_max=None, shift_terminal=None)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        c = self.config
        if c.prediction_type != "flow_prediction" or not c.use_flow_sigmas:
            raise NotImplementedError("UniPCMultistepScheduler: only the flow-matching configuration "
                                      "(prediction_type='flow_prediction', use_flow_sigmas=True) is on the hot path")
        if (not c.predict_x0 or c.solver_order not in (1, 2) or c.solver_type not in ("bh1", "bh2") or c.thresholding
                or c.disable_corrector or c.solver_p is not None or c.use_karras_sigmas or c.use_exponential_sigmas
                or c.use_beta_sigmas or c.use_dynamic_shifting or c.shift_terminal or not c.lower_order_final
                or c.final_sigmas_type != "zero"):
            raise NotImplementedError("UniPCMultistepScheduler: unsupported option for the HIP step")
        self.init_noise_sigma = 1.0
        ts = np.linspace(0, c.num_train_timesteps - 1, c.num_train_timesteps, dtype=np.float32)[::-1].copy()
        self.timesteps = torch.from_numpy(ts)
        self._timesteps_host = ts
        self._coef = None
        self._hist = None
        self.sigmas = None

    def _coeffs(self, sigma_t, sigma_s0, hist_sigmas, order):
        """The scalar part of multistep_uni_{p,c}_bh_update, fp32 torch ops in the reference's order."""
        alpha_t, alpha_s0 = 1 - sigma_t, 1 - sigma_s0
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        rks = [((torch.log(1 - s_) - torch.log(s_)) - lambda_s0) / h for s_ in hist_sigmas[: order - 1]]
        rks.append(torch.ones(()))
        rks = torch.stack(rks)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = hh if self.config.solver_type == "bh1" else torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return sigma_t / sigma_s0, alpha_t * h_phi_1, alpha_t * B_h, rks, torch.stack(R), torch.stack(b)

    def set_timesteps(self, num_inference_steps: int = None, device=None, sigmas=None, mu=None):
        c = self.config
        if sigmas is not None:
            raise NotImplementedError("custom sigmas are not supported by the HIP UniPC scheduler")
        n = num_inference_steps
        sig = np.linspace(1, 1 / c.num_train_timesteps, n + 1)[:-1]
        sig = c.flow_shift * sig / (1 + (c.flow_shift - 1) * sig)
        if np.fabs(sig[0] - 1) < 1e-6:
            sig[0] -= 1e-6
        ts = (sig * c.num_train_timesteps).copy()
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts).to(device=device, dtype=torch.int64)
        self._timesteps_host = self.timesteps.cpu().numpy()
        self.num_inference_steps = n
        self._step_index = None
        self._begin_index = None
        S = self.sigmas
        coef = np.zeros((n, 16), dtype=np.float32)
        rows = np.zeros((n, 8), dtype=np.float32)
        lower, prev_order = 0, None
        for i in range(n):
            coef[i, 0] = float(S[i])
            if i > 0:      # corrector of step i runs with the predictor order of step i-1 (step(): :1262-1270)
                oc = prev_order
                c1, c2, c3, rks, R, b = self._coeffs(S[i], S[i - 1], [S[i - (k + 1)] for k in range(1, oc)], oc)
                rhos = torch.ones(1) * 0.5 if oc == 1 else torch.linalg.solve(R, b)
                coef[i, 1], coef[i, 2] = 1.0, float(oc)
                coef[i, 3], coef[i, 4], coef[i, 5] = float(c1), float(c2), float(c3)
                coef[i, 6] = float(rks[0]) if oc == 2 else 1.0
                coef[i, 7] = float(rhos[0]) if oc == 2 else 0.0
                coef[i, 8] = float(rhos[-1])
            op = min(min(c.solver_order, n - i), lower + 1)
            c1, c2, c3, rks, _, _ = self._coeffs(S[i + 1], S[i], [S[i - k] for k in range(1, op)], op)
            coef[i, 9], coef[i, 10], coef[i, 11], coef[i, 12] = float(op), float(c1), float(c2), float(c3)
            coef[i, 13] = float(rks[0]) if op == 2 else 1.0
            rows[i, 0], rows[i, 1], rows[i, 7] = float(S[i]), float(S[i + 1]), float(self._timesteps_host[i])
            prev_order = op
            if lower < c.solver_order:
                lower += 1
        self._upload(rows, device)
        host = torch.from_numpy(coef)
        if self._coef is not None and tuple(self._coef.shape) == tuple(host.shape) and self._coef.device == self._table.device:
            self._coef.copy_(host)          # in place: captured graphs keep their addresses
        else:
            self._coef = host.to(self._table.device)
        # the history tensors are kept (a captured graph holds their addresses); their stale content is never used:
        # step 0 runs without corrector at order 1, and every later read was written by an earlier step

    def _history(self, sample):
        if self._hist is None or self._hist[0].shape != sample.shape or self._hist[0].dtype != sample.dtype:
            self._hist = tuple(torch.zeros_like(sample) for _ in range(3))     # last_sample, m1, m2
        return self._hist

    def step(self, model_output, timestep, sample, return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the "
                             "scheduler")
        if self._step_index is None:
            self._init_step_index(timestep)
        prev = sample.clone()
        last, m1, m2 = self._history(sample)
        ops.unipc_flow_step_(model_output, prev, last, m1, m2, self._coef, self._step_dev)
        self._advance()
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev)

    def step_cfg(self, model_output_2b, sample, guidance_scale: float, out=None):
        """Engine extension: CFG combine + UniPC step, written over ``sample`` (graph replay)."""
        if self._step_index is None:
            self._init_step_index(self.timesteps[0])
        if out is not None and out.data_ptr() != sample.data_ptr():
            raise ValueError("UniPC step_cfg updates `sample` in place")
        last, m1, m2 = self._history(sample)
        ops.unipc_flow_step_(model_output_2b, sample, last, m1, m2, self._coef, self._step_dev, cfg=True,
                             guidance=float(guidance_scale))
        self._advance()
        return sample

    def step_inplace(self, model_output, sample):
        if self._step_index is None:
            self._init_step_index(self.timesteps[0])
        last, m1, m2 = self._history(sample)
        ops.unipc_flow_step_(model_output, sample, last, m1, m2, self._coef, self._step_dev)
        self._advance()
        return sample
