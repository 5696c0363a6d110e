"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference schedulers' per-step math.

Follows (paths under /root/reference/src/diffusers/):
  Euler      schedulers/scheduling_euler_discrete.py:203-276 (tables), :350-481 (set_timesteps), :326-348, :685-800
  DDIM       schedulers/scheduling_ddim.py:212-236, :334-382, :384-514
  DDPM       schedulers/scheduling_ddpm.py:348-416, :461-567
  FlowMatch  schedulers/scheduling_flow_match_euler_discrete.py:283-382, :423-523
  CFG        pipelines/stable_diffusion/pipeline_stable_diffusion.py:1054-1055

Pinned against the reference's own known-answer vectors (tests/schedulers/test_scheduler_euler.py:129-137 -> 10.0807 /
0.0131; test_scheduler_ddim.py:114-150 -> 172.0067, 149.8295, 149.0784; test_scheduler_ddpm.py:75-104 -> 258.9606) in
tests/test_oracle_schedulers.py, and against live reference runs frozen in tests/golden/schedulers.npz.
Tensor ops are torch CPU ops, so dtype promotion / rounding points are the reference's by construction.

One torch quirk matters for bf16 latents: ``scalar_tensor * bf16_tensor`` (0-d fp32 scalar FIRST) rounds the scalar to
bf16 on torch's CPU kernels, while torch's device kernels (``opmath_symmetric_gpu_kernel_with_scalars``) keep it in
fp32 in either operand order.  Every oracle class therefore takes ``device_scalars``: False (default) = the reference
as it runs on CPU (what tests/golden/schedulers.npz froze), True = the reference as it runs on a GPU (fp32 scalars),
which is the behaviour the HIP kernels implement bit for bit.  For fp32 latents both modes are identical.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def make_betas(schedule: str, beta_start: float, beta_end: float, n: int) -> torch.Tensor:
    if schedule == "linear":
        return torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    if schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    raise NotImplementedError(schedule)


def spaced_timesteps(spacing: str, n_train: int, n_inf: int, steps_offset: int, as_float: bool) -> np.ndarray:
    if spacing == "linspace":
        if as_float:
            return np.linspace(0, n_train - 1, n_inf, dtype=np.float32)[::-1].copy()
        return np.linspace(0, n_train - 1, n_inf).round()[::-1].copy().astype(np.int64)
    if spacing == "leading":
        ratio = n_train // n_inf
        ts = (np.arange(0, n_inf) * ratio).round()[::-1].copy()
        ts = ts.astype(np.float32 if as_float else np.int64)
        return ts + steps_offset
    if spacing == "trailing":
        ratio = n_train / n_inf
        ts = np.round(np.arange(n_train, 0, -ratio))
        ts = ts.astype(np.float32 if as_float else np.int64)
        return ts - 1
    raise ValueError(spacing)


def smul(scalar, tensor: torch.Tensor, device_scalars: bool) -> torch.Tensor:
    """``scalar * tensor`` with the scalar first, as the reference writes it (see the module docstring)."""
    if device_scalars and tensor.dtype in (torch.bfloat16, torch.float16):
        return (tensor.float() * scalar).to(tensor.dtype)
    return scalar * tensor


def cfg_combine(uncond: torch.Tensor, cond: torch.Tensor, g: float) -> torch.Tensor:
    return uncond + g * (cond - uncond)


def rescale_noise_cfg(noise_cfg: torch.Tensor, noise_pred_text: torch.Tensor, guidance_rescale: float = 0.0) -> torch.Tensor:
    """pipelines/stable_diffusion/pipeline_stable_diffusion.py:69-92 (pipeline_stable_diffusion_xl.py:84-107): match the
    per-sample std of the guided prediction to that of the text prediction, then mix by ``guidance_rescale``."""
    dims = list(range(1, noise_pred_text.ndim))
    std_text = noise_pred_text.std(dim=dims, keepdim=True)
    std_cfg = noise_cfg.std(dim=dims, keepdim=True)
    rescaled = noise_cfg * (std_text / std_cfg)
    return guidance_rescale * rescaled + (1 - guidance_rescale) * noise_cfg


class EulerOracle:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 timestep_spacing="linspace", steps_offset=0, prediction_type="epsilon", device_scalars=False):
        self.n_train = num_train_timesteps
        self.dev = device_scalars
        self.pred = prediction_type
        self.spacing, self.offset = timestep_spacing, steps_offset
        betas = make_betas(beta_schedule, beta_start, beta_end, num_train_timesteps)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).flip(0)
        self.sigmas = torch.cat([sig, torch.zeros(1)])
        self.step_index = 0

    @property
    def init_noise_sigma(self):
        m = self.sigmas.max()
        return m if self.spacing in ("linspace", "trailing") else (m ** 2 + 1) ** 0.5

    def set_timesteps(self, n):
        ts = spaced_timesteps(self.spacing, self.n_train, n, self.offset, as_float=True)
        base = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(base)), base)
        sig = np.concatenate([sig, [0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sig)
        self.timesteps = torch.from_numpy(ts.astype(np.float32))
        self.step_index = 0

    def scale_model_input(self, sample):
        sigma = self.sigmas[self.step_index]
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output, sample):
        sample = sample.to(torch.float32)
        sigma = self.sigmas[self.step_index]
        sigma_hat = sigma * (0.0 + 1)
        if self.pred in ("sample", "original_sample"):                     # scheduling_euler_discrete.py:760-775
            pred_original = model_output
        elif self.pred == "epsilon":
            pred_original = sample - smul(sigma_hat, model_output, self.dev)
        elif self.pred == "v_prediction":
            pred_original = model_output * (-sigma / (sigma ** 2 + 1) ** 0.5) + (sample / (sigma ** 2 + 1))
        else:
            raise ValueError(self.pred)
        derivative = (sample - pred_original) / sigma_hat
        dt = self.sigmas[self.step_index + 1] - sigma_hat
        prev = sample + derivative * dt
        self.step_index += 1
        return prev.to(model_output.dtype)


class DDIMOracle:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, timestep_spacing="leading",
                 clip_sample_range=1.0, prediction_type="epsilon", device_scalars=False):
        self.n_train = num_train_timesteps
        self.dev = device_scalars
        self.pred = prediction_type
        betas = make_betas(beta_schedule, beta_start, beta_end, num_train_timesteps)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.clip, self.clip_range = clip_sample, clip_sample_range
        self.spacing, self.offset = timestep_spacing, steps_offset
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        self.n_inf = n
        self.timesteps = torch.from_numpy(spaced_timesteps(self.spacing, self.n_train, n, self.offset, as_float=False))

    def step(self, model_output, timestep, sample, eta=0.0, variance_noise=None):
        t = int(timestep)
        prev_t = t - self.n_train // self.n_inf
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        if self.pred == "epsilon":                                          # scheduling_ddim.py:455-468
            x0 = (sample - smul(b_t ** 0.5, model_output, self.dev)) / a_t ** 0.5
            pred_eps = model_output
        elif self.pred == "sample":
            x0 = model_output
            pred_eps = (sample - smul(a_t ** 0.5, x0, self.dev)) / b_t ** 0.5
        elif self.pred == "v_prediction":
            x0 = smul(a_t ** 0.5, sample, self.dev) - smul(b_t ** 0.5, model_output, self.dev)
            pred_eps = smul(a_t ** 0.5, model_output, self.dev) + smul(b_t ** 0.5, sample, self.dev)
        else:
            raise ValueError(self.pred)
        if self.clip:
            x0 = x0.clamp(-self.clip_range, self.clip_range)
        variance = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
        std = eta * variance ** 0.5
        direction = smul((1 - a_prev - std ** 2) ** 0.5, pred_eps, self.dev)
        prev = smul(a_prev ** 0.5, x0, self.dev) + direction
        if eta > 0:
            prev = prev + smul(std, variance_noise, self.dev)
        return prev


class DDPMOracle:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 variance_type="fixed_small", clip_sample=True, clip_sample_range=1.0, timestep_spacing="leading",
                 steps_offset=0, prediction_type="epsilon", device_scalars=False):
        self.n_train = num_train_timesteps
        self.dev = device_scalars
        self.pred = prediction_type
        betas = make_betas(beta_schedule, beta_start, beta_end, num_train_timesteps)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.one = torch.tensor(1.0)
        self.variance_type = variance_type
        self.clip, self.clip_range = clip_sample, clip_sample_range
        self.spacing, self.offset = timestep_spacing, steps_offset
        self.n_inf = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        self.n_inf = n
        self.timesteps = torch.from_numpy(spaced_timesteps(self.spacing, self.n_train, n, self.offset, as_float=False))

    def _prev(self, t):
        """previous_timestep (scheduling_ddpm.py:648-668): the next entry of the schedule once set_timesteps has run
        (-1 after the last), t - 1 otherwise."""
        if not self.n_inf:
            return t - 1
        ts = [int(v) for v in self.timesteps]
        i = ts.index(int(t))
        return ts[i + 1] if i + 1 < len(ts) else -1

    def step(self, model_output, timestep, sample, generator=None, noise=None):
        t = int(timestep)
        prev_t = self._prev(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        if self.pred == "epsilon":                                          # scheduling_ddpm.py:505-517
            x0 = (sample - smul(b_t ** 0.5, model_output, self.dev)) / a_t ** 0.5
        elif self.pred == "sample":
            x0 = model_output
        elif self.pred == "v_prediction":
            x0 = smul(a_t ** 0.5, sample, self.dev) - smul(b_t ** 0.5, model_output, self.dev)
        else:
            raise ValueError(self.pred)
        if self.clip:
            x0 = x0.clamp(-self.clip_range, self.clip_range)
        k0 = (a_prev ** 0.5 * cur_b) / b_t
        kx = cur_a ** 0.5 * b_prev / b_t
        prev = smul(k0, x0, self.dev) + smul(kx, sample, self.dev)
        variance = 0
        if t > 0:
            if noise is None:
                noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_b, min=1e-20)
            if self.variance_type == "fixed_large":
                var = cur_b
            variance = smul(var ** 0.5, noise, self.dev)
        return prev + variance


class FlowMatchOracle:
    def __init__(self, num_train_timesteps=1000, shift=1.0, use_dynamic_shifting=False, device_scalars=False):
        self.n_train = num_train_timesteps
        self.dev = device_scalars
        self.shift, self.dynamic = shift, use_dynamic_shifting
        ts = torch.from_numpy(np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy())
        sig = ts / num_train_timesteps
        if not use_dynamic_shifting:
            sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigma_min, self.sigma_max = sig[-1].item(), sig[0].item()
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n=None, sigmas=None, mu=None):
        if sigmas is None:
            ts = np.linspace(self.sigma_max * self.n_train, self.sigma_min * self.n_train, n)
            sig = ts / self.n_train
        else:
            sig = np.array(sigmas).astype(np.float32)
        if self.dynamic:
            sig = math.exp(mu) / (math.exp(mu) + (1 / sig - 1) ** 1.0)
        else:
            sig = self.shift * sig / (1 + (self.shift - 1) * sig)
        sig = torch.from_numpy(np.asarray(sig)).to(torch.float32)
        self.timesteps = sig * self.n_train
        self.sigmas = torch.cat([sig, torch.zeros(1)])
        self.step_index = 0

    def step(self, model_output, sample):
        sample = sample.to(torch.float32)
        dt = self.sigmas[self.step_index + 1] - self.sigmas[self.step_index]
        prev = sample + smul(dt, model_output, self.dev)
        self.step_index += 1
        return prev.to(model_output.dtype)


class UniPCFlowOracle:
    """UniPCMultistepScheduler in the configuration Wan 2.1 ships (pipeline_wan.py:52-59): prediction_type
    "flow_prediction", use_flow_sigmas, flow_shift, solver_order 2, solver_type "bh2", predict_x0, lower_order_final.
    Restates schedulers/scheduling_unipc_multistep.py: set_timesteps :428-466 (flow sigmas), convert_model_output
    :760-831, multistep_uni_p_bh_update :833-960, multistep_uni_c_bh_update :962-1098, step :1153-1232."""

    def __init__(self, num_train_timesteps=1000, solver_order=2, flow_shift=1.0, solver_type="bh2", device_scalars=False):
        self.n_train, self.order_max, self.shift, self.solver_type = num_train_timesteps, solver_order, flow_shift, solver_type
        self.init_noise_sigma = 1.0
        # The reference keeps ``sigmas`` on the CPU (:466), so every coefficient below is a 0-d CPU fp32 tensor written
        # FIRST in its product: torch's CPU kernels round it to the tensor's dtype, the device kernels keep fp32
        # (module docstring).  ``(m - m0) / rk`` keeps the fp32 rk in both (CPU: original_scalar_value); the device
        # kernel's a * (1 / rk) shortcut (<= 1 ulp of the quotient) is not modelled -- the HIP kernel divides.
        self.dev = device_scalars

    def set_timesteps(self, n):
        sig = np.linspace(1, 1 / self.n_train, n + 1)[:-1]
        sig = self.shift * sig / (1 + (self.shift - 1) * sig)
        if np.fabs(sig[0] - 1) < 1e-6:
            sig[0] -= 1e-6
        ts = (sig * self.n_train).copy()
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts).to(dtype=torch.int64)
        self.model_outputs = [None] * self.order_max
        self.lower_order_nums = 0
        self.last_sample = None
        self.step_index = 0
        self.this_order = None

    @staticmethod
    def _alpha_sigma(sigma):
        return 1 - sigma, sigma

    def _coeffs(self, sigma_t, sigma_s0, hist_sigmas, order):
        alpha_t, sigma_t = self._alpha_sigma(sigma_t)
        alpha_s0, sigma_s0 = self._alpha_sigma(sigma_s0)
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        rks = []
        for s in hist_sigmas[: order - 1]:
            a_si, s_si = self._alpha_sigma(s)
            rks.append(((torch.log(a_si) - torch.log(s_si)) - lambda_s0) / h)
        rks.append(torch.ones(()))
        rks = torch.stack(rks)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = hh if self.solver_type == "bh1" else torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return alpha_t, sigma_t, sigma_s0, h_phi_1, B_h, rks, torch.stack(R), torch.stack(b)

    def step(self, model_output, sample):
        i = self.step_index
        x0 = sample - smul(self.sigmas[i], model_output, self.dev)          # convert_model_output, flow_prediction
        if i > 0 and self.last_sample is not None:                       # corrector (multistep_uni_c_bh_update)
            order = self.this_order
            m0 = self.model_outputs[-1]
            hist = [self.sigmas[i - (k + 1)] for k in range(1, order)]
            alpha_t, sigma_t, sigma_s0, h_phi_1, B_h, rks, R, b = self._coeffs(self.sigmas[i], self.sigmas[i - 1], hist, order)
            D1s = [(self.model_outputs[-(k + 1)] - m0) / rks[k - 1] for k in range(1, order)]
            rhos_c = torch.ones(1, dtype=sample.dtype) * 0.5 if order == 1 else torch.linalg.solve(R, b).to(sample.dtype)
            x_t_ = smul(sigma_t / sigma_s0, self.last_sample, self.dev) - smul(alpha_t * h_phi_1, m0, self.dev)
            corr = torch.einsum("k,bkc...->bc...", rhos_c[:-1], torch.stack(D1s, dim=1)) if D1s else 0
            sample = (x_t_ - smul(alpha_t * B_h, corr + rhos_c[-1] * (x0 - m0), self.dev)).to(sample.dtype)
        for k in range(self.order_max - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = x0
        this_order = min(self.order_max, len(self.timesteps) - i)
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        order = self.this_order                                           # predictor (multistep_uni_p_bh_update)
        m0 = self.model_outputs[-1]
        hist = [self.sigmas[i - k] for k in range(1, order)]
        alpha_t, sigma_t, sigma_s0, h_phi_1, B_h, rks, R, b = self._coeffs(self.sigmas[i + 1], self.sigmas[i], hist, order)
        D1s = [(self.model_outputs[-(k + 1)] - m0) / rks[k - 1] for k in range(1, order)]
        x_t_ = smul(sigma_t / sigma_s0, sample, self.dev) - smul(alpha_t * h_phi_1, m0, self.dev)
        if D1s:
            rhos_p = torch.ones(1, dtype=sample.dtype) * 0.5 if order == 2 else \
                torch.linalg.solve(R[:-1, :-1], b[:-1]).to(sample.dtype)
            pred = torch.einsum("k,bkc...->bc...", rhos_p, torch.stack(D1s, dim=1))
        else:
            pred = 0
        prev = (x_t_ - (smul(alpha_t * B_h, pred, self.dev) if D1s else 0)).to(sample.dtype)
        if self.lower_order_nums < self.order_max:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev
