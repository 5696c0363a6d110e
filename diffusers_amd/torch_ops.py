"""``torch.library`` registrations of the C-ABI kernels: ``torch.ops.mi355x.*``.

The reference binds its native attention kernels exactly this way (models/attention_dispatch.py:746-816: a
``torch.library.custom_op`` with ``device_types="cuda"`` around the native entry point plus a ``register_fake`` shape
function), which is what makes an op visible to ``torch.compile`` / ``torch.export`` (the fake implementation gives the
tracer output shapes and dtypes without running the kernel) and callable from inside any reference module's ``forward``.
Each op below is the ctypes call of :mod:`diffusers_amd.ops` -- device pointers, sizes and the CURRENT stream go to
``libdiffusers_amd.so`` -- behind a functional signature (tensors in, new tensor out; nothing mutated except where the
name ends in ``_``).  Registration happens on import of this module (``import diffusers_amd.torch_ops``);
``diffusers_amd.ops`` itself stays free of it so the eager engine path has no dispatcher overhead.

    y = torch.ops.mi355x.gemm(x, w, bias, 0)                   # F.linear + fused activation
    y = torch.ops.mi355x.conv2d_nhwc(x, w, bias, 3, 1, False)  # F.conv2d on channels-last activations
    o = torch.ops.mi355x.flash_attn(q, k, v, scale)            # F.scaled_dot_product_attention, (B, S, H, D) layout
    y = torch.ops.mi355x.groupnorm(x, gamma, beta, 32, 1e-5, True)
    y = torch.ops.mi355x.layernorm(x, gamma, beta, 1e-5)
    x_next = torch.ops.mi355x.euler_step(eps, x, table, step_idx, True, 5.0, 0)
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

NAMESPACE = "mi355x"
bf16 = torch.bfloat16


@torch.library.custom_op(f"{NAMESPACE}::gemm", mutates_args=(), device_types="cuda")
def gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act: int) -> torch.Tensor:
    """out[M][N] = act(x[M][K] @ w[N][K]^T + bias) -- nn.Linear (da_gemm_bf16, conv == 0); act = DA_ACT_* (GEGLU expects the
    packed weight of ops.pack_geglu and returns N / 2 columns)."""
    return ops.linear(x, w, bias, act=act)


@gemm.register_fake
def _(x, w, bias, act):
    n = w.shape[0] // 2 if act in (1, 6) else w.shape[0]   # DA_ACT_GEGLU / DA_ACT_GEGLU_TANH
    return x.new_empty((x.shape[0], n))


@torch.library.custom_op(f"{NAMESPACE}::conv2d_nhwc", mutates_args=(), device_types="cuda")
def conv2d_nhwc(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], ksize: int, stride: int,
                upsample2x: bool) -> torch.Tensor:
    """Implicit-GEMM Conv2d on channels-last [B][H][W][C] activations, weight [Cout][k*k*Cin] (ops.pack_conv_weight);
    ``upsample2x`` fuses F.interpolate(scale_factor=2, mode="nearest") into the gather."""
    return ops.conv2d_nhwc(x, w, bias, ksize=ksize, stride=stride, up=upsample2x)


@conv2d_nhwc.register_fake
def _(x, w, bias, ksize, stride, upsample2x):
    B, H, W_, _ = x.shape
    pad = (ksize - 1) // 2
    hv, wv = (2 * H, 2 * W_) if upsample2x else (H, W_)
    return x.new_empty((B, (hv + 2 * pad - ksize) // stride + 1, (wv + 2 * pad - ksize) // stride + 1, w.shape[0]))


@torch.library.custom_op(f"{NAMESPACE}::flash_attn", mutates_args=(), device_types="cuda")
def flash_attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float]) -> torch.Tensor:
    """softmax(scale * q k^T) v on (B, S, H, D) tensors -- the layout of the reference's attention backends
    (attention_dispatch.py:257-283); no mask, no dropout, not causal."""
    from .attention_backend import mi355x_flash_attention
    return mi355x_flash_attention(q, k, v, scale=scale)


@flash_attn.register_fake
def _(q, k, v, scale):
    return q.new_empty(q.shape)


@torch.library.custom_op(f"{NAMESPACE}::groupnorm", mutates_args=(), device_types="cuda")
def groupnorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float,
              silu: bool) -> torch.Tensor:
    """nn.GroupNorm (+ SiLU) on channels-last [B][H][W][C] / [B][HW][C] activations."""
    return ops.group_norm_nhwc(x, gamma, beta, groups, eps, silu=silu)


@groupnorm.register_fake
def _(x, gamma, beta, groups, eps, silu):
    return torch.empty_like(x)


@torch.library.custom_op(f"{NAMESPACE}::layernorm", mutates_args=(), device_types="cuda")
def layernorm(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float) -> torch.Tensor:
    """nn.LayerNorm over the last dim of a token matrix [M][C]."""
    return ops.layer_norm(x, gamma, beta, eps)


@layernorm.register_fake
def _(x, gamma, beta, eps):
    return torch.empty_like(x)


@torch.library.custom_op(f"{NAMESPACE}::euler_step", mutates_args=(), device_types="cuda")
def euler_step(model_output: torch.Tensor, sample: torch.Tensor, table: torch.Tensor, step_idx: torch.Tensor, cfg: bool,
               guidance_scale: float, pred_type: int) -> torch.Tensor:
    """EulerDiscreteScheduler.step (+ the CFG combine when ``cfg``: model_output = [uncond ; cond]) from the scheduler's
    device table; returns the new sample."""
    return ops.euler_step(model_output, sample, table, step_idx, cfg=cfg, guidance=guidance_scale, pred_type=pred_type)


@euler_step.register_fake
def _(model_output, sample, table, step_idx, cfg, guidance_scale, pred_type):
    return torch.empty_like(sample)


OPS = ("gemm", "conv2d_nhwc", "flash_attn", "groupnorm", "layernorm", "euler_step")
