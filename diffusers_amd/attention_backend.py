"""The two inner drop-in boundaries of SURVEY.md 8b for attention:

B4  ``mi355x_flash_attention`` -- an attention BACKEND function with the reference's backend signature and (B, S, H, D)
    layout (models/attention_dispatch.py:494-515; registered backends are called by ``dispatch_attention_fn`` from
    Flux / Wan and ~90 newer models).  ``register_backend()`` shows the registration a reference maintainer would add.
B3  ``MI355XAttnProcessor`` -- an attention PROCESSOR for the reference ``Attention`` module
    (models/attention_processor.py:52-309; contract of AttnProcessor2_0.__call__, :2696-2787): installed with
    ``model.set_attn_processor(MI355XAttnProcessor())`` it replaces q/k/v projections, SDPA and to_out of every
    attention layer of a reference UNet2DConditionModel / AutoencoderKL by the HIP kernels, leaving the rest of the
    reference module graph untouched.

Both raise for arguments the kernels do not implement (masks, dropout, causal, GQA, LSE); neither has a fallback.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .layers import KERNEL_HEAD_DIMS

bf16 = torch.bfloat16


def mi355x_flash_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                           attn_mask: Optional[torch.Tensor] = None, dropout_p: float = 0.0, is_causal: bool = False,
                           scale: Optional[float] = None, enable_gqa: bool = False, return_lse: bool = False,
                           _parallel_config=None) -> torch.Tensor:
    """query (B, Sq, H, D), key / value (B, Skv, H, D) bf16 HIP tensors -> (B, Sq, H, D)."""
    if attn_mask is not None or dropout_p != 0.0 or is_causal or enable_gqa or return_lse or _parallel_config is not None:
        raise ValueError("mi355x_flash_attention: attn_mask / dropout / causal / GQA / LSE / context parallel are not supported")
    if query.dim() != 4 or key.shape != value.shape or query.shape[0] != key.shape[0] or query.shape[2:] != key.shape[2:]:
        raise ValueError("mi355x_flash_attention: expected (B, S, H, D) tensors with matching batch / heads / head_dim")
    B, Sq, H, D = query.shape
    Skv = key.shape[1]
    if D not in KERNEL_HEAD_DIMS:
        raise ValueError(f"mi355x_flash_attention: head_dim {D} not in {KERNEL_HEAD_DIMS}")
    for t_, n in ((query, "query"), (key, "key"), (value, "value")):
        if t_.dtype != bf16 or not t_.is_cuda:
            raise ValueError(f"mi355x_flash_attention: {n} must be a bf16 HIP tensor")
    inner = H * D
    q2 = query.contiguous().view(B * Sq, inner)
    sa = (Skv + 7) // 8 * 8
    if sa == Skv:
        k2 = key.contiguous().view(B * Skv, inner)
        v2 = value.contiguous().view(B * Skv, inner)
    else:  # pad the key axis to a multiple of 8 (16-byte aligned V^T rows); the kernel masks keys >= Skv
        k2 = torch.zeros((B, sa, inner), device=key.device, dtype=bf16)
        v2 = torch.zeros((B, sa, inner), device=key.device, dtype=bf16)
        k2[:, :Skv] = key.reshape(B, Skv, inner)
        v2[:, :Skv] = value.reshape(B, Skv, inner)
        k2, v2 = k2.view(B * sa, inner), v2.view(B * sa, inner)
    vt = ops.transpose(v2)                                                    # [inner][B*sa]
    o = ops.attention(q2, k2, vt, B=B, H=H, D=D, Sq=Sq, Skv=Skv, Skv_alloc=sa, q_row_stride=inner, k_row_stride=inner,
                      q_batch_stride=Sq * inner, k_batch_stride=sa * inner, vt_ld=B * sa, vt_batch_stride=sa,
                      scale=scale)
    return o.view(B, Sq, H, D)


def register_backend(registry=None, name=None):
    """Register :func:`mi355x_flash_attention` with the reference's ``_AttentionBackendRegistry``
    (attention_dispatch.py:257-283).  ``AttentionBackendName`` is a closed Enum, so the function takes over an existing
    slot (default: ``AITER_FA2_HUB``), after which ``model.set_attention_backend("aiter_fa2_hub")`` /
    ``DIFFUSERS_ATTN_BACKEND=aiter_fa2_hub``
    routes every ``dispatch_attention_fn`` call to the HIP kernel."""
    if registry is None:
        from diffusers.models.attention_dispatch import AttentionBackendName, _AttentionBackendRegistry
        registry, name = _AttentionBackendRegistry, (name or AttentionBackendName.AITER_FA2_HUB)
    registry._backends[name] = mi355x_flash_attention
    registry._constraints[name] = []
    registry._supported_arg_names[name] = {"query", "key", "value", "attn_mask", "dropout_p", "is_causal", "scale",
                                           "enable_gqa", "return_lse", "_parallel_config"}
    return name


class MI355XAttnProcessor:
    """Processor for the reference ``Attention`` module, AttnProcessor2_0 contract (attention_processor.py:2705-2787):
    optional spatial 4-D input, optional group_norm, q/k/v Linear (optional bias), SDPA, to_out[0], residual,
    ``rescale_output_factor``.  Weights are read from the module (bf16, on the HIP device) at call time."""

    def __call__(self, attn, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, temb: Optional[torch.Tensor] = None) -> torch.Tensor:
        if attention_mask is not None:
            raise ValueError("MI355XAttnProcessor: attention_mask is not supported")
        if getattr(attn, "spatial_norm", None) is not None or getattr(attn, "norm_cross", None):
            raise ValueError("MI355XAttnProcessor: spatial_norm / norm_cross are not supported")
        if getattr(attn, "norm_q", None) is not None or getattr(attn, "norm_k", None) is not None:
            raise ValueError("MI355XAttnProcessor: q/k norms are not supported (use the model classes)")
        residual = hidden_states
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            Bn, Cn, Hn, Wn = hidden_states.shape
            hidden_states = hidden_states.view(Bn, Cn, Hn * Wn).transpose(1, 2)
        B, S, C = hidden_states.shape
        x = hidden_states.contiguous()
        gn = getattr(attn, "group_norm", None)
        if gn is not None:   # nn.GroupNorm over channels of the token tensor == channels-last GroupNorm
            x = ops.group_norm_nhwc(x, gn.weight, gn.bias, gn.num_groups, gn.eps)
        ctx = x if encoder_hidden_states is None else encoder_hidden_states.contiguous()
        Skv = ctx.shape[1]
        heads = attn.heads
        x2, c2 = x.view(B * S, C), ctx.view(B * Skv, ctx.shape[-1])
        q = ops.linear(x2, attn.to_q.weight, attn.to_q.bias)
        k = ops.linear(c2, attn.to_k.weight, attn.to_k.bias)
        v = ops.linear(c2, attn.to_v.weight, attn.to_v.bias)
        inner = q.shape[-1]
        D = inner // heads
        o = mi355x_flash_attention(q.view(B, S, heads, D), k.view(B, Skv, heads, D), v.view(B, Skv, heads, D),
                                   scale=getattr(attn, "scale", None))
        out = ops.linear(o.view(B * S, inner), attn.to_out[0].weight, attn.to_out[0].bias).view(B, S, -1)
        if input_ndim == 4:
            out = out.transpose(-1, -2).reshape(Bn, Cn, Hn, Wn)
        if getattr(attn, "residual_connection", False):
            out = out + residual
        rs = getattr(attn, "rescale_output_factor", 1.0)
        if rs != 1.0:
            out = out / rs
        return out
