"""WanTransformer3DModel (Wan 2.1 T2V) on the gfx950 kernels.

Mirrors the reference class (models/transformers/transformer_wan.py:523-735) for the text-to-video configuration of
BASELINE config 5 (no image branch): same constructor kwargs, ``state_dict`` keys / shapes, ``forward`` arguments
((B, C, F, H, W) latents in and out), same ``ValueError``s for arguments the engine does not implement.

Per call: the Conv3d patch embedding (kernel == stride) is a gather kernel + GEMM (K = 16*1*2*2 = 64); per block
  fp32 LayerNorm * (1 + scale) + shift with the fp32 ``scale_shift_table + temb`` vectors      transformer_wan.py:483-488
  [to_q ; to_k] GEMM, swapped GEMM for V^T (+ row bias), RMSNorm ACROSS heads + RoPE in place   :95-119
  flash attention over the 32 760-token sequence (71 % of the model's FLOPs)                    :144-155
  to_out GEMM with the fp32 gate and the residual in its epilogue                               :491
  affine fp32 LayerNorm, cross-attention to the text tokens (K / V^T hoisted: step invariant)   :494-496
  FFN GELU-tanh GEMM, down GEMM with fp32 gate + residual                                       :499-502
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch

from . import _lib as L
from .config_utils import check_to
from .loading import PretrainedMixin
from . import ops
from .layers import LayerNorm, Linear, TimestepEmbedding, Weights
from .transformer_flux import Transformer2DModelOutput
from .unet_2d_condition import FrozenConfig

bf16 = torch.bfloat16

_DEFAULTS = dict(patch_size=(1, 2, 2), num_attention_heads=40, attention_head_dim=128, in_channels=16, out_channels=16,
                 text_dim=4096, freq_dim=256, ffn_dim=13824, num_layers=40, cross_attn_norm=True,
                 qk_norm="rms_norm_across_heads", eps=1e-6, image_dim=None, added_kv_proj_dim=None, rope_max_seq_len=1024,
                 pos_embed_seq_len=None)


def wan_rope_tables(head_dim: int, frames: int, height: int, width: int, max_seq_len: int, theta: float = 10000.0):
    """WanRotaryPosEmbed (transformer_wan.py:354-416): cos / sin fp32 [frames*height*width][head_dim] (float64 freqs)."""
    h_dim = w_dim = 2 * (head_dim // 6)
    t_dim = head_dim - h_dim - w_dim
    if max(frames, height, width) > max_seq_len:
        raise ValueError("latent grid exceeds rope_max_seq_len")
    tabs = []
    for dim in (t_dim, h_dim, w_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64) / dim))
        f = torch.outer(torch.arange(max_seq_len), freqs)
        tabs.append((f.cos().repeat_interleave(2, dim=1).float(), f.sin().repeat_interleave(2, dim=1).float()))

    def grid(i):
        ct = tabs[0][i][:frames].view(frames, 1, 1, -1).expand(frames, height, width, -1)
        ch = tabs[1][i][:height].view(1, height, 1, -1).expand(frames, height, width, -1)
        cw = tabs[2][i][:width].view(1, 1, width, -1).expand(frames, height, width, -1)
        return torch.cat([ct, ch, cw], dim=-1).reshape(frames * height * width, head_dim).contiguous()
    return grid(0), grid(1)


class _Block:
    def __init__(self, w: Weights, p: str, cross_attn_norm: bool, eps: float):
        a = p + ".attn1"
        self.wqk = torch.cat([w.get(a + ".to_q.weight"), w.get(a + ".to_k.weight")], 0).contiguous()
        self.bqk = torch.cat([w.get(a + ".to_q.bias"), w.get(a + ".to_k.bias")], 0).contiguous()
        self.wv, self.bv = w.get(a + ".to_v.weight"), w.get(a + ".to_v.bias")
        self.norm_q, self.norm_k = w.get(a + ".norm_q.weight"), w.get(a + ".norm_k.weight")
        self.to_out = Linear(w, a + ".to_out.0")
        c = p + ".attn2"
        self.c_q = Linear(w, c + ".to_q")
        self.c_k, self.c_v = Linear(w, c + ".to_k"), Linear(w, c + ".to_v")
        self.c_norm_q, self.c_norm_k = w.get(c + ".norm_q.weight"), w.get(c + ".norm_k.weight")
        self.c_out = Linear(w, c + ".to_out.0")
        self.norm2 = LayerNorm(w, p + ".norm2", eps=eps) if cross_attn_norm else None
        self.ffn0, self.ffn2 = Linear(w, p + ".ffn.net.0.proj"), Linear(w, p + ".ffn.net.2")
        self.table = w.get_f32(p + ".scale_shift_table").reshape(-1).contiguous()     # fp32 [6*dim]


class WanTransformer3DModel(PretrainedMixin):
    """Drop-in for the reference ``WanTransformer3DModel`` (T2V, inference, bf16, HIP device only)."""

    def __init__(self, **kwargs):
        unknown = set(kwargs) - set(_DEFAULTS)
        if unknown:
            raise TypeError(f"WanTransformer3DModel: unexpected config keys {sorted(unknown)}")
        cfg = dict(_DEFAULTS)
        cfg.update(kwargs)
        cfg["patch_size"] = tuple(cfg["patch_size"])
        self.config = FrozenConfig(cfg)
        c = self.config
        if c.image_dim is not None or c.added_kv_proj_dim is not None or c.pos_embed_seq_len is not None:
            raise ValueError("the image-conditioned (I2V) branch is not on the BASELINE hot path")
        if c.qk_norm != "rms_norm_across_heads":
            raise ValueError("only qk_norm='rms_norm_across_heads' is supported")
        if c.attention_head_dim not in (64, 128):
            raise ValueError("attention_head_dim must be 64 or 128 (flash kernel head sizes)")
        k = c.in_channels * c.patch_size[0] * c.patch_size[1] * c.patch_size[2]
        if k % 64:
            raise ValueError("in_channels * prod(patch_size) must be a multiple of 64 (one K slice of the patch GEMM)")
        self.inner_dim = c.num_attention_heads * c.attention_head_dim
        self.dtype = bf16
        self.device = None
        self._built = False
        self._rope_cache: Dict[Any, Any] = {}

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], device="cuda", strict: bool = True):
        c = self.config
        w = Weights(state_dict, device)
        self.device = torch.device(device)
        pw = w.get("patch_embedding.weight")
        self.patch_w = pw.reshape(pw.shape[0], -1).contiguous()        # [inner][C*pt*ph*pw]
        self.patch_b = w.get("patch_embedding.bias")
        self.time_embedder = TimestepEmbedding(w, "condition_embedder.time_embedder")
        self.time_proj = Linear(w, "condition_embedder.time_proj")
        self.text1 = Linear(w, "condition_embedder.text_embedder.linear_1")
        self.text2 = Linear(w, "condition_embedder.text_embedder.linear_2")
        self.blocks = [_Block(w, f"blocks.{i}", c.cross_attn_norm, c.eps) for i in range(c.num_layers)]
        self.proj_out = Linear(w, "proj_out")
        self.table = w.get_f32("scale_shift_table").reshape(-1).contiguous()          # fp32 [2*dim]
        if strict and w.unused():
            raise RuntimeError(f"unexpected keys in state_dict: {w.unused()[:8]} ...")
        self._built = True
        return self

    def to(self, *args, **kwargs):
        return check_to(self, args, kwargs)

    def eval(self):
        return self

    def cache_context(self, name):
        import contextlib
        return contextlib.nullcontext()

    # ------------------------------------------------------------------------------------------------------------
    def precompute_conditioning(self, encoder_hidden_states: torch.Tensor) -> Dict[str, Any]:
        """Step-invariant work: the text embedder MLP and every block's cross-attention K (RMS-normed) / V^T."""
        ops.require_hip(encoder_hidden_states, "encoder_hidden_states")
        c = self.config
        B, St, Td = encoder_hidden_states.shape
        if St % 8:
            raise ValueError("text sequence length must be a multiple of 8 (16-byte aligned V^T rows)")
        C, Hh, D = self.inner_dim, c.num_attention_heads, c.attention_head_dim
        e2 = encoder_hidden_states.reshape(B * St, Td).contiguous()
        ctx = ops.linear(e2, self.text1.weight, self.text1.bias, act=L.ACT_GELU_TANH)
        ctx = ops.linear(ctx, self.text2.weight, self.text2.bias)
        kvs = []
        for blk in self.blocks:
            k = ops.linear(ctx, blk.c_k.weight, blk.c_k.bias)
            ops.rmsnorm_rope_(k, heads=Hh, head_dim=D, col_offsets=(0,), weights=(blk.c_norm_k,), eps=c.eps,
                              norm="across_heads")
            vt = ops.linear(blk.c_v.weight, ctx, bias_rows=blk.c_v.bias)             # [C][B*St]
            kvs.append((k, vt))
        return {"kvs": kvs, "St": St, "batch": B}

    def _rope(self, f, h, w):
        key = (f, h, w)
        if key not in self._rope_cache:
            cos, sin = wan_rope_tables(self.config.attention_head_dim, f, h, w, self.config.rope_max_seq_len)
            self._rope_cache = {key: (cos.to(self.device), sin.to(self.device))}
        return self._rope_cache[key]

    # ------------------------------------------------------------------------------------------------------------
    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    def forward(self, hidden_states: torch.Tensor, timestep: torch.Tensor = None, encoder_hidden_states: torch.Tensor = None,
                encoder_hidden_states_image: Optional[torch.Tensor] = None, return_dict: bool = True,
                attention_kwargs: Optional[Dict[str, Any]] = None, conditioning: Optional[Dict[str, Any]] = None,
                sampler_table=None, step_idx=None):
        """Reference signature (transformer_wan.py:629-637) plus ``conditioning`` / ``sampler_table`` / ``step_idx``
        (see FluxTransformer2DModel.forward)."""
        if not self._built:
            raise RuntimeError("WanTransformer3DModel: call load_state_dict() first")
        if encoder_hidden_states_image is not None:
            raise ValueError("diffusers_amd WanTransformer3DModel.forward: `encoder_hidden_states_image` is not supported")
        if attention_kwargs:
            raise ValueError("diffusers_amd WanTransformer3DModel.forward: `attention_kwargs` is not supported")
        ops.require_hip(hidden_states, "hidden_states")
        if timestep is not None and torch.is_tensor(timestep) and timestep.ndim == 2:
            raise ValueError("per-token timesteps (Wan 2.2 TI2V) are not supported")
        c = self.config
        C, Hh, D, eps = self.inner_dim, c.num_attention_heads, c.attention_head_dim, c.eps
        B, Cin, Fr, H, W_ = hidden_states.shape
        pt, ph, pw = c.patch_size
        if Cin != c.in_channels or Fr % pt or H % ph or W_ % pw:
            raise ValueError("hidden_states shape does not match in_channels / patch_size")
        f, h, w = Fr // pt, H // ph, W_ // pw
        S = f * h * w
        if S % 8:
            raise ValueError("token count must be a multiple of 8 (16-byte aligned V^T rows)")
        if conditioning is None:
            conditioning = self.precompute_conditioning(encoder_hidden_states.to(device=self.device, dtype=bf16))
        if conditioning["batch"] != B:
            raise ValueError("conditioning batch does not match hidden_states batch")
        St = conditioning["St"]
        cos, sin = self._rope(f, h, w)
        dev = hidden_states.device

        # ---- condition embedder (transformer_wan.py:331-351) ----
        if sampler_table is not None:
            t_emb = ops.timestep_embedding(None, c.freq_dim, batch=B, flip_sin_to_cos=True, shift=0.0,
                                           table=sampler_table, step_idx=step_idx)
        else:
            t = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
            if t.numel() == 1:
                t = t.expand(B)
            t_emb = ops.timestep_embedding(t.contiguous(), c.freq_dim, batch=B, flip_sin_to_cos=True, shift=0.0)
        temb = self.time_embedder(t_emb)                                                                   # (B, C)
        tproj = ops.linear_small_m(temb, self.time_proj.weight, self.time_proj.bias, act_in=L.ACT_SILU)    # (B, 6C)

        # ---- patch embedding ----
        tok = ops.patchify3d(hidden_states.contiguous(), c.patch_size)
        x = ops.linear(tok, self.patch_w, self.patch_b)                                                    # [B*S][C]

        def chunk(m, i):
            return m[:, i * C:(i + 1) * C]

        for blk, (ck, cvt) in zip(self.blocks, conditioning["kvs"]):
            mod = ops.bcast_add_f32(blk.table, tproj)                     # fp32 (B, 6C): shift, scale, gate, c_shift, ...
            n = ops.layer_norm(x, None, None, eps, mod_scale=chunk(mod, 1), mod_shift=chunk(mod, 0), rows_per_batch=S)
            qk = ops.linear(n, blk.wqk, blk.bqk)
            vt = ops.linear(blk.wv, n, bias_rows=blk.bv)
            ops.rmsnorm_rope_(qk, heads=Hh, head_dim=D, col_offsets=(0, C), weights=(blk.norm_q, blk.norm_k), eps=eps,
                              cos=cos, sin=sin, rope_row0=0, rows_per_batch=S, norm="across_heads")
            o = ops.attention(qk, qk[:, C:], vt, B=B, H=Hh, D=D, Sq=S, Skv=S, Skv_alloc=S, q_row_stride=2 * C,
                              k_row_stride=2 * C, q_batch_stride=S * 2 * C, k_batch_stride=S * 2 * C, vt_ld=B * S,
                              vt_batch_stride=S, scale=D ** -0.5)
            x = ops.linear(o, blk.to_out.weight, blk.to_out.bias, gate=chunk(mod, 2), rows_per_batch=S, residual=x)
            n2 = blk.norm2(x) if blk.norm2 is not None else x
            q = ops.linear(n2, blk.c_q.weight, blk.c_q.bias)
            ops.rmsnorm_rope_(q, heads=Hh, head_dim=D, col_offsets=(0,), weights=(blk.c_norm_q,), eps=eps,
                              norm="across_heads")
            o2 = ops.attention(q, ck, cvt, B=B, H=Hh, D=D, Sq=S, Skv=St, Skv_alloc=St, q_row_stride=C, k_row_stride=C,
                               q_batch_stride=S * C, k_batch_stride=St * C, vt_ld=B * St, vt_batch_stride=St,
                               scale=D ** -0.5)
            x = ops.linear(o2, blk.c_out.weight, blk.c_out.bias, residual=x)
            n3 = ops.layer_norm(x, None, None, eps, mod_scale=chunk(mod, 4), mod_shift=chunk(mod, 3), rows_per_batch=S)
            hmid = ops.linear(n3, blk.ffn0.weight, blk.ffn0.bias, act=L.ACT_GELU_TANH)
            x = ops.linear(hmid, blk.ffn2.weight, blk.ffn2.bias, gate=chunk(mod, 5), rows_per_batch=S, residual=x)

        # ---- output norm (shift, scale = table + temb; transformer_wan.py:713-723), projection, unpatchify ----
        shift = ops.bcast_add_f32(self.table[:C], temb)
        scale = ops.bcast_add_f32(self.table[C:], temb)
        n = ops.layer_norm(x, None, None, eps, mod_scale=scale, mod_shift=shift, rows_per_batch=S)
        y = ops.linear(n, self.proj_out.weight, self.proj_out.bias)                    # [B*S][pt*ph*pw*Cout]
        out = ops.unpatchify3d(y, (B, c.out_channels, Fr, H, W_), c.patch_size)
        if not return_dict:
            return (out,)
        return Transformer2DModelOutput(sample=out)
