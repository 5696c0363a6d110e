"""FluxTransformer2DModel on the gfx950 kernels.

Mirrors the reference class (models/transformers/transformer_flux.py:529-821) for ``guidance_embeds=False`` (FLUX.1-schnell,
BASELINE config 4): same constructor kwargs, ``state_dict`` keys / shapes, ``forward`` arguments and tensor shapes, same
``ValueError``s for arguments the engine does not implement.

Layout inside: one token-major joint buffer per call, rows of a batch ordered [text tokens ; image tokens] exactly as the
reference concatenates them for attention (transformer_flux.py:120-122), so the per-stream projections of the double
blocks write row slices of ONE q|k buffer and column slices of ONE V^T buffer and the reference's ``torch.cat`` / ``split``
copies (one (1, 4608, 3072) tensor per block) never happen.  Per block:

  AdaLN-Zero      Linear(SiLU(temb)) as a weight-streaming GEMV (da_linear_small_m_bf16), LayerNorm + (1+scale), shift
                  in one pass (da_layernorm_bf16)                                        normalization.py:130-202
  q|k, V^T        one GEMM for [to_q ; to_k] (+bias), one swapped GEMM for V^T (+row bias)  transformer_flux.py:43-57
  RMSNorm + RoPE  in place on the q|k buffer (da_rmsnorm_rope_bf16)                      transformer_flux.py:101-125
  attention       flash kernel over the joint sequence (da_attention_bf16), written straight into the column block of
                  the single blocks' [attn | mlp] buffer                                 transformer_flux.py:127-141,:403
  to_out / FF     GEMMs with bias + GELU-tanh / gate * (.) + residual epilogues          transformer_flux.py:400-405,:470-494
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from typing import Any, Dict, Optional

import torch

from . import _lib as L
from .config_utils import check_to
from .loading import PretrainedMixin
from . import ops
from .layers import Linear, TimestepEmbedding, Weights
from .unet_2d_condition import FrozenConfig

bf16 = torch.bfloat16


@dataclass
class Transformer2DModelOutput:
    sample: torch.Tensor


_DEFAULTS = dict(patch_size=1, in_channels=64, out_channels=None, num_layers=19, num_single_layers=38,
                 attention_head_dim=128, num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768,
                 guidance_embeds=False, axes_dims_rope=(16, 56, 56))


def rope_tables(ids: torch.Tensor, axes_dim, theta: float = 10000.0):
    """FluxPosEmbed (transformer_flux.py:500-526): cos / sin [S][sum(axes_dim)] fp32 from float64 frequencies, built on the
    host (a few hundred KB, once per call shape)."""
    pos = ids.detach().to("cpu", torch.float32)
    cos_out, sin_out = [], []
    for i, dim in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64) / dim))
        f = torch.outer(pos[:, i], freqs)
        cos_out.append(f.cos().repeat_interleave(2, dim=1).float())
        sin_out.append(f.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos_out, dim=-1).contiguous(), torch.cat(sin_out, dim=-1).contiguous()


class _QKV:
    """to_q / to_k fused into one [2C][C] GEMM operand, to_v kept for the swapped (V^T) product."""

    def __init__(self, w: Weights, pq: str, pk: str, pv: str):
        self.wqk = torch.cat([w.get(pq + ".weight"), w.get(pk + ".weight")], 0).contiguous()
        self.bqk = torch.cat([w.get(pq + ".bias"), w.get(pk + ".bias")], 0).contiguous()
        self.wv = w.get(pv + ".weight")
        self.bv = w.get(pv + ".bias")


class _DoubleBlock:
    def __init__(self, w: Weights, p: str):
        self.norm1 = Linear(w, p + ".norm1.linear")
        self.norm1_context = Linear(w, p + ".norm1_context.linear")
        self.img = _QKV(w, p + ".attn.to_q", p + ".attn.to_k", p + ".attn.to_v")
        self.txt = _QKV(w, p + ".attn.add_q_proj", p + ".attn.add_k_proj", p + ".attn.add_v_proj")
        self.norm_q, self.norm_k = w.get(p + ".attn.norm_q.weight"), w.get(p + ".attn.norm_k.weight")
        self.norm_added_q = w.get(p + ".attn.norm_added_q.weight")
        self.norm_added_k = w.get(p + ".attn.norm_added_k.weight")
        self.to_out = Linear(w, p + ".attn.to_out.0")
        self.to_add_out = Linear(w, p + ".attn.to_add_out")
        self.ff0, self.ff2 = Linear(w, p + ".ff.net.0.proj"), Linear(w, p + ".ff.net.2")
        self.ffc0, self.ffc2 = Linear(w, p + ".ff_context.net.0.proj"), Linear(w, p + ".ff_context.net.2")


class _SingleBlock:
    def __init__(self, w: Weights, p: str):
        self.norm = Linear(w, p + ".norm.linear")
        self.proj_mlp = Linear(w, p + ".proj_mlp")
        self.proj_out = Linear(w, p + ".proj_out")
        self.qkv = _QKV(w, p + ".attn.to_q", p + ".attn.to_k", p + ".attn.to_v")
        self.norm_q, self.norm_k = w.get(p + ".attn.norm_q.weight"), w.get(p + ".attn.norm_k.weight")


class FluxTransformer2DModel(PretrainedMixin):
    """Drop-in for the reference ``FluxTransformer2DModel`` (inference, bf16, HIP device only)."""

    def __init__(self, **kwargs):
        unknown = set(kwargs) - set(_DEFAULTS)
        if unknown:
            raise TypeError(f"FluxTransformer2DModel: unexpected config keys {sorted(unknown)}")
        cfg = dict(_DEFAULTS)
        cfg.update(kwargs)
        cfg["axes_dims_rope"] = tuple(cfg["axes_dims_rope"])
        self.config = FrozenConfig(cfg)
        c = self.config
        if c.guidance_embeds:
            raise ValueError("guidance_embeds=True (FLUX.1-dev) is not on the BASELINE hot path")
        if c.attention_head_dim not in (64, 128):
            raise ValueError("attention_head_dim must be 64 or 128 (flash kernel head sizes)")
        if sum(c.axes_dims_rope) != c.attention_head_dim:
            raise ValueError("sum(axes_dims_rope) must equal attention_head_dim")
        if c.patch_size != 1:
            raise ValueError("patch_size != 1 is not supported")
        self.out_channels = c.out_channels or c.in_channels
        self.inner_dim = c.num_attention_heads * c.attention_head_dim
        self.dtype = bf16
        self.device = None
        self._built = False
        self._rope_cache: Dict[Any, Any] = {}

    # ------------------------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], device="cuda", strict: bool = True):
        c = self.config
        w = Weights(state_dict, device)
        self.device = torch.device(device)
        self.time_embedder = TimestepEmbedding(w, "time_text_embed.timestep_embedder")
        self.text_embedder = TimestepEmbedding(w, "time_text_embed.text_embedder")  # Linear -> SiLU -> Linear as well
        self.context_embedder = Linear(w, "context_embedder")
        self.x_embedder = Linear(w, "x_embedder")
        self.double = [_DoubleBlock(w, f"transformer_blocks.{i}") for i in range(c.num_layers)]
        self.single = [_SingleBlock(w, f"single_transformer_blocks.{i}") for i in range(c.num_single_layers)]
        self.norm_out = Linear(w, "norm_out.linear")
        self.proj_out = Linear(w, "proj_out")
        if strict and w.unused():
            raise RuntimeError(f"unexpected keys in state_dict: {w.unused()[:8]} ...")
        self._built = True
        return self

    def to(self, *args, **kwargs):
        return check_to(self, args, kwargs)

    def eval(self):
        return self

    def cache_context(self, name):  # pipelines call `with transformer.cache_context("cond")` (models/cache_utils.py:155)
        import contextlib
        return contextlib.nullcontext()

    # ------------------------------------------------------------------------------------------------------------
    # step-invariant work: pooled-text embedding MLP, RoPE tables
    # ------------------------------------------------------------------------------------------------------------
    def precompute_conditioning(self, pooled_projections: torch.Tensor, img_ids: torch.Tensor,
                                txt_ids: torch.Tensor) -> Dict[str, Any]:
        ops.require_hip(pooled_projections, "pooled_projections")
        if txt_ids.ndim == 3:
            txt_ids = txt_ids[0]
        if img_ids.ndim == 3:
            img_ids = img_ids[0]
        # the key must IDENTIFY the id tensors: (shape, sum) does not -- a 48x84 and an 84x48 grid have the same count
        # and the same coordinate sum -- so it is a digest of the exact id values (a few tens of KB, once per call)
        ids_host = torch.cat((txt_ids.detach().to("cpu", torch.float32), img_ids.detach().to("cpu", torch.float32)), dim=0)
        key = (tuple(img_ids.shape), tuple(txt_ids.shape),
               hashlib.sha1(ids_host.contiguous().numpy().tobytes()).hexdigest())
        if key not in self._rope_cache:
            cos, sin = rope_tables(ids_host, self.config.axes_dims_rope)
            self._rope_cache = {key: (cos.to(self.device), sin.to(self.device))}
        cos, sin = self._rope_cache[key]
        pooled_emb = self.text_embedder(pooled_projections.contiguous())  # (B, inner)
        return {"pooled_emb": pooled_emb, "cos": cos, "sin": sin, "St": int(txt_ids.shape[0]),
                "Si": int(img_ids.shape[0]), "batch": int(pooled_projections.shape[0])}

    # ------------------------------------------------------------------------------------------------------------
    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                pooled_projections: torch.Tensor = None, timestep: torch.Tensor = None, img_ids: torch.Tensor = None,
                txt_ids: torch.Tensor = None, guidance: torch.Tensor = None, joint_attention_kwargs=None,
                controlnet_block_samples=None, controlnet_single_block_samples=None, return_dict: bool = True,
                controlnet_blocks_repeat: bool = False, conditioning: Optional[Dict[str, Any]] = None,
                sampler_table=None, step_idx=None):
        """Reference signature (transformer_flux.py:671-685) plus the engine extensions ``conditioning`` (result of
        :meth:`precompute_conditioning`) and ``sampler_table`` / ``step_idx`` (the sinusoid reads the model timestep
        from column 7 of the device-resident sampler table: HIP-graph replayable)."""
        if not self._built:
            raise RuntimeError("FluxTransformer2DModel: call load_state_dict() first")
        for name, v in (("guidance", guidance), ("controlnet_block_samples", controlnet_block_samples),
                        ("controlnet_single_block_samples", controlnet_single_block_samples)):
            if v is not None:
                raise ValueError(f"diffusers_amd FluxTransformer2DModel.forward: `{name}` is not supported on the HIP path")
        if joint_attention_kwargs:
            raise ValueError("diffusers_amd FluxTransformer2DModel.forward: `joint_attention_kwargs` is not supported")
        ops.require_hip(hidden_states, "hidden_states")
        c = self.config
        C, Hh, D = self.inner_dim, c.num_attention_heads, c.attention_head_dim
        B, Si, Cin = hidden_states.shape
        if Cin != c.in_channels:
            raise ValueError(f"hidden_states last dim {Cin} != in_channels {c.in_channels}")
        if conditioning is None:
            conditioning = self.precompute_conditioning(pooled_projections.to(device=self.device, dtype=bf16), img_ids,
                                                        txt_ids)
        St = conditioning["St"]
        if conditioning["Si"] != Si or conditioning["batch"] != B or encoder_hidden_states.shape[1] != St:
            raise ValueError("conditioning does not match hidden_states / encoder_hidden_states")
        if St % 8 or Si % 8:
            raise ValueError("text and image sequence lengths must be multiples of 8 (16-byte aligned V^T rows)")
        S = St + Si
        cos, sin = conditioning["cos"], conditioning["sin"]
        dev = hidden_states.device

        # ---- temb (CombinedTimestepTextProjEmbeddings, embeddings.py:1585-1601) ----
        if sampler_table is not None:
            t_emb = ops.timestep_embedding(None, 256, batch=B, flip_sin_to_cos=True, shift=0.0, table=sampler_table,
                                           step_idx=step_idx)
        else:
            # reference: timestep.to(hidden dtype) * 1000 (bf16 arithmetic), then the fp32 sinusoid
            t = (timestep.to(device=dev, dtype=bf16).reshape(-1) * 1000).float()
            if t.numel() == 1:
                t = t.expand(B)
            t_emb = ops.timestep_embedding(t.contiguous(), 256, batch=B, flip_sin_to_cos=True, shift=0.0)
        temb = self.time_embedder(t_emb, residual=conditioning["pooled_emb"])  # (B, C)

        def rows(buf, b, which):
            lo = b * S + (0 if which == "txt" else St)
            return buf[lo: lo + (St if which == "txt" else Si)]

        # ---- joint hidden state: [txt ; img] rows per batch ----
        Hbuf = torch.empty((B * S, C), device=dev, dtype=bf16)
        ehs2 = encoder_hidden_states.to(device=dev, dtype=bf16).contiguous()
        hs2 = hidden_states.contiguous()
        for b in range(B):
            ops.linear(ehs2[b], self.context_embedder.weight, self.context_embedder.bias, out=rows(Hbuf, b, "txt"))
            ops.linear(hs2[b], self.x_embedder.weight, self.x_embedder.bias, out=rows(Hbuf, b, "img"))

        def chunk(t_, i):
            return t_[:, i * C:(i + 1) * C]

        def attention(qk, vt, out=None):
            return ops.attention(qk, qk[:, C:], vt, B=B, H=Hh, D=D, Sq=S, Skv=S, Skv_alloc=S, q_row_stride=2 * C,
                                 k_row_stride=2 * C, q_batch_stride=S * 2 * C, k_batch_stride=S * 2 * C, vt_ld=B * S,
                                 vt_batch_stride=S, scale=D ** -0.5, out=out)

        # ---- double-stream blocks (transformer_flux.py:443-497) ----
        for blk in self.double:
            mi = ops.linear_small_m(temb, blk.norm1.weight, blk.norm1.bias, act_in=L.ACT_SILU)            # (B, 6C)
            mc = ops.linear_small_m(temb, blk.norm1_context.weight, blk.norm1_context.bias, act_in=L.ACT_SILU)
            qk = torch.empty((B * S, 2 * C), device=dev, dtype=bf16)
            vt = torch.empty((C, B * S), device=dev, dtype=bf16)
            for b in range(B):
                for which, m, pw, nq, nk in (("img", mi, blk.img, blk.norm_q, blk.norm_k),
                                             ("txt", mc, blk.txt, blk.norm_added_q, blk.norm_added_k)):
                    x = rows(Hbuf, b, which)
                    n = ops.layer_norm(x, None, None, 1e-6, mod_scale=chunk(m, 1)[b:b + 1],
                                       mod_shift=chunk(m, 0)[b:b + 1], rows_per_batch=x.shape[0])
                    qk_rows = rows(qk, b, which)
                    ops.linear(n, pw.wqk, pw.bqk, out=qk_rows)
                    lo = b * S + (0 if which == "txt" else St)
                    ops.linear(pw.wv, n, bias_rows=pw.bv, out=vt[:, lo: lo + x.shape[0]])
                    ops.rmsnorm_rope_(qk_rows, heads=Hh, head_dim=D, col_offsets=(0, C), weights=(nq, nk), eps=1e-6,
                                      cos=cos, sin=sin, rope_row0=(0 if which == "txt" else St))
            o = attention(qk, vt)
            H2 = torch.empty_like(Hbuf)
            H3 = torch.empty_like(Hbuf)
            for b in range(B):
                for which, m, to_out, f0, f2 in (("img", mi, blk.to_out, blk.ff0, blk.ff2),
                                                 ("txt", mc, blk.to_add_out, blk.ffc0, blk.ffc2)):
                    x, n_rows = rows(Hbuf, b, which), (St if which == "txt" else Si)
                    ops.linear(rows(o, b, which), to_out.weight, to_out.bias, gate=chunk(m, 2)[b:b + 1],
                               rows_per_batch=n_rows, residual=x, out=rows(H2, b, which))
                    x2 = rows(H2, b, which)
                    n2 = ops.layer_norm(x2, None, None, 1e-6, mod_scale=chunk(m, 4)[b:b + 1],
                                        mod_shift=chunk(m, 3)[b:b + 1], rows_per_batch=n_rows)
                    h = ops.linear(n2, f0.weight, f0.bias, act=L.ACT_GELU_TANH)
                    ops.linear(h, f2.weight, f2.bias, gate=chunk(m, 5)[b:b + 1], rows_per_batch=n_rows, residual=x2,
                               out=rows(H3, b, which))
            Hbuf = H3

        # ---- single-stream blocks (transformer_flux.py:383-412) on the joint sequence ----
        for blk in self.single:
            m = ops.linear_small_m(temb, blk.norm.weight, blk.norm.bias, act_in=L.ACT_SILU)              # (B, 3C)
            n = ops.layer_norm(Hbuf, None, None, 1e-6, mod_scale=chunk(m, 1), mod_shift=chunk(m, 0), rows_per_batch=S)
            cat = torch.empty((B * S, 5 * C), device=dev, dtype=bf16)                                     # [attn | mlp]
            ops.linear(n, blk.proj_mlp.weight, blk.proj_mlp.bias, act=L.ACT_GELU_TANH, out=cat[:, C:])
            qk = ops.linear(n, blk.qkv.wqk, blk.qkv.bqk)
            vt = ops.linear(blk.qkv.wv, n, bias_rows=blk.qkv.bv)
            ops.rmsnorm_rope_(qk, heads=Hh, head_dim=D, col_offsets=(0, C), weights=(blk.norm_q, blk.norm_k), eps=1e-6,
                              cos=cos, sin=sin, rope_row0=0, rows_per_batch=S)
            attention(qk, vt, out=cat[:, :C])
            Hbuf = ops.linear(cat, blk.proj_out.weight, blk.proj_out.bias, gate=chunk(m, 2), rows_per_batch=S,
                              residual=Hbuf)

        # ---- AdaLayerNormContinuous (scale first, then shift; normalization.py:346-351) + proj_out on the image rows ----
        emb = ops.linear_small_m(temb, self.norm_out.weight, self.norm_out.bias, act_in=L.ACT_SILU)       # (B, 2C)
        out = torch.empty((B, Si, self.proj_out.weight.shape[0]), device=dev, dtype=bf16)
        for b in range(B):
            x = rows(Hbuf, b, "img")
            n = ops.layer_norm(x, None, None, 1e-6, mod_scale=emb[b:b + 1, :C], mod_shift=emb[b:b + 1, C:],
                               rows_per_batch=Si)
            ops.linear(n, self.proj_out.weight, self.proj_out.bias, out=out[b])
        if not return_dict:
            return (out,)
        return Transformer2DModelOutput(sample=out)
