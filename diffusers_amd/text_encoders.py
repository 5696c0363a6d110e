"""Text encoders on the engine's kernels (the half that round 1 left to PyTorch): the CLIP text
transformers of SD / SDXL / Flux and the T5 / UMT5 encoders of Flux / Wan, as drop-ins for the ``transformers`` modules the
reference pipelines hold in their ``text_encoder*`` slots (pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl.py:283-520
calls ``text_encoder(ids, output_hidden_states=True)`` and reads ``[0]`` / ``.hidden_states[-2]``; pipelines/flux/pipeline_flux.py:
219-399 reads ``.pooler_output`` and ``[0]``; pipelines/wan/pipeline_wan.py:164-215 reads ``.last_hidden_state``).

Same constructor configs, same ``state_dict`` key names and shapes, same call signatures and output attribute names as
``transformers`` (the reference's third-party dependency for this part; restated from its modeling_clip.py / modeling_t5.py /
modeling_umt5.py, version 5.x as installed: CLIPTextTransformer.forward, CLIPEncoderLayer, T5Stack / T5Block / T5Attention
incl. ``_relative_position_bucket``, T5LayerNorm, T5DenseGatedActDense).  Every matrix product, attention, normalisation
and activation is one of the engine's HIP kernels:

  * Q|K projection + swapped V^T projection in one paired launch, flash attention with the MASKED variant (causal for CLIP;
    additive relative-position bias, scale 1 and key-padding mask for T5 / UMT5);
  * fc1 + quick_gelu / gelu and wi_0 / wi_1 + gated tanh-GELU in the GEMM epilogue (DA_ACT_QUICK_GELU, DA_ACT_GEGLU_TANH);
  * LayerNorm / T5LayerNorm (da_layernorm_bf16 / da_rmsnorm_bf16); bias + residual in the GEMM epilogue.
Only the embedding gather (``F.embedding``) and the EOS-row pick are torch indexing ops (memory moves, no arithmetic).
Head size must be 64 (CLIP-L / CLIP-bigG / T5-XXL / UMT5-XXL all are).
"""
from __future__ import annotations

import math
from typing import Any, Dict, Mapping, Optional

import torch

from . import _lib as L
from . import ops
from .config_utils import check_to
from .layers import Weights
from .unet_2d_condition import FrozenConfig

bf16 = torch.bfloat16


class ModelOutput(dict):
    """Attribute + integer access over the non-None fields, like transformers' ModelOutput (``out[0]`` is the first
    field that is set, ``out.hidden_states`` the named one)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __getitem__(self, k):
        if isinstance(k, int):
            return [v for v in self.values() if v is not None][k]
        return super().__getitem__(k)

    def to_tuple(self):
        return tuple(v for v in self.values() if v is not None)


def _cfg(config, defaults: Dict[str, Any]) -> FrozenConfig:
    src = config if isinstance(config, Mapping) else (config.to_dict() if hasattr(config, "to_dict") else vars(config))
    return FrozenConfig({k: src.get(k, d) for k, d in defaults.items()})


def _pad_rows(S: int) -> int:
    return ((S + 15) // 16) * 16          # 16-byte aligned V^T rows, as layers.pad_encoder_states


class _SelfAttention:
    """q / k / v / out projections + flash attention over [B * S_alloc][C] token matrices."""

    def __init__(self, wq, wk, wv, wo, bq=None, bk=None, bv=None, bo=None, heads: int = 1, scale: Optional[float] = None):
        inner = wq.shape[0]
        if inner % heads or inner // heads != 64:
            raise ValueError("text encoder attention: head size must be 64 (the masked flash kernel's size)")
        self.heads, self.inner = heads, inner
        self.wqk = torch.cat([wq, wk], 0).contiguous()
        self.bqk = torch.cat([bq, bk], 0).contiguous() if bq is not None else None
        self.wv, self.bv, self.wo, self.bo = wv, bv, wo, bo
        self.scale = 64 ** -0.5 if scale is None else scale

    def __call__(self, x, B, S, S_alloc, residual, causal=False, bias=None):
        """x: [B * S_alloc][C] (rows S .. S_alloc of every batch are padding: never attended, outputs ignored)."""
        qk, vt = ops.linear_pair({"x": x, "w": self.wqk, "bias": self.bqk},
                                 {"x": self.wv, "w": x, "bias_rows": self.bv})
        inner = self.inner
        kw = dict(H=self.heads, D=64, Sq=S, Skv=S, Skv_alloc=S_alloc, q_row_stride=2 * inner, k_row_stride=2 * inner,
                  q_batch_stride=S_alloc * 2 * inner, k_batch_stride=S_alloc * 2 * inner, vt_ld=B * S_alloc,
                  vt_batch_stride=S_alloc, scale=self.scale, causal=causal)
        if S_alloc == S:
            o = ops.attention(qk, qk[:, inner:], vt, B=B, bias=bias, **kw)
        else:
            # the kernel packs its output at S rows per batch; the token matrices here keep S_alloc: one launch per batch
            o = torch.zeros((B * S_alloc, inner), device=x.device, dtype=bf16)
            for b in range(B):
                r0 = b * S_alloc
                ops.attention(qk[r0:], qk[r0:, inner:], vt[:, r0:], B=1, out=o[r0:r0 + S],
                              bias=None if bias is None else (bias[b:b + 1] if bias.shape[0] > 1 else bias), **kw)
        return ops.linear(o, self.wo, self.bo, residual=residual)


# ----------------------------------------------------------------------------------------------------------------------
# CLIP
# ----------------------------------------------------------------------------------------------------------------------
_CLIP_DEFAULTS = dict(vocab_size=49408, hidden_size=512, intermediate_size=2048, projection_dim=512, num_hidden_layers=12,
                      num_attention_heads=8, max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                      eos_token_id=49407, pad_token_id=1, bos_token_id=49406)
_CLIP_ACT = {"quick_gelu": L.ACT_QUICK_GELU, "gelu": L.ACT_GELU_ERF, "gelu_new": L.ACT_GELU_TANH,
             "gelu_pytorch_tanh": L.ACT_GELU_TANH}


class CLIPTextModel:
    """``transformers.CLIPTextModel`` (modeling_clip.py CLIPTextTransformer.forward): embeddings -> N x {LN1, causal
    self-attention, LN2, MLP} -> final_layer_norm; pooled = the EOS row."""

    with_projection = False

    def __init__(self, config):
        self.config = _cfg(config, _CLIP_DEFAULTS)
        c = self.config
        if c.hidden_act not in _CLIP_ACT:
            raise ValueError(f"CLIP hidden_act {c.hidden_act!r} is not supported")
        if c.hidden_size % c.num_attention_heads or c.hidden_size // c.num_attention_heads != 64 or c.hidden_size % 64:
            raise ValueError("diffusers_amd CLIP text encoder: head size must be 64 and hidden_size a multiple of 64")
        self.dtype, self.device, self._built = bf16, None, False

    def load_state_dict(self, state_dict, device="cuda", strict: bool = True):
        c = self.config
        w = Weights(state_dict, device)
        self.device = torch.device(device)
        # checkpoints and transformers <= 4.x: "text_model.*"; transformers 5.x CLIPTextModel.state_dict(): no prefix
        p = "text_model." if w.has("text_model.embeddings.token_embedding.weight") else ""
        self.tok = w.get(p + "embeddings.token_embedding.weight")
        self.pos = w.get(p + "embeddings.position_embedding.weight")
        self.layers = []
        for i in range(c.num_hidden_layers):
            q = f"{p}encoder.layers.{i}."
            a = _SelfAttention(*(w.get(q + f"self_attn.{n}_proj.weight") for n in ("q", "k", "v", "out")),
                               *(w.get(q + f"self_attn.{n}_proj.bias") for n in ("q", "k", "v", "out")),
                               heads=c.num_attention_heads)
            self.layers.append({"ln1": (w.get(q + "layer_norm1.weight"), w.get(q + "layer_norm1.bias")), "attn": a,
                                "ln2": (w.get(q + "layer_norm2.weight"), w.get(q + "layer_norm2.bias")),
                                "fc1": (w.get(q + "mlp.fc1.weight"), w.get(q + "mlp.fc1.bias")),
                                "fc2": (w.get(q + "mlp.fc2.weight"), w.get(q + "mlp.fc2.bias"))})
        self.final_ln = (w.get(p + "final_layer_norm.weight"), w.get(p + "final_layer_norm.bias"))
        self.text_projection = w.get("text_projection.weight") if self.with_projection else None
        w.used.add(p + "embeddings.position_ids")
        if strict and [k for k in w.unused() if not k.endswith("position_ids")]:
            raise RuntimeError(f"unexpected keys in state_dict: {w.unused()[:8]} ...")
        self._built = True
        return self

    def to(self, *args, **kwargs):
        return check_to(self, args, kwargs)

    def eval(self):
        return self

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    def _unpad(self, h, B, S, S_alloc):
        return h.view(B, S_alloc, -1)[:, :S]

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask=None, position_ids=None, output_hidden_states: Optional[bool] = None,
                return_dict: bool = True, **kwargs):
        if not self._built:
            raise RuntimeError(f"{type(self).__name__}: call load_state_dict() first")
        if attention_mask is not None and not bool(torch.as_tensor(attention_mask).bool().all()):
            raise ValueError("diffusers_amd CLIP text encoder: key-padding masks are not supported (the diffusers pipelines "
                             "pass none: pipeline_stable_diffusion_xl.py:406)")
        if position_ids is not None:
            raise ValueError("position_ids are not supported")
        c = self.config
        ids = input_ids.view(-1, input_ids.shape[-1])
        B, S = ids.shape
        if S > c.max_position_embeddings:
            raise ValueError(f"Sequence length must be less than max_position_embeddings (got `sequence length`: {S} and "
                             f"max_position_embeddings: {c.max_position_embeddings}")
        S_alloc = _pad_rows(S)
        dev = self.device
        emb = torch.nn.functional.embedding(ids.to(dev), self.tok) + self.pos[:S][None]          # gather + add (bf16)
        x = torch.zeros((B, S_alloc, c.hidden_size), device=dev, dtype=bf16)
        x[:, :S] = emb
        x = x.view(B * S_alloc, c.hidden_size)
        act = _CLIP_ACT[c.hidden_act]
        hidden = [self._unpad(x, B, S, S_alloc)] if output_hidden_states else None
        for ly in self.layers:
            h = ops.layer_norm(x, *ly["ln1"], c.layer_norm_eps)
            x = ly["attn"](h, B, S, S_alloc, residual=x, causal=True)
            h = ops.layer_norm(x, *ly["ln2"], c.layer_norm_eps)
            h = ops.linear(h, *ly["fc1"], act=act)
            x = ops.linear(h, *ly["fc2"], residual=x)
            if output_hidden_states:
                hidden.append(self._unpad(x, B, S, S_alloc))
        last = self._unpad(ops.layer_norm(x, *self.final_ln, c.layer_norm_eps), B, S, S_alloc)
        idc = ids.to("cpu", torch.int)
        eos = idc.argmax(dim=-1) if c.eos_token_id == 2 else (idc == c.eos_token_id).int().argmax(dim=-1)
        pooled = last[torch.arange(B, device=dev), eos.to(dev)]
        hs = tuple(hidden) if output_hidden_states else None
        if self.with_projection:
            te = ops.linear_small_m(pooled.contiguous(), self.text_projection) if B <= 8 else \
                ops.linear(pooled.contiguous(), self.text_projection)
            out = ModelOutput(text_embeds=te, last_hidden_state=last, hidden_states=hs)
        else:
            out = ModelOutput(last_hidden_state=last, pooler_output=pooled, hidden_states=hs)
        return out if return_dict else out.to_tuple()


class CLIPTextModelWithProjection(CLIPTextModel):
    """``transformers.CLIPTextModelWithProjection``: ``[0]`` = ``text_embeds`` = text_projection(pooled) (what SDXL reads
    as the pooled prompt embedding of its second encoder)."""

    with_projection = True


# ----------------------------------------------------------------------------------------------------------------------
# T5 / UMT5 encoders
# ----------------------------------------------------------------------------------------------------------------------
_T5_DEFAULTS = dict(vocab_size=32128, d_model=512, d_kv=64, d_ff=2048, num_layers=6, num_heads=8,
                    relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6,
                    feed_forward_proj="relu", dense_act_fn=None, is_gated_act=None)


def relative_position_bucket(relative_position: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """modeling_t5.py T5Attention._relative_position_bucket, bidirectional (encoder)."""
    num_buckets //= 2
    buckets = (relative_position > 0).to(torch.long) * num_buckets
    rp = torch.abs(relative_position)
    max_exact = num_buckets // 2
    is_small = rp < max_exact
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(is_small, rp, large)


class T5EncoderModel:
    """``transformers.T5EncoderModel`` (v1.1 "gated-gelu" checkpoints: FLUX's text_encoder_2): embed -> N x {T5LayerNorm,
    self-attention with the relative position bias of block 0 (no 1/sqrt(d)), T5LayerNorm, gelu_new(wi_0 x) * (wi_1 x) -> wo}
    -> final T5LayerNorm.  ``per_layer_bias``: UMT5 gives every block its own bias table."""

    per_layer_bias = False

    def __init__(self, config):
        self.config = _cfg(config, _T5_DEFAULTS)
        c = self.config
        gated = c.feed_forward_proj.startswith("gated-") if c.is_gated_act is None else c.is_gated_act
        act = c.dense_act_fn or c.feed_forward_proj.split("-")[-1]
        if not gated or act not in ("gelu", "gelu_new"):
            raise ValueError("diffusers_amd T5 encoder: only the gated-GELU feed-forward (T5 v1.1 / UMT5) is supported")
        if c.d_kv != 64 or c.d_model % 64 or (2 * c.d_ff) % 128:
            raise ValueError("diffusers_amd T5 encoder: d_kv must be 64, d_model a multiple of 64, d_ff a multiple of 64")
        self.dtype, self.device, self._built = bf16, None, False
        self._bias_cache: Dict[Any, torch.Tensor] = {}

    def load_state_dict(self, state_dict, device="cuda", strict: bool = True):
        c = self.config
        w = Weights(state_dict, device)
        self.device = torch.device(device)
        self.embed = w.get("shared.weight") if w.has("shared.weight") else w.get("encoder.embed_tokens.weight")
        w.used.update({"shared.weight", "encoder.embed_tokens.weight"})
        self.blocks = []
        for i in range(c.num_layers):
            p = f"encoder.block.{i}.layer."
            att = _SelfAttention(*(w.get(f"{p}0.SelfAttention.{n}.weight") for n in ("q", "k", "v", "o")),
                                 heads=c.num_heads, scale=1.0)
            wi0, wi1 = w.get(p + "1.DenseReluDense.wi_0.weight"), w.get(p + "1.DenseReluDense.wi_1.weight")
            wff, _ = ops.pack_geglu(torch.cat([wi1, wi0], 0), None)       # value rows = wi_1, gate rows = wi_0
            blk = {"ln0": w.get(p + "0.layer_norm.weight"), "attn": att, "ln1": w.get(p + "1.layer_norm.weight"),
                   "wff": wff, "wo": w.get(p + "1.DenseReluDense.wo.weight"), "rel": None}
            if w.has(p + "0.SelfAttention.relative_attention_bias.weight"):
                blk["rel"] = w.get_f32(p + "0.SelfAttention.relative_attention_bias.weight")
            self.blocks.append(blk)
        if self.blocks[0]["rel"] is None or (self.per_layer_bias and any(b["rel"] is None for b in self.blocks)):
            raise KeyError("relative_attention_bias table missing")
        self.final_ln = w.get("encoder.final_layer_norm.weight")
        if strict and w.unused():
            raise RuntimeError(f"unexpected keys in state_dict: {w.unused()[:8]} ...")
        self._built = True
        return self

    def to(self, *args, **kwargs):
        return check_to(self, args, kwargs)

    def eval(self):
        return self

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    def _position_bias(self, table: torch.Tensor, S: int, S_alloc: int, mask: Optional[torch.Tensor], key) -> torch.Tensor:
        """[1 or B][H][S][ceil64(S_alloc)] bf16: bias[h][i][j] = table[bucket(j - i)][h] (+ -inf where key j is padding)."""
        c = self.config
        ck = (key, S, None if mask is None else mask.cpu().numpy().tobytes())
        if ck in self._bias_cache:
            return self._bias_cache[ck]
        ctx = torch.arange(S, dtype=torch.long)[:, None]
        mem = torch.arange(S, dtype=torch.long)[None, :]
        bk = relative_position_bucket(mem - ctx, c.relative_attention_num_buckets, c.relative_attention_max_distance)
        vals = table.detach().to("cpu", torch.float32).to(bf16)[bk]                 # (S, S, H), model dtype as the reference
        vals = vals.permute(2, 0, 1).unsqueeze(0).float()                           # (1, H, S, S)
        if mask is not None:
            m = mask.to("cpu").bool()                                               # (B, S): True = keep
            vals = vals.expand(m.shape[0], -1, -1, -1).clone()
            vals.masked_fill_(~m[:, None, None, :], -1e30)
        ld = ((S + 63) // 64) * 64
        out = torch.zeros(vals.shape[:3] + (ld,), dtype=torch.float32)
        out[..., :S] = vals
        out = out.to(self.device)
        if len(self._bias_cache) > 8:
            self._bias_cache.clear()
        self._bias_cache[ck] = out
        return out

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                output_hidden_states: Optional[bool] = None, return_dict: bool = True, **kwargs):
        if not self._built:
            raise RuntimeError(f"{type(self).__name__}: call load_state_dict() first")
        c = self.config
        B, S = input_ids.shape
        S_alloc = _pad_rows(S)
        dev = self.device
        mask = None
        if attention_mask is not None and not bool(attention_mask.bool().all()):
            mask = attention_mask
        x = torch.zeros((B, S_alloc, c.d_model), device=dev, dtype=bf16)
        x[:, :S] = torch.nn.functional.embedding(input_ids.to(dev), self.embed)
        x = x.view(B * S_alloc, c.d_model)
        eps = c.layer_norm_epsilon
        bias = None
        # transformers' T5Stack: hidden_states = (embeddings, every block's output ..., the FINAL-NORMED last state)
        hidden = [x.view(B, S_alloc, -1)[:, :S].clone()] if output_hidden_states else None
        for i, blk in enumerate(self.blocks):
            if blk["rel"] is not None and (bias is None or self.per_layer_bias):
                bias = self._position_bias(blk["rel"], S, S_alloc, mask, i)
            h = ops.rms_norm(x, blk["ln0"], eps)
            x = blk["attn"](h, B, S, S_alloc, residual=x, bias=bias)
            h = ops.rms_norm(x, blk["ln1"], eps)
            h = ops.linear(h, blk["wff"], act=L.ACT_GEGLU_TANH)
            x = ops.linear(h, blk["wo"], residual=x)
            if output_hidden_states and i + 1 < len(self.blocks):
                hidden.append(x.view(B, S_alloc, -1)[:, :S].clone())
        last = ops.rms_norm(x, self.final_ln, eps).view(B, S_alloc, -1)[:, :S]
        if output_hidden_states:
            hidden.append(last)
        out = ModelOutput(last_hidden_state=last, hidden_states=tuple(hidden) if output_hidden_states else None)
        return out if return_dict else out.to_tuple()


class UMT5EncoderModel(T5EncoderModel):
    """``transformers.UMT5EncoderModel`` (Wan's text encoder): T5 v1.1 with a relative position bias table in EVERY block
    (modeling_umt5.py UMT5Block)."""

    per_layer_bias = True
