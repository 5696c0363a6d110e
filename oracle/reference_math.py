"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package ``diffusers_amd``.

CPU restatement (plain PyTorch, fp32 by default) of the reference's algorithm for the denoising hot path, written as
functions over a reference-format ``state_dict``.  Every function cites the reference lines it follows (paths under
/root/reference/src/diffusers/).  Parity status: PINNED -- ``oracle/make_golden.py`` runs the real reference classes in
the build container on the same seeded weights/inputs and ``tests/test_oracle_vs_golden.py`` checks this restatement
against those committed outputs (tests/golden/*.npz), plus the reference's own known-answer vectors for the schedulers
(tests/schedulers/test_scheduler_{ddim,euler,ddpm}.py full-loop sums).

The arithmetic (conv, matmul, softmax ...) is delegated to torch CPU ops exactly as the reference delegates it to torch
(the reference has no kernels of its own; its third-party arithmetic dependency is torch>=2.6, here
torch 2.10.0).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F


def _tup(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


# --------------------------------------------------------------------------------------------------------------------
# op-level references
# --------------------------------------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int, flip_sin_to_cos: bool, shift: float, scale: float = 1.0,
                       max_period: float = 10000.0) -> torch.Tensor:
    """models/embeddings.py:27-78 (get_timestep_embedding), fp32."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=t.device)
    exponent = exponent / (half - shift)
    emb = torch.exp(exponent)
    emb = t[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def geglu(x, w, b):
    """models/activations.py:93-124: proj -> chunk(2) -> hidden * gelu(gate)."""
    h = F.linear(x, w, b)
    h, gate = h.chunk(2, dim=-1)
    return h * F.gelu(gate)


def attention(q, k, v, heads: int, scale: Optional[float] = None):
    """models/attention_processor.py:2753-2773: (B,S,H*D) -> heads -> softmax(q k^T / sqrt(D)) v -> (B,S,H*D)."""
    B, Sq, C = q.shape
    D = C // heads
    qh = q.view(B, Sq, heads, D).transpose(1, 2)
    kh = k.view(B, -1, heads, D).transpose(1, 2)
    vh = v.view(B, -1, heads, D).transpose(1, 2)
    if q.is_cuda:
        # on a GPU the reference's call IS F.scaled_dot_product_attention (attention_processor.py:2767) with whatever
        # backend PyTorch-ROCm picks; the explicit form below is its CPU "math" path and would materialise S x S scores
        o = F.scaled_dot_product_attention(qh, kh, vh, dropout_p=0.0, is_causal=False, scale=scale)
        return o.transpose(1, 2).reshape(B, Sq, C)
    s = (qh @ kh.transpose(-1, -2)) * (D ** -0.5 if scale is None else scale)
    p = torch.softmax(s, dim=-1)
    o = p @ vh
    return o.transpose(1, 2).reshape(B, Sq, C)


def resnet_block(sd: Dict[str, torch.Tensor], p: str, x, temb, groups: int, eps: float, out_scale: float = 1.0):
    """models/resnet.py:319-377 (default time scale shift, SiLU)."""
    h = F.group_norm(x, groups, sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1)
    if temb is not None and f"{p}.time_emb_proj.weight" in sd:
        t = F.linear(F.silu(temb), sd[f"{p}.time_emb_proj.weight"], sd[f"{p}.time_emb_proj.bias"])
        h = h + t[:, :, None, None]
    h = F.group_norm(h, groups, sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], padding=1)
    if f"{p}.conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.conv_shortcut.weight"], sd[f"{p}.conv_shortcut.bias"])
    return (x + h) / out_scale


def basic_transformer_block(sd, p: str, x, ctx, heads: int):
    """models/attention.py:960-1080 (layer_norm variant) + AttnProcessor2_0 + FeedForward(geglu)."""
    def attn(a, h, c):
        q = F.linear(h, sd[f"{p}.{a}.to_q.weight"])
        k = F.linear(c, sd[f"{p}.{a}.to_k.weight"])
        v = F.linear(c, sd[f"{p}.{a}.to_v.weight"])
        o = attention(q, k, v, heads)
        return F.linear(o, sd[f"{p}.{a}.to_out.0.weight"], sd[f"{p}.{a}.to_out.0.bias"])

    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], 1e-5)
    x = x + attn("attn1", h, h)
    h = F.layer_norm(x, (C,), sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], 1e-5)
    x = x + attn("attn2", h, ctx)
    h = F.layer_norm(x, (C,), sd[f"{p}.norm3.weight"], sd[f"{p}.norm3.bias"], 1e-5)
    ff = geglu(h, sd[f"{p}.ff.net.0.proj.weight"], sd[f"{p}.ff.net.0.proj.bias"])
    ff = F.linear(ff, sd[f"{p}.ff.net.2.weight"], sd[f"{p}.ff.net.2.bias"])
    return x + ff


def transformer_2d(sd, p: str, x, ctx, heads: int, layers: int, groups: int, linear_proj: bool):
    """models/transformers/transformer_2d.py:324-512 (continuous)."""
    B, C, H, W = x.shape
    res = x
    h = F.group_norm(x, groups, sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], 1e-6)
    if linear_proj:
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = F.linear(h, sd[f"{p}.proj_in.weight"], sd[f"{p}.proj_in.bias"])
    else:
        h = F.conv2d(h, sd[f"{p}.proj_in.weight"], sd[f"{p}.proj_in.bias"])
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    for k in range(layers):
        h = basic_transformer_block(sd, f"{p}.transformer_blocks.{k}", h, ctx, heads)
    if linear_proj:
        h = F.linear(h, sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    else:
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = F.conv2d(h, sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])
    return h + res


# --------------------------------------------------------------------------------------------------------------------
# UNet2DConditionModel.forward  (models/unets/unet_2d_condition.py:979-1235)
# --------------------------------------------------------------------------------------------------------------------
def unet_forward(sd: Dict[str, torch.Tensor], cfg: dict, sample, timestep, encoder_hidden_states,
                 added_cond_kwargs: Optional[dict] = None):
    boc = tuple(cfg["block_out_channels"])
    n = len(boc)
    groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    heads = _tup(cfg.get("num_attention_heads") or cfg["attention_head_dim"], n)
    lpb = _tup(cfg["layers_per_block"], n)
    tlpb = _tup(cfg["transformer_layers_per_block"], n)
    lin = cfg["use_linear_projection"]
    B = sample.shape[0]
    dt = sample.dtype

    t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).to(sample.device)   # any device: the GPU tests and
    if t.numel() == 1:                                                                # bench legs run this graph on cuda
        t = t.expand(B)
    t_emb = timestep_embedding(t, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"]).to(dt)          # :852-872
    emb = F.linear(t_emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    emb = F.linear(F.silu(emb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    if cfg["addition_embed_type"] == "text_time":                                                   # :906-922
        te = added_cond_kwargs["text_embeds"]
        ids = added_cond_kwargs["time_ids"]
        tid = timestep_embedding(ids.flatten().float(), cfg["addition_time_embed_dim"], cfg["flip_sin_to_cos"],
                                 cfg["freq_shift"])
        tid = tid.reshape(te.shape[0], -1)
        add = torch.cat([te.float(), tid], dim=-1).to(dt)
        aug = F.linear(add, sd["add_embedding.linear_1.weight"], sd["add_embedding.linear_1.bias"])
        aug = F.linear(F.silu(aug), sd["add_embedding.linear_2.weight"], sd["add_embedding.linear_2.bias"])
        emb = emb + aug

    x = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)                       # :1108
    skips = [x]
    for i, bt in enumerate(cfg["down_block_types"]):                                                # :1136
        for j in range(lpb[i]):
            x = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", x, emb, groups, eps)
            if bt == "CrossAttnDownBlock2D":
                x = transformer_2d(sd, f"down_blocks.{i}.attentions.{j}", x, encoder_hidden_states, heads[i], tlpb[i],
                                   groups, lin)
            skips.append(x)
        if i != n - 1:
            p = f"down_blocks.{i}.downsamplers.0.conv"
            x = F.conv2d(x, sd[f"{p}.weight"], sd[f"{p}.bias"], stride=2, padding=1)
            skips.append(x)
    x = resnet_block(sd, "mid_block.resnets.0", x, emb, groups, eps, cfg["mid_block_scale_factor"])  # :1171
    x = transformer_2d(sd, "mid_block.attentions.0", x, encoder_hidden_states, heads[-1], tlpb[-1], groups, lin)
    x = resnet_block(sd, "mid_block.resnets.1", x, emb, groups, eps, cfg["mid_block_scale_factor"])
    rheads = tuple(reversed(heads))
    rlpb = tuple(reversed(lpb))
    rt = cfg.get("reverse_transformer_layers_per_block")
    rtlpb = tuple(reversed(tlpb)) if rt is None else _tup(rt, n)
    for i, bt in enumerate(cfg["up_block_types"]):                                                  # :1196
        for j in range(rlpb[i] + 1):
            x = torch.cat([x, skips.pop()], dim=1)                                                  # blocks.py:2444
            x = resnet_block(sd, f"up_blocks.{i}.resnets.{j}", x, emb, groups, eps)
            if bt == "CrossAttnUpBlock2D":
                x = transformer_2d(sd, f"up_blocks.{i}.attentions.{j}", x, encoder_hidden_states, rheads[i], rtlpb[i],
                                   groups, lin)
        if i != n - 1:
            p = f"up_blocks.{i}.upsamplers.0.conv"
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")                                  # upsampling.py:177
            x = F.conv2d(x, sd[f"{p}.weight"], sd[f"{p}.bias"], padding=1)
    x = F.group_norm(x, groups, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps)          # :1227-1230
    x = F.silu(x)
    return F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


# --------------------------------------------------------------------------------------------------------------------
# AutoencoderKL.decode  (autoencoder_kl.py:199-240, vae.py:279-311, unet_2d_blocks.py:736-748,:2637-2645)
# --------------------------------------------------------------------------------------------------------------------
def vae_attention(sd, p: str, x, groups: int, eps: float):
    """Legacy 4-D Attention (attention_processor.py:2705-2787): GN, q/k/v with bias, 1 head, + residual."""
    B, C, H, W = x.shape
    res = x
    h = F.group_norm(x, groups, sd[f"{p}.group_norm.weight"], sd[f"{p}.group_norm.bias"], eps)
    h = h.view(B, C, H * W).transpose(1, 2)
    q = F.linear(h, sd[f"{p}.to_q.weight"], sd[f"{p}.to_q.bias"])
    k = F.linear(h, sd[f"{p}.to_k.weight"], sd[f"{p}.to_k.bias"])
    v = F.linear(h, sd[f"{p}.to_v.weight"], sd[f"{p}.to_v.bias"])
    o = attention(q, k, v, heads=1)
    o = F.linear(o, sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"])
    o = o.transpose(1, 2).reshape(B, C, H, W)
    return o + res


def vae_decode(sd: Dict[str, torch.Tensor], cfg: dict, z):
    groups, eps = cfg["norm_num_groups"], 1e-6
    boc = tuple(cfg["block_out_channels"])
    if cfg["use_post_quant_conv"]:
        z = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = F.conv2d(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    x = resnet_block(sd, "decoder.mid_block.resnets.0", x, None, groups, eps)
    if cfg["mid_block_add_attention"]:
        x = vae_attention(sd, "decoder.mid_block.attentions.0", x, groups, eps)
    x = resnet_block(sd, "decoder.mid_block.resnets.1", x, None, groups, eps)
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"] + 1):
            x = resnet_block(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, None, groups, eps)
        if i != len(boc) - 1:
            p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"{p}.weight"], sd[f"{p}.bias"], padding=1)
    x = F.group_norm(x, groups, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], eps)
    x = F.silu(x)
    return F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


# --------------------------------------------------------------------------------------------------------------------
# Flux (models/transformers/transformer_flux.py)
# --------------------------------------------------------------------------------------------------------------------
def rope_tables(ids: torch.Tensor, axes_dim, theta: float = 10000.0):
    """FluxPosEmbed.forward + get_1d_rotary_pos_embed(use_real=True, repeat_interleave_real=True, float64 freqs)
    (transformer_flux.py:500-526, embeddings.py:1120-1184): returns (cos, sin) fp32 [S][sum(axes_dim)]."""
    pos = ids.float()
    cos_out, sin_out = [], []
    for i, dim in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64, device=ids.device) / dim))
        f = torch.outer(pos[:, i], freqs)          # float32 pos x float64 freqs -> float64
        cos_out.append(f.cos().repeat_interleave(2, dim=1).float())
        sin_out.append(f.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos_out, dim=-1), torch.cat(sin_out, dim=-1)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """apply_rotary_emb(use_real=True, use_real_unbind_dim=-1, sequence_dim=1), embeddings.py:1187-1232.
    x: (B, S, H, D); cos/sin: (S, D)."""
    c = cos[None, :, None, :]
    s = sin[None, :, None, :]
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * c + rot.float() * s).to(x.dtype)


def _sdpa(q, k, v):
    """F.scaled_dot_product_attention on (B, H, S, D) tensors.  Same arithmetic; on a GPU in fp32 at video sequence lengths
    (Wan: 32 760 tokens, where PyTorch-ROCm's fp32 path would materialise 12 x 32 760^2 scores) the queries are processed in
    blocks -- each block still sees every key, so every output row is the exact softmax(q k^T / sqrt D) v."""
    Sq, Skv = q.shape[-2], k.shape[-2]
    if not (q.is_cuda and q.dtype == torch.float32 and Sq * Skv > (1 << 26)):
        return F.scaled_dot_product_attention(q, k, v)
    out = torch.empty_like(q)
    step = max(256, (1 << 26) // Skv)
    for s0 in range(0, Sq, step):
        out[..., s0:s0 + step, :] = F.scaled_dot_product_attention(q[..., s0:s0 + step, :], k, v)
    return out


def _rms(x, w, eps):
    """torch.nn.RMSNorm(head_dim, eps) (transformer_flux.py:316-317)."""
    return F.rms_norm(x, (x.shape[-1],), w, eps)


def _flux_attention(q, k, v, heads):
    """dispatch_attention_fn native backend on (B, S, H, D) tensors (attention_dispatch.py:3709)."""
    o = _sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    return o.transpose(1, 2).flatten(2, 3)


def _ada_chunks(sd, p, temb, n):
    return F.linear(F.silu(temb), sd[f"{p}.weight"], sd[f"{p}.bias"]).chunk(n, dim=1)


def flux_double_block(sd, p, x, ctx, temb, cos, sin, heads, D):
    """FluxTransformerBlock.forward (transformer_flux.py:443-497) + FluxAttnProcessor (:76-142)."""
    C = x.shape[-1]
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = _ada_chunks(sd, f"{p}.norm1.linear", temb, 6)
    csh_a, csc_a, cg_a, csh_m, csc_m, cg_m = _ada_chunks(sd, f"{p}.norm1_context.linear", temb, 6)
    nx = F.layer_norm(x, (C,), None, None, 1e-6) * (1 + sc_a[:, None]) + sh_a[:, None]
    nc = F.layer_norm(ctx, (C,), None, None, 1e-6) * (1 + csc_a[:, None]) + csh_a[:, None]

    def proj(h, nm):
        return F.linear(h, sd[f"{p}.attn.{nm}.weight"], sd[f"{p}.attn.{nm}.bias"]).unflatten(-1, (heads, D))
    q, k, v = proj(nx, "to_q"), proj(nx, "to_k"), proj(nx, "to_v")
    q = _rms(q, sd[f"{p}.attn.norm_q.weight"], 1e-6)
    k = _rms(k, sd[f"{p}.attn.norm_k.weight"], 1e-6)
    eq, ek, ev = proj(nc, "add_q_proj"), proj(nc, "add_k_proj"), proj(nc, "add_v_proj")
    eq = _rms(eq, sd[f"{p}.attn.norm_added_q.weight"], 1e-6)
    ek = _rms(ek, sd[f"{p}.attn.norm_added_k.weight"], 1e-6)
    q = apply_rope(torch.cat([eq, q], dim=1), cos, sin)
    k = apply_rope(torch.cat([ek, k], dim=1), cos, sin)
    v = torch.cat([ev, v], dim=1)
    o = _flux_attention(q, k, v, heads)
    St = ctx.shape[1]
    o_ctx, o_x = o[:, :St], o[:, St:]
    o_x = F.linear(o_x, sd[f"{p}.attn.to_out.0.weight"], sd[f"{p}.attn.to_out.0.bias"])
    o_ctx = F.linear(o_ctx, sd[f"{p}.attn.to_add_out.weight"], sd[f"{p}.attn.to_add_out.bias"])

    x = x + g_a[:, None] * o_x
    n2 = F.layer_norm(x, (C,), None, None, 1e-6) * (1 + sc_m[:, None]) + sh_m[:, None]
    ff = F.gelu(F.linear(n2, sd[f"{p}.ff.net.0.proj.weight"], sd[f"{p}.ff.net.0.proj.bias"]), approximate="tanh")
    ff = F.linear(ff, sd[f"{p}.ff.net.2.weight"], sd[f"{p}.ff.net.2.bias"])
    x = x + g_m[:, None] * ff

    ctx = ctx + cg_a[:, None] * o_ctx
    n2c = F.layer_norm(ctx, (C,), None, None, 1e-6) * (1 + csc_m[:, None]) + csh_m[:, None]
    ffc = F.gelu(F.linear(n2c, sd[f"{p}.ff_context.net.0.proj.weight"], sd[f"{p}.ff_context.net.0.proj.bias"]),
                 approximate="tanh")
    ffc = F.linear(ffc, sd[f"{p}.ff_context.net.2.weight"], sd[f"{p}.ff_context.net.2.bias"])
    ctx = ctx + cg_m[:, None] * ffc
    return ctx, x


def flux_single_block(sd, p, x, ctx, temb, cos, sin, heads, D):
    """FluxSingleTransformerBlock.forward (transformer_flux.py:383-412)."""
    St = ctx.shape[1]
    h = torch.cat([ctx, x], dim=1)
    C = h.shape[-1]
    res = h
    sh, sc, g = _ada_chunks(sd, f"{p}.norm.linear", temb, 3)
    n = F.layer_norm(h, (C,), None, None, 1e-6) * (1 + sc[:, None]) + sh[:, None]
    mlp = F.gelu(F.linear(n, sd[f"{p}.proj_mlp.weight"], sd[f"{p}.proj_mlp.bias"]), approximate="tanh")

    def proj(nm):
        return F.linear(n, sd[f"{p}.attn.{nm}.weight"], sd[f"{p}.attn.{nm}.bias"]).unflatten(-1, (heads, D))
    q = apply_rope(_rms(proj("to_q"), sd[f"{p}.attn.norm_q.weight"], 1e-6), cos, sin)
    k = apply_rope(_rms(proj("to_k"), sd[f"{p}.attn.norm_k.weight"], 1e-6), cos, sin)
    o = _flux_attention(q, k, proj("to_v"), heads)
    out = F.linear(torch.cat([o, mlp], dim=2), sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])
    h = res + g[:, None] * out
    return h[:, :St], h[:, St:]


def flux_forward(sd: Dict[str, torch.Tensor], cfg: dict, hidden_states, encoder_hidden_states, pooled_projections,
                 timestep, img_ids, txt_ids):
    """FluxTransformer2DModel.forward (transformer_flux.py:671-821), guidance_embeds=False.
    hidden_states (B, S_img, in_channels) packed latents; timestep (B,) in [0, 1]."""
    heads, D = cfg["num_attention_heads"], cfg["attention_head_dim"]
    dt = hidden_states.dtype
    x = F.linear(hidden_states, sd["x_embedder.weight"], sd["x_embedder.bias"])
    t = timestep.to(dt) * 1000
    # CombinedTimestepTextProjEmbeddings (embeddings.py:1585-1601)
    tp = timestep_embedding(t, 256, flip_sin_to_cos=True, shift=0.0).to(dt)
    te = F.linear(tp, sd["time_text_embed.timestep_embedder.linear_1.weight"],
                  sd["time_text_embed.timestep_embedder.linear_1.bias"])
    te = F.linear(F.silu(te), sd["time_text_embed.timestep_embedder.linear_2.weight"],
                  sd["time_text_embed.timestep_embedder.linear_2.bias"])
    pe = F.linear(pooled_projections, sd["time_text_embed.text_embedder.linear_1.weight"],
                  sd["time_text_embed.text_embedder.linear_1.bias"])
    pe = F.linear(F.silu(pe), sd["time_text_embed.text_embedder.linear_2.weight"],
                  sd["time_text_embed.text_embedder.linear_2.bias"])
    temb = te + pe
    ctx = F.linear(encoder_hidden_states, sd["context_embedder.weight"], sd["context_embedder.bias"])
    cos, sin = rope_tables(torch.cat((txt_ids, img_ids), dim=0), cfg["axes_dims_rope"])
    for i in range(cfg["num_layers"]):
        ctx, x = flux_double_block(sd, f"transformer_blocks.{i}", x, ctx, temb, cos, sin, heads, D)
    for i in range(cfg["num_single_layers"]):
        ctx, x = flux_single_block(sd, f"single_transformer_blocks.{i}", x, ctx, temb, cos, sin, heads, D)
    # AdaLayerNormContinuous (normalization.py:307-351): scale first, then shift
    emb = F.linear(F.silu(temb), sd["norm_out.linear.weight"], sd["norm_out.linear.bias"])
    scale, shift = emb.chunk(2, dim=1)
    C = x.shape[-1]
    x = F.layer_norm(x, (C,), None, None, 1e-6) * (1 + scale)[:, None, :] + shift[:, None, :]
    return F.linear(x, sd["proj_out.weight"], sd["proj_out.bias"])


# --------------------------------------------------------------------------------------------------------------------
# Wan 2.1 T2V (models/transformers/transformer_wan.py)
# --------------------------------------------------------------------------------------------------------------------
def wan_rope_tables(head_dim: int, frames: int, height: int, width: int, max_seq_len: int, theta: float = 10000.0):
    """WanRotaryPosEmbed (transformer_wan.py:354-416) for a (frames, height, width) post-patch grid:
    (cos, sin) fp32 [frames*height*width][head_dim], each frequency repeated for the pair it rotates."""
    h_dim = w_dim = 2 * (head_dim // 6)
    t_dim = head_dim - h_dim - w_dim
    tabs = []
    for dim in (t_dim, h_dim, w_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64) / dim))
        f = torch.outer(torch.arange(max_seq_len), freqs)
        tabs.append((f.cos().repeat_interleave(2, dim=1).float(), f.sin().repeat_interleave(2, dim=1).float()))

    def grid(i):
        ct, st = tabs[0][i][:frames].view(frames, 1, 1, -1).expand(frames, height, width, -1), None
        ch = tabs[1][i][:height].view(1, height, 1, -1).expand(frames, height, width, -1)
        cw = tabs[2][i][:width].view(1, 1, width, -1).expand(frames, height, width, -1)
        return torch.cat([ct, ch, cw], dim=-1).reshape(frames * height * width, head_dim)
    return grid(0), grid(1)


def wan_apply_rope(x, cos, sin):
    """WanAttnProcessor.apply_rotary_emb (transformer_wan.py:103-116).  x: (B, S, H, D); cos/sin: (S, D)."""
    x1, x2 = x.unflatten(-1, (-1, 2)).unbind(-1)
    c = cos[None, :, None, 0::2]
    s = sin[None, :, None, 1::2]
    out = torch.empty_like(x)
    out[..., 0::2] = x1 * c - x2 * s
    out[..., 1::2] = x1 * s + x2 * c
    return out.type_as(x)


def _fp32_layer_norm(x, w, b, eps):
    """FP32LayerNorm (normalization.py:429-444)."""
    return F.layer_norm(x.float(), (x.shape[-1],), None if w is None else w.float(), None if b is None else b.float(),
                        eps).to(x.dtype)


def wan_attention(sd, p, x, ctx, heads, cos=None, sin=None, eps=1e-6):
    """WanAttention + WanAttnProcessor (transformer_wan.py:68-162): q/k RMSNorm over ALL heads, RoPE, SDPA, to_out."""
    src = x if ctx is None else ctx
    q = F.linear(x, sd[f"{p}.to_q.weight"], sd[f"{p}.to_q.bias"])
    k = F.linear(src, sd[f"{p}.to_k.weight"], sd[f"{p}.to_k.bias"])
    v = F.linear(src, sd[f"{p}.to_v.weight"], sd[f"{p}.to_v.bias"])
    q = F.rms_norm(q, (q.shape[-1],), sd[f"{p}.norm_q.weight"], eps)
    k = F.rms_norm(k, (k.shape[-1],), sd[f"{p}.norm_k.weight"], eps)
    q, k, v = (t.unflatten(2, (heads, -1)) for t in (q, k, v))
    if cos is not None:
        q, k = wan_apply_rope(q, cos, sin), wan_apply_rope(k, cos, sin)
    o = _sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2)
    o = o.flatten(2, 3).type_as(q)
    return F.linear(o, sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"])


def wan_block(sd, p, x, ctx, tproj, cos, sin, heads, eps, cross_attn_norm):
    """WanTransformerBlock.forward (transformer_wan.py:462-504), temb (B, 6, dim)."""
    sh, sc, g, csh, csc, cg = (sd[f"{p}.scale_shift_table"].float() + tproj.float()).chunk(6, dim=1)
    n = (_fp32_layer_norm(x.float(), None, None, eps) * (1 + sc) + sh).type_as(x)
    a = wan_attention(sd, f"{p}.attn1", n, None, heads, cos, sin, eps)
    x = (x.float() + a * g).type_as(x)
    if cross_attn_norm:
        n = _fp32_layer_norm(x.float(), sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], eps).type_as(x)
    else:
        n = x
    x = x + wan_attention(sd, f"{p}.attn2", n, ctx, heads, None, None, eps)
    n = (_fp32_layer_norm(x.float(), None, None, eps) * (1 + csc) + csh).type_as(x)
    ff = F.gelu(F.linear(n, sd[f"{p}.ffn.net.0.proj.weight"], sd[f"{p}.ffn.net.0.proj.bias"]), approximate="tanh")
    ff = F.linear(ff, sd[f"{p}.ffn.net.2.weight"], sd[f"{p}.ffn.net.2.bias"])
    return (x.float() + ff.float() * cg).type_as(x)


def wan_forward(sd: Dict[str, torch.Tensor], cfg: dict, hidden_states, timestep, encoder_hidden_states):
    """WanTransformer3DModel.forward (transformer_wan.py:629-735), T2V.  hidden_states (B, C, F, H, W)."""
    heads, D, eps = cfg["num_attention_heads"], cfg["attention_head_dim"], cfg["eps"]
    B, C, Fr, H, W = hidden_states.shape
    pt, ph, pw = cfg["patch_size"]
    f, h, w = Fr // pt, H // ph, W // pw
    cos, sin = (t.to(hidden_states.device) for t in wan_rope_tables(D, f, h, w, cfg["rope_max_seq_len"]))
    x = F.conv3d(hidden_states, sd["patch_embedding.weight"], sd["patch_embedding.bias"], stride=(pt, ph, pw))
    x = x.flatten(2).transpose(1, 2).contiguous()
    # WanTimeTextImageEmbedding (transformer_wan.py:308-351)
    tp = timestep_embedding(timestep, cfg["freq_dim"], flip_sin_to_cos=True, shift=0.0)
    te = F.linear(tp.to(x.dtype), sd["condition_embedder.time_embedder.linear_1.weight"],
                  sd["condition_embedder.time_embedder.linear_1.bias"])
    temb = F.linear(F.silu(te), sd["condition_embedder.time_embedder.linear_2.weight"],
                    sd["condition_embedder.time_embedder.linear_2.bias"]).type_as(encoder_hidden_states)
    tproj = F.linear(F.silu(temb), sd["condition_embedder.time_proj.weight"], sd["condition_embedder.time_proj.bias"])
    tproj = tproj.unflatten(1, (6, -1))
    ctx = F.linear(encoder_hidden_states, sd["condition_embedder.text_embedder.linear_1.weight"],
                   sd["condition_embedder.text_embedder.linear_1.bias"])
    ctx = F.linear(F.gelu(ctx, approximate="tanh"), sd["condition_embedder.text_embedder.linear_2.weight"],
                   sd["condition_embedder.text_embedder.linear_2.bias"])
    for i in range(cfg["num_layers"]):
        x = wan_block(sd, f"blocks.{i}", x, ctx, tproj, cos, sin, heads, eps, cfg.get("cross_attn_norm", True))
    shift, scale = (sd["scale_shift_table"] + temb.unsqueeze(1)).chunk(2, dim=1)
    x = (_fp32_layer_norm(x.float(), None, None, eps) * (1 + scale) + shift).type_as(x)
    x = F.linear(x, sd["proj_out.weight"], sd["proj_out.bias"])
    x = x.reshape(B, f, h, w, pt, ph, pw, -1).permute(0, 7, 1, 4, 2, 5, 3, 6)
    return x.flatten(6, 7).flatten(4, 5).flatten(2, 3)


# --------------------------------------------------------------------------------------------------------------------
# UNet2DModel (models/unets/unet_2d.py) -- the DDPM pixel-space U-Net
# --------------------------------------------------------------------------------------------------------------------
def attn_block_2d(sd, p: str, x, groups: int, eps: float, heads: int):
    """Legacy Attention on a 4-D input (attention_processor.py:2705-2787: GroupNorm, biased q/k/v, residual), as used by
    AttnDownBlock2D / AttnUpBlock2D / UNetMidBlock2D (unet_2d_blocks.py:1018-1146, :2185-2313, :736-748)."""
    B, C, H, W = x.shape
    h = F.group_norm(x, groups, sd[f"{p}.group_norm.weight"], sd[f"{p}.group_norm.bias"], eps)
    h = h.view(B, C, H * W).transpose(1, 2)
    q = F.linear(h, sd[f"{p}.to_q.weight"], sd[f"{p}.to_q.bias"])
    k = F.linear(h, sd[f"{p}.to_k.weight"], sd[f"{p}.to_k.bias"])
    v = F.linear(h, sd[f"{p}.to_v.weight"], sd[f"{p}.to_v.bias"])
    o = attention(q, k, v, heads)
    o = F.linear(o, sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"])
    return o.transpose(1, 2).reshape(B, C, H, W) + x


def unet2d_forward(sd: Dict[str, torch.Tensor], cfg: dict, sample, timestep):
    """UNet2DModel.forward (unet_2d.py:249-353), positional time embedding, conv down / up sampling."""
    boc = tuple(cfg["block_out_channels"])
    n = len(boc)
    groups, eps, L_ = cfg["norm_num_groups"], cfg["norm_eps"], cfg["layers_per_block"]
    hd = cfg["attention_head_dim"]
    B = sample.shape[0]
    t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(B)
    t_emb = timestep_embedding(t, boc[0], cfg["flip_sin_to_cos"], float(cfg["freq_shift"])).to(sample.dtype)
    emb = F.linear(t_emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    emb = F.linear(F.silu(emb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    x = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    skips = [x]
    for i, bt in enumerate(cfg["down_block_types"]):
        heads = boc[i] // (hd if hd is not None else boc[i])
        for j in range(L_):
            x = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", x, emb, groups, eps)
            if bt == "AttnDownBlock2D":
                x = attn_block_2d(sd, f"down_blocks.{i}.attentions.{j}", x, groups, eps, heads)
            skips.append(x)
        if i != n - 1:
            p = f"down_blocks.{i}.downsamplers.0.conv"
            if cfg["downsample_padding"] == 0:   # downsampling.py:139-141
                x = F.conv2d(F.pad(x, (0, 1, 0, 1)), sd[f"{p}.weight"], sd[f"{p}.bias"], stride=2)
            else:
                x = F.conv2d(x, sd[f"{p}.weight"], sd[f"{p}.bias"], stride=2, padding=cfg["downsample_padding"])
            skips.append(x)
    mheads = boc[-1] // (hd if hd is not None else boc[-1])
    x = resnet_block(sd, "mid_block.resnets.0", x, emb, groups, eps, cfg["mid_block_scale_factor"])
    if cfg.get("add_attention", True):
        x = attn_block_2d(sd, "mid_block.attentions.0", x, groups, eps, mheads)
    x = resnet_block(sd, "mid_block.resnets.1", x, emb, groups, eps, cfg["mid_block_scale_factor"])
    rboc = tuple(reversed(boc))
    for i, bt in enumerate(cfg["up_block_types"]):
        heads = rboc[i] // (hd if hd is not None else rboc[i])
        for j in range(L_ + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(sd, f"up_blocks.{i}.resnets.{j}", x, emb, groups, eps)
            if bt == "AttnUpBlock2D":
                x = attn_block_2d(sd, f"up_blocks.{i}.attentions.{j}", x, groups, eps, heads)
        if i != n - 1:
            p = f"up_blocks.{i}.upsamplers.0.conv"
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"{p}.weight"], sd[f"{p}.bias"], padding=1)
    x = F.group_norm(x, groups, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps)
    return F.conv2d(F.silu(x), sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


# --------------------------------------------------------------------------------------------------------------------
# AutoencoderKLWan.decode (models/autoencoders/autoencoder_kl_wan.py) -- SURVEY.md 8f rank 2
# --------------------------------------------------------------------------------------------------------------------
def wan_causal_conv3d(x, w, b):
    """WanCausalConv3d (:131-173): zero padding (kt - 1) frames in FRONT only, symmetric in space."""
    kt, kh, kw = w.shape[2:]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, kt - 1, 0))
    return F.conv3d(x, w, b)


def wan_rms_norm(x, gamma):
    """WanRMS_norm (:176-206), channel-first, no bias: F.normalize over channels * sqrt(C) * gamma."""
    return F.normalize(x, dim=1) * (x.shape[1] ** 0.5) * gamma


def wan_vae_resblock(sd, p, x):
    """WanResidualBlock (:315-386) over the WHOLE frame sequence.  The reference decodes one latent frame per call and
    hands each causal conv its previous two input frames through ``feat_cache``; with zero frames in front of frame 0
    that is exactly the causal convolution of the full sequence."""
    h = wan_causal_conv3d(x, sd[f"{p}.conv_shortcut.weight"], sd[f"{p}.conv_shortcut.bias"]) \
        if f"{p}.conv_shortcut.weight" in sd else x
    x = F.silu(wan_rms_norm(x, sd[f"{p}.norm1.gamma"]))
    x = wan_causal_conv3d(x, sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"])
    x = F.silu(wan_rms_norm(x, sd[f"{p}.norm2.gamma"]))
    x = wan_causal_conv3d(x, sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"])
    return x + h


def wan_vae_attention(sd, p, x):
    """WanAttentionBlock (:389-431): per-frame single-head self-attention over the h*w positions."""
    B, C, T, H, W = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    y = wan_rms_norm(y, sd[f"{p}.norm.gamma"])
    qkv = F.conv2d(y, sd[f"{p}.to_qkv.weight"], sd[f"{p}.to_qkv.bias"]).reshape(B * T, 3 * C, H * W).transpose(1, 2)
    q, k, v = qkv.chunk(3, dim=-1)
    o = torch.softmax(q @ k.transpose(1, 2) * (C ** -0.5), dim=-1) @ v
    o = o.transpose(1, 2).reshape(B * T, C, H, W)
    o = F.conv2d(o, sd[f"{p}.proj.weight"], sd[f"{p}.proj.bias"])
    return o.view(B, T, C, H, W).permute(0, 2, 1, 3, 4) + x


def wan_vae_upsample(sd, p, x, temporal: bool):
    """WanResample 'upsample2d' / 'upsample3d' (:224-312) over the whole sequence.  Chunked decoding marks the FIRST
    latent frame "Rep": it skips time_conv and stays one frame; every later frame t goes through the causal (3,1,1)
    time_conv whose history starts at frame 1 (the cache after the "Rep" chunk is zeros, :286-287), and its 2C output
    channels become the two frames 2t-1, 2t (:297-299)."""
    B, C, T, H, W = x.shape
    if temporal and T > 1:
        y = wan_causal_conv3d(x[:, :, 1:], sd[f"{p}.time_conv.weight"], sd[f"{p}.time_conv.bias"])   # [B][2C][T-1]
        y = y.reshape(B, 2, C, T - 1, H, W)
        y = torch.stack((y[:, 0], y[:, 1]), 3).reshape(B, C, 2 * (T - 1), H, W)
        x = torch.cat([x[:, :, :1], y], 2)
    T2 = x.shape[2]
    y = x.permute(0, 2, 1, 3, 4).reshape(B * T2, C, H, W)
    y = F.interpolate(y, scale_factor=(2.0, 2.0), mode="nearest-exact")
    y = F.conv2d(y, sd[f"{p}.resample.1.weight"], sd[f"{p}.resample.1.bias"], padding=1)
    return y.view(B, T2, y.shape[1], 2 * H, 2 * W).permute(0, 2, 1, 3, 4)


def wan_vae_decode(sd: Dict[str, torch.Tensor], cfg: dict, z):
    """AutoencoderKLWan._decode (:1187-1217) + WanDecoder3d.forward (:879-914), Wan 2.1 layout (is_residual False,
    patch_size None).  z: [B][z_dim][T][H][W] -> [B][3][1 + 4 (T-1)][8H][8W], clamped to [-1, 1]."""
    if cfg.get("is_residual") or cfg.get("patch_size") is not None:
        raise NotImplementedError("Wan 2.2 residual / patchified VAE")
    mult = list(cfg["dim_mult"])
    t_up = list(cfg["temperal_downsample"])[::-1]
    x = wan_causal_conv3d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = wan_causal_conv3d(x, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"])
    x = wan_vae_resblock(sd, "decoder.mid_block.resnets.0", x)
    x = wan_vae_attention(sd, "decoder.mid_block.attentions.0", x)
    x = wan_vae_resblock(sd, "decoder.mid_block.resnets.1", x)
    for i in range(len(mult)):
        for j in range(cfg["num_res_blocks"] + 1):
            x = wan_vae_resblock(sd, f"decoder.up_blocks.{i}.resnets.{j}", x)
        if i != len(mult) - 1:
            x = wan_vae_upsample(sd, f"decoder.up_blocks.{i}.upsamplers.0", x, t_up[i])
    x = F.silu(wan_rms_norm(x, sd["decoder.norm_out.gamma"]))
    x = wan_causal_conv3d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"])
    return torch.clamp(x, min=-1.0, max=1.0)
