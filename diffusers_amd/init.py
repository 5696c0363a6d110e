"""Parameter inventories (reference ``state_dict`` key -> shape) and deterministic seeded initialisation.

No checkpoints exist offline, so parity tests and the benchmark use seeded random weights.  The inventories are derived
from the config exactly as the reference constructors derive them (unet_2d_condition.py:179-560, unet_2d_blocks.py,
vae.py:180-278); ``oracle/make_golden.py`` checks them against the reference's own ``state_dict()`` key/shape set.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch


def _tup(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def _resnet(sh, p, cin, cout, temb):
    sh[f"{p}.norm1.weight"] = (cin,)
    sh[f"{p}.norm1.bias"] = (cin,)
    sh[f"{p}.conv1.weight"] = (cout, cin, 3, 3)
    sh[f"{p}.conv1.bias"] = (cout,)
    if temb:
        sh[f"{p}.time_emb_proj.weight"] = (cout, temb)
        sh[f"{p}.time_emb_proj.bias"] = (cout,)
    sh[f"{p}.norm2.weight"] = (cout,)
    sh[f"{p}.norm2.bias"] = (cout,)
    sh[f"{p}.conv2.weight"] = (cout, cout, 3, 3)
    sh[f"{p}.conv2.bias"] = (cout,)
    if cin != cout:
        sh[f"{p}.conv_shortcut.weight"] = (cout, cin, 1, 1)
        sh[f"{p}.conv_shortcut.bias"] = (cout,)


def _transformer(sh, p, c, cross, layers, linear_proj):
    sh[f"{p}.norm.weight"] = (c,)
    sh[f"{p}.norm.bias"] = (c,)
    proj = (c, c) if linear_proj else (c, c, 1, 1)
    sh[f"{p}.proj_in.weight"] = proj
    sh[f"{p}.proj_in.bias"] = (c,)
    for k in range(layers):
        b = f"{p}.transformer_blocks.{k}"
        for nm in ("norm1", "norm2", "norm3"):
            sh[f"{b}.{nm}.weight"] = (c,)
            sh[f"{b}.{nm}.bias"] = (c,)
        for a, kd in (("attn1", c), ("attn2", cross)):
            sh[f"{b}.{a}.to_q.weight"] = (c, c)
            sh[f"{b}.{a}.to_k.weight"] = (c, kd)
            sh[f"{b}.{a}.to_v.weight"] = (c, kd)
            sh[f"{b}.{a}.to_out.0.weight"] = (c, c)
            sh[f"{b}.{a}.to_out.0.bias"] = (c,)
        sh[f"{b}.ff.net.0.proj.weight"] = (8 * c, c)
        sh[f"{b}.ff.net.0.proj.bias"] = (8 * c,)
        sh[f"{b}.ff.net.2.weight"] = (c, 4 * c)
        sh[f"{b}.ff.net.2.bias"] = (c,)
    sh[f"{p}.proj_out.weight"] = proj
    sh[f"{p}.proj_out.bias"] = (c,)


def unet_param_shapes(cfg) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict inventory of UNet2DConditionModel for the supported configs (SD1.5 / SDXL families)."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = tuple(cfg["block_out_channels"])
    n = len(boc)
    temb = boc[0] * 4
    lpb = _tup(cfg["layers_per_block"], n)
    tlpb = _tup(cfg["transformer_layers_per_block"], n)
    cross = _tup(cfg["cross_attention_dim"], n)
    lin = cfg["use_linear_projection"]
    sh["conv_in.weight"] = (boc[0], cfg["in_channels"], 3, 3)
    sh["conv_in.bias"] = (boc[0],)
    sh["time_embedding.linear_1.weight"] = (temb, boc[0])
    sh["time_embedding.linear_1.bias"] = (temb,)
    sh["time_embedding.linear_2.weight"] = (temb, temb)
    sh["time_embedding.linear_2.bias"] = (temb,)
    if cfg["addition_embed_type"] == "text_time":
        sh["add_embedding.linear_1.weight"] = (temb, cfg["projection_class_embeddings_input_dim"])
        sh["add_embedding.linear_1.bias"] = (temb,)
        sh["add_embedding.linear_2.weight"] = (temb, temb)
        sh["add_embedding.linear_2.bias"] = (temb,)
    out_c = boc[0]
    for i, bt in enumerate(cfg["down_block_types"]):
        in_c, out_c = out_c, boc[i]
        for j in range(lpb[i]):
            _resnet(sh, f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c, temb)
            if bt == "CrossAttnDownBlock2D":
                _transformer(sh, f"down_blocks.{i}.attentions.{j}", out_c, cross[i], tlpb[i], lin)
        if i != n - 1:
            sh[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            sh[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (out_c,)
    mid = boc[-1]
    _resnet(sh, "mid_block.resnets.0", mid, mid, temb)
    _transformer(sh, "mid_block.attentions.0", mid, cross[-1], tlpb[-1], lin)
    _resnet(sh, "mid_block.resnets.1", mid, mid, temb)
    rboc = tuple(reversed(boc))
    rlpb = tuple(reversed(lpb))
    rcross = tuple(reversed(cross))
    rt = cfg.get("reverse_transformer_layers_per_block")
    rtlpb = tuple(reversed(tlpb)) if rt is None else _tup(rt, n)
    out_c = rboc[0]
    for i, bt in enumerate(cfg["up_block_types"]):
        prev_out = out_c
        out_c = rboc[i]
        in_c = rboc[min(i + 1, n - 1)]
        nl = rlpb[i] + 1
        for j in range(nl):
            skip_c = in_c if j == nl - 1 else out_c
            res_in = prev_out if j == 0 else out_c
            _resnet(sh, f"up_blocks.{i}.resnets.{j}", res_in + skip_c, out_c, temb)
            if bt == "CrossAttnUpBlock2D":
                _transformer(sh, f"up_blocks.{i}.attentions.{j}", out_c, rcross[i], rtlpb[i], lin)
        if i != n - 1:
            sh[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            sh[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (out_c,)
    sh["conv_norm_out.weight"] = (boc[0],)
    sh["conv_norm_out.bias"] = (boc[0],)
    sh["conv_out.weight"] = (cfg["out_channels"], boc[0], 3, 3)
    sh["conv_out.bias"] = (cfg["out_channels"],)
    return sh


def vae_decoder_param_shapes(cfg) -> "OrderedDict[str, Tuple[int, ...]]":
    """Decoder half (+ post_quant_conv) of AutoencoderKL's state_dict."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = tuple(cfg["block_out_channels"])
    lat = cfg["latent_channels"]
    if cfg["use_post_quant_conv"]:
        sh["post_quant_conv.weight"] = (lat, lat, 1, 1)
        sh["post_quant_conv.bias"] = (lat,)
    top = boc[-1]
    sh["decoder.conv_in.weight"] = (top, lat, 3, 3)
    sh["decoder.conv_in.bias"] = (top,)
    _resnet(sh, "decoder.mid_block.resnets.0", top, top, 0)
    if cfg["mid_block_add_attention"]:
        a = "decoder.mid_block.attentions.0"
        sh[f"{a}.group_norm.weight"] = (top,)
        sh[f"{a}.group_norm.bias"] = (top,)
        for nm in ("to_q", "to_k", "to_v", "to_out.0"):
            sh[f"{a}.{nm}.weight"] = (top, top)
            sh[f"{a}.{nm}.bias"] = (top,)
    _resnet(sh, "decoder.mid_block.resnets.1", top, top, 0)
    rboc = tuple(reversed(boc))
    out_c = rboc[0]
    for i in range(len(boc)):
        prev, out_c = out_c, rboc[i]
        for j in range(cfg["layers_per_block"] + 1):
            _resnet(sh, f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else out_c, out_c, 0)
        if i != len(boc) - 1:
            sh[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            sh[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (out_c,)
    sh["decoder.conv_norm_out.weight"] = (boc[0],)
    sh["decoder.conv_norm_out.bias"] = (boc[0],)
    sh["decoder.conv_out.weight"] = (cfg["out_channels"], boc[0], 3, 3)
    sh["decoder.conv_out.bias"] = (cfg["out_channels"],)
    return sh


def _lin(sh, p, n_out, n_in, bias=True):
    sh[f"{p}.weight"] = (n_out, n_in)
    if bias:
        sh[f"{p}.bias"] = (n_out,)


def flux_param_shapes(cfg) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict inventory of FluxTransformer2DModel (transformer_flux.py:613-669) with guidance_embeds=False."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    D = cfg["attention_head_dim"]
    inner = cfg["num_attention_heads"] * D
    _lin(sh, "time_text_embed.timestep_embedder.linear_1", inner, 256)
    _lin(sh, "time_text_embed.timestep_embedder.linear_2", inner, inner)
    if cfg.get("guidance_embeds"):
        _lin(sh, "time_text_embed.guidance_embedder.linear_1", inner, 256)
        _lin(sh, "time_text_embed.guidance_embedder.linear_2", inner, inner)
    _lin(sh, "time_text_embed.text_embedder.linear_1", inner, cfg["pooled_projection_dim"])
    _lin(sh, "time_text_embed.text_embedder.linear_2", inner, inner)
    _lin(sh, "context_embedder", inner, cfg["joint_attention_dim"])
    _lin(sh, "x_embedder", inner, cfg["in_channels"])
    for i in range(cfg["num_layers"]):
        b = f"transformer_blocks.{i}"
        _lin(sh, f"{b}.norm1.linear", 6 * inner, inner)
        _lin(sh, f"{b}.norm1_context.linear", 6 * inner, inner)
        sh[f"{b}.attn.norm_q.weight"] = (D,)
        sh[f"{b}.attn.norm_k.weight"] = (D,)
        for nm in ("to_q", "to_k", "to_v"):
            _lin(sh, f"{b}.attn.{nm}", inner, inner)
        _lin(sh, f"{b}.attn.to_out.0", inner, inner)
        sh[f"{b}.attn.norm_added_q.weight"] = (D,)
        sh[f"{b}.attn.norm_added_k.weight"] = (D,)
        for nm in ("add_q_proj", "add_k_proj", "add_v_proj"):
            _lin(sh, f"{b}.attn.{nm}", inner, inner)
        _lin(sh, f"{b}.attn.to_add_out", inner, inner)
        _lin(sh, f"{b}.ff.net.0.proj", 4 * inner, inner)
        _lin(sh, f"{b}.ff.net.2", inner, 4 * inner)
        _lin(sh, f"{b}.ff_context.net.0.proj", 4 * inner, inner)
        _lin(sh, f"{b}.ff_context.net.2", inner, 4 * inner)
    for i in range(cfg["num_single_layers"]):
        b = f"single_transformer_blocks.{i}"
        _lin(sh, f"{b}.norm.linear", 3 * inner, inner)
        _lin(sh, f"{b}.proj_mlp", 4 * inner, inner)
        _lin(sh, f"{b}.proj_out", inner, 5 * inner)
        sh[f"{b}.attn.norm_q.weight"] = (D,)
        sh[f"{b}.attn.norm_k.weight"] = (D,)
        for nm in ("to_q", "to_k", "to_v"):
            _lin(sh, f"{b}.attn.{nm}", inner, inner)
    _lin(sh, "norm_out.linear", 2 * inner, inner)
    out_c = cfg.get("out_channels") or cfg["in_channels"]
    _lin(sh, "proj_out", cfg.get("patch_size", 1) ** 2 * out_c, inner)
    return sh


def wan_param_shapes(cfg) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict inventory of WanTransformer3DModel (transformer_wan.py:571-627), T2V (no image branch)."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    inner = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    pt, ph, pw = cfg["patch_size"]
    sh["patch_embedding.weight"] = (inner, cfg["in_channels"], pt, ph, pw)
    sh["patch_embedding.bias"] = (inner,)
    _lin(sh, "condition_embedder.time_embedder.linear_1", inner, cfg["freq_dim"])
    _lin(sh, "condition_embedder.time_embedder.linear_2", inner, inner)
    _lin(sh, "condition_embedder.time_proj", 6 * inner, inner)
    _lin(sh, "condition_embedder.text_embedder.linear_1", inner, cfg["text_dim"])
    _lin(sh, "condition_embedder.text_embedder.linear_2", inner, inner)
    for i in range(cfg["num_layers"]):
        b = f"blocks.{i}"
        for a in ("attn1", "attn2"):
            for nm in ("to_q", "to_k", "to_v", "to_out.0"):
                _lin(sh, f"{b}.{a}.{nm}", inner, inner)
            sh[f"{b}.{a}.norm_q.weight"] = (inner,)
            sh[f"{b}.{a}.norm_k.weight"] = (inner,)
        if cfg.get("cross_attn_norm", True):
            sh[f"{b}.norm2.weight"] = (inner,)
            sh[f"{b}.norm2.bias"] = (inner,)
        _lin(sh, f"{b}.ffn.net.0.proj", cfg["ffn_dim"], inner)
        _lin(sh, f"{b}.ffn.net.2", inner, cfg["ffn_dim"])
        sh[f"{b}.scale_shift_table"] = (1, 6, inner)
    _lin(sh, "proj_out", cfg["out_channels"] * pt * ph * pw, inner)
    sh["scale_shift_table"] = (1, 2, inner)
    return sh


def wan_vae_decoder_param_shapes(cfg) -> "OrderedDict[str, Tuple[int, ...]]":
    """Decoder half (+ post_quant_conv) of AutoencoderKLWan's state_dict (autoencoder_kl_wan.py:803-877,:1056), the
    Wan 2.1 layout (is_residual=False)."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    dim = cfg.get("decoder_base_dim") or cfg["base_dim"]
    z, mult, nres = cfg["z_dim"], list(cfg["dim_mult"]), cfg["num_res_blocks"]
    t_up = list(cfg["temperal_downsample"])[::-1]
    dims = [dim * u for u in [mult[-1]] + mult[::-1]]

    def res(p, cin, cout):
        sh[f"{p}.norm1.gamma"] = (cin, 1, 1, 1)
        sh[f"{p}.conv1.weight"] = (cout, cin, 3, 3, 3)
        sh[f"{p}.conv1.bias"] = (cout,)
        sh[f"{p}.norm2.gamma"] = (cout, 1, 1, 1)
        sh[f"{p}.conv2.weight"] = (cout, cout, 3, 3, 3)
        sh[f"{p}.conv2.bias"] = (cout,)
        if cin != cout:
            sh[f"{p}.conv_shortcut.weight"] = (cout, cin, 1, 1, 1)
            sh[f"{p}.conv_shortcut.bias"] = (cout,)

    sh["post_quant_conv.weight"] = (z, z, 1, 1, 1)
    sh["post_quant_conv.bias"] = (z,)
    sh["decoder.conv_in.weight"] = (dims[0], z, 3, 3, 3)
    sh["decoder.conv_in.bias"] = (dims[0],)
    a = "decoder.mid_block.attentions.0"
    sh[f"{a}.norm.gamma"] = (dims[0], 1, 1)
    sh[f"{a}.to_qkv.weight"] = (3 * dims[0], dims[0], 1, 1)
    sh[f"{a}.to_qkv.bias"] = (3 * dims[0],)
    sh[f"{a}.proj.weight"] = (dims[0], dims[0], 1, 1)
    sh[f"{a}.proj.bias"] = (dims[0],)
    res("decoder.mid_block.resnets.0", dims[0], dims[0])
    res("decoder.mid_block.resnets.1", dims[0], dims[0])
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        if i > 0:
            cin //= 2
        for j in range(nres + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i != len(mult) - 1:
            u = f"decoder.up_blocks.{i}.upsamplers.0"
            sh[f"{u}.resample.1.weight"] = (cout // 2, cout, 3, 3)
            sh[f"{u}.resample.1.bias"] = (cout // 2,)
            if t_up[i]:
                sh[f"{u}.time_conv.weight"] = (2 * cout, cout, 3, 1, 1)
                sh[f"{u}.time_conv.bias"] = (2 * cout,)
    sh["decoder.norm_out.gamma"] = (dims[-1], 1, 1, 1)
    sh["decoder.conv_out.weight"] = (cfg["out_channels"], dims[-1], 3, 3, 3)
    sh["decoder.conv_out.bias"] = (cfg["out_channels"],)
    return sh


def _attn_block(sh, p, c):
    sh[f"{p}.group_norm.weight"] = (c,)
    sh[f"{p}.group_norm.bias"] = (c,)
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        sh[f"{p}.{nm}.weight"] = (c, c)
        sh[f"{p}.{nm}.bias"] = (c,)


def unet2d_param_shapes(cfg) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict inventory of UNet2DModel (unet_2d.py:95-247) for the DDPM family (positional embedding, conv resampling)."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = tuple(cfg["block_out_channels"])
    n = len(boc)
    temb = cfg.get("time_embedding_dim") or boc[0] * 4
    L_ = cfg["layers_per_block"]
    sh["conv_in.weight"] = (boc[0], cfg["in_channels"], 3, 3)
    sh["conv_in.bias"] = (boc[0],)
    _lin(sh, "time_embedding.linear_1", temb, boc[0])
    _lin(sh, "time_embedding.linear_2", temb, temb)
    out_c = boc[0]
    for i, bt in enumerate(cfg["down_block_types"]):
        in_c, out_c = out_c, boc[i]
        for j in range(L_):
            _resnet(sh, f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c, temb)
            if bt == "AttnDownBlock2D":
                _attn_block(sh, f"down_blocks.{i}.attentions.{j}", out_c)
        if i != n - 1:
            sh[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            sh[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (out_c,)
    mid = boc[-1]
    _resnet(sh, "mid_block.resnets.0", mid, mid, temb)
    if cfg.get("add_attention", True):
        _attn_block(sh, "mid_block.attentions.0", mid)
    _resnet(sh, "mid_block.resnets.1", mid, mid, temb)
    rboc = tuple(reversed(boc))
    out_c = rboc[0]
    for i, bt in enumerate(cfg["up_block_types"]):
        prev_out, out_c = out_c, rboc[i]
        in_c = rboc[min(i + 1, n - 1)]
        for j in range(L_ + 1):
            skip_c = in_c if j == L_ else out_c
            res_in = prev_out if j == 0 else out_c
            _resnet(sh, f"up_blocks.{i}.resnets.{j}", res_in + skip_c, out_c, temb)
            if bt == "AttnUpBlock2D":
                _attn_block(sh, f"up_blocks.{i}.attentions.{j}", out_c)
        if i != n - 1:
            sh[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            sh[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (out_c,)
    sh["conv_norm_out.weight"] = (boc[0],)
    sh["conv_norm_out.bias"] = (boc[0],)
    sh["conv_out.weight"] = (cfg["out_channels"], boc[0], 3, 3)
    sh["conv_out.bias"] = (cfg["out_channels"],)
    return sh


def _seed_for(name: str, seed: int) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:7], "little")


def random_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, device="cpu",
                      dtype=torch.bfloat16) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic per-tensor init (independent of enumeration order): weights ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)),
    norm scales ~ 1 + 0.1 N(0,1), biases ~ 0.1 N(0,1)... scaled by 1/sqrt(fan_in) of their layer.  Values are drawn in
    fp32 then rounded to ``dtype`` so every consumer (engine, oracle, reference) sees bit-identical weights."""
    dev = torch.device(device)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in shapes.items():
        g = torch.Generator(device=dev)
        g.manual_seed(_seed_for(name, seed))
        is_norm = (".norm" in name or "norm." in name or "group_norm" in name or name.startswith("conv_norm_out")
                   or "conv_norm_out" in name) and ".linear." not in name
        if is_norm and (name.endswith(".weight") or name.endswith(".gamma")):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=dev, dtype=torch.float32)
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g, device=dev, dtype=torch.float32)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            b = 1.0 / (fan_in ** 0.5)
            t = (torch.rand(shape, generator=g, device=dev, dtype=torch.float32) * 2 - 1) * b
        out[name] = t.to(dtype)
    return out


# ---- canonical configs of the BASELINE ----
SDXL_UNET = dict(sample_size=128, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280),
                 layers_per_block=2, cross_attention_dim=2048, attention_head_dim=(5, 10, 20),
                 down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                 up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                 transformer_layers_per_block=(1, 2, 10), use_linear_projection=True, addition_embed_type="text_time",
                 addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816)
SD15_UNET = dict(sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, cross_attention_dim=768, attention_head_dim=8,
                 down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                 up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3)
SD_VAE = dict(in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
              down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4,
              latent_channels=4, sample_size=512, scaling_factor=0.18215)
SDXL_VAE = dict(SD_VAE, sample_size=1024, scaling_factor=0.13025)
FLUX_SCHNELL = dict(patch_size=1, in_channels=64, out_channels=None, num_layers=19, num_single_layers=38,
                    attention_head_dim=128, num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768,
                    guidance_embeds=False, axes_dims_rope=(16, 56, 56))
FLUX_VAE = dict(SD_VAE, latent_channels=16, sample_size=1024, use_quant_conv=False, use_post_quant_conv=False,
                scaling_factor=0.3611, shift_factor=0.1159)
TINY_FLUX = dict(patch_size=1, in_channels=64, out_channels=None, num_layers=2, num_single_layers=2,
                 attention_head_dim=64, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=64,
                 guidance_embeds=False, axes_dims_rope=(8, 28, 28))
TINY_FLUX_VAE = dict(in_channels=3, out_channels=3, block_out_channels=(64, 128), layers_per_block=1,
                     down_block_types=("DownEncoderBlock2D",) * 2, up_block_types=("UpDecoderBlock2D",) * 2,
                     latent_channels=16, sample_size=32, use_quant_conv=False, use_post_quant_conv=False,
                     scaling_factor=0.3611, shift_factor=0.1159)
# tiny members of the same families (every channel count a multiple of 64, head dim 64) for parity tests
TINY_SDXL_UNET = dict(sample_size=16, in_channels=4, out_channels=4, block_out_channels=(64, 128),
                      layers_per_block=1, cross_attention_dim=64, attention_head_dim=(1, 2),
                      down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"),
                      up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 2),
                      use_linear_projection=True, addition_embed_type="text_time", addition_time_embed_dim=32,
                      projection_class_embeddings_input_dim=256)
TINY_SD15_UNET = dict(sample_size=16, in_channels=4, out_channels=4, block_out_channels=(64, 128),
                      layers_per_block=1, cross_attention_dim=64, attention_head_dim=(1, 2),
                      down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                      up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"))
DDPM_CAT = dict(sample_size=256, in_channels=3, out_channels=3, block_out_channels=(128, 128, 256, 256, 512, 512),
                layers_per_block=2, down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
                up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4, attention_head_dim=None,
                norm_eps=1e-6, downsample_padding=0, flip_sin_to_cos=False, freq_shift=1)
DDPM_SCHEDULER = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                      variance_type="fixed_small", clip_sample=True)
TINY_DDPM = dict(DDPM_CAT, sample_size=32, block_out_channels=(64, 64, 128), layers_per_block=1,
                 down_block_types=("DownBlock2D", "AttnDownBlock2D", "DownBlock2D"),
                 up_block_types=("UpBlock2D", "AttnUpBlock2D", "UpBlock2D"))
WAN_1_3B = dict(patch_size=(1, 2, 2), num_attention_heads=12, attention_head_dim=128, in_channels=16, out_channels=16,
                text_dim=4096, freq_dim=256, ffn_dim=8960, num_layers=30, cross_attn_norm=True,
                qk_norm="rms_norm_across_heads", eps=1e-6, image_dim=None, added_kv_proj_dim=None, rope_max_seq_len=1024,
                pos_embed_seq_len=None)
TINY_WAN = dict(WAN_1_3B, num_attention_heads=2, attention_head_dim=64, text_dim=64, ffn_dim=256, num_layers=2,
                rope_max_seq_len=32)
WAN_VAE_LATENTS_MEAN = (-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517,
                        -0.3632, -0.1922, -0.9497, 0.2503, -0.2921)
WAN_VAE_LATENTS_STD = (2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579,
                       1.6382, 1.1253, 2.8251, 1.9160)
# AutoencoderKLWan defaults = the Wan 2.1 VAE (autoencoder_kl_wan.py:976-1030)
WAN_VAE = dict(base_dim=96, decoder_base_dim=None, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2, attn_scales=(),
               temperal_downsample=(False, True, True), dropout=0.0, latents_mean=WAN_VAE_LATENTS_MEAN,
               latents_std=WAN_VAE_LATENTS_STD, is_residual=False, in_channels=3, out_channels=3, patch_size=None,
               scale_factor_temporal=4, scale_factor_spatial=8)
# base_dim 24 -> channel counts 96 / 48 / 24: none a multiple of 64, so every zero-padding path is exercised
TINY_WAN_VAE = dict(WAN_VAE, base_dim=24, num_res_blocks=1)
# SD1.5's head geometry (attention_head_dim=8 means 8 HEADS: head dims 40 / 80 / 160) on a small spatial size
SMALL_SD15_UNET = dict(sample_size=8, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280),
                       layers_per_block=1, cross_attention_dim=64, attention_head_dim=8,
                       down_block_types=("CrossAttnDownBlock2D",) * 3, up_block_types=("CrossAttnUpBlock2D",) * 3)
TINY_VAE = dict(in_channels=3, out_channels=3, block_out_channels=(64, 128), layers_per_block=1,
                down_block_types=("DownEncoderBlock2D",) * 2, up_block_types=("UpDecoderBlock2D",) * 2,
                latent_channels=4, sample_size=32, scaling_factor=0.13025)
