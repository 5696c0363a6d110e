"""UNet2DModel (unconditional pixel-space U-Net of the DDPM pipelines, BASELINE config 1) on the gfx950 kernels.

Mirrors the reference class (models/unets/unet_2d.py:95-353) for the google/ddpm-* family: DownBlock2D / AttnDownBlock2D
/ UNetMidBlock2D / AttnUpBlock2D / UpBlock2D (unet_2d_blocks.py:1018-1146, :1346-1369, :736-748, :2185-2313, :2524-2572),
positional time embedding, ``attention_head_dim=None`` (one head as wide as the block) or a multiple-of-32 head size.
Same constructor kwargs, ``state_dict`` keys / shapes and ``forward(sample, timestep)`` signature; NCHW in / out.
The reference runs this model in fp32; the engine computes in bf16 with fp32 accumulation like every other row (the
tolerance is stated in the tests).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import torch

from . import ops
from .config_utils import check_to
from .loading import PretrainedMixin
from .autoencoder_kl import VaeAttention
from .layers import Downsample2D, GroupNorm, ResnetBlock2D, TimeProjections, TimestepEmbedding, Upsample2D, Weights
from .unet_2d_condition import FrozenConfig

bf16 = torch.bfloat16


@dataclass
class UNet2DOutput:
    sample: torch.Tensor


_DEFAULTS = dict(
    sample_size=None, in_channels=3, out_channels=3, center_input_sample=False, time_embedding_type="positional",
    time_embedding_dim=None, freq_shift=0, flip_sin_to_cos=True,
    down_block_types=("DownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D"),
    mid_block_type="UNetMidBlock2D", up_block_types=("AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "UpBlock2D"),
    block_out_channels=(224, 448, 672, 896), layers_per_block=2, mid_block_scale_factor=1, downsample_padding=1,
    downsample_type="conv", upsample_type="conv", dropout=0.0, act_fn="silu", attention_head_dim=8, norm_num_groups=32,
    attn_norm_num_groups=None, norm_eps=1e-5, resnet_time_scale_shift="default", add_attention=True,
    class_embed_type=None, num_class_embeds=None, num_train_timesteps=None,
)


class UNet2DModel(PretrainedMixin):
    """Drop-in for the reference ``UNet2DModel`` (inference, bf16, HIP device only)."""

    def __init__(self, **kwargs):
        unknown = set(kwargs) - set(_DEFAULTS)
        if unknown:
            raise TypeError(f"UNet2DModel: unexpected config keys {sorted(unknown)}")
        cfg = dict(_DEFAULTS)
        cfg.update(kwargs)
        self.config = FrozenConfig(cfg)
        c = self.config
        if len(c.down_block_types) != len(c.up_block_types):
            raise ValueError(f"Must provide the same number of `down_block_types` as `up_block_types`. "
                             f"`down_block_types`: {c.down_block_types}. `up_block_types`: {c.up_block_types}.")
        if len(c.block_out_channels) != len(c.down_block_types):
            raise ValueError(f"Must provide the same number of `block_out_channels` as `down_block_types`. "
                             f"`block_out_channels`: {c.block_out_channels}. `down_block_types`: {c.down_block_types}.")
        for t in c.down_block_types:
            if t not in ("DownBlock2D", "AttnDownBlock2D"):
                raise ValueError(f"{t} does not exist.")
        for t in c.up_block_types:
            if t not in ("UpBlock2D", "AttnUpBlock2D"):
                raise ValueError(f"{t} does not exist.")
        if (c.time_embedding_type != "positional" or c.act_fn != "silu" or c.resnet_time_scale_shift != "default"
                or c.downsample_type != "conv" or c.upsample_type != "conv" or c.class_embed_type is not None
                or c.num_class_embeds is not None or c.center_input_sample or c.mid_block_type != "UNetMidBlock2D"
                or c.attn_norm_num_groups is not None):
            raise ValueError("diffusers_amd UNet2DModel supports the positional-embedding / conv-resampling DDPM family")
        if any(ch % 64 for ch in c.block_out_channels):
            raise ValueError("block_out_channels must be multiples of 64 (one K slice of the implicit GEMM)")
        self.dtype = bf16
        self.device = None
        self._built = False

    def _heads(self, channels):
        hd = self.config.attention_head_dim
        hd = channels if hd is None else hd
        return channels // hd

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], device="cuda", strict: bool = True):
        c = self.config
        w = Weights(state_dict, device)
        self.device = torch.device(device)
        boc = tuple(c.block_out_channels)
        n = len(boc)
        groups, eps = c.norm_num_groups, c.norm_eps
        self.conv_in_w = ops.pack_conv_weight(w.get("conv_in.weight"))
        self.conv_in_b = w.get("conv_in.bias")
        self.time_embedding = TimestepEmbedding(w, "time_embedding")
        self.down = []
        for i, bt in enumerate(c.down_block_types):
            pre = f"down_blocks.{i}"
            st = {"resnets": [], "attns": [], "down": None}
            for j in range(c.layers_per_block):
                st["resnets"].append(ResnetBlock2D(w, f"{pre}.resnets.{j}", groups, eps))
                if bt == "AttnDownBlock2D":
                    st["attns"].append(VaeAttention(w, f"{pre}.attentions.{j}", groups, eps, heads=self._heads(boc[i])))
            if i != n - 1:
                st["down"] = Downsample2D(w, f"{pre}.downsamplers.0", padding=c.downsample_padding)
            self.down.append(st)
        self.mid_res0 = ResnetBlock2D(w, "mid_block.resnets.0", groups, eps, c.mid_block_scale_factor)
        self.mid_attn = VaeAttention(w, "mid_block.attentions.0", groups, eps, heads=self._heads(boc[-1])) \
            if c.add_attention else None
        self.mid_res1 = ResnetBlock2D(w, "mid_block.resnets.1", groups, eps, c.mid_block_scale_factor)
        rboc = tuple(reversed(boc))
        self.up = []
        for i, bt in enumerate(c.up_block_types):
            pre = f"up_blocks.{i}"
            st = {"resnets": [], "attns": [], "up": None}
            for j in range(c.layers_per_block + 1):
                st["resnets"].append(ResnetBlock2D(w, f"{pre}.resnets.{j}", groups, eps))
                if bt == "AttnUpBlock2D":
                    st["attns"].append(VaeAttention(w, f"{pre}.attentions.{j}", groups, eps, heads=self._heads(rboc[i])))
            if i != n - 1:
                st["up"] = Upsample2D(w, f"{pre}.upsamplers.0")
            self.up.append(st)
        # every block's time_emb_proj in one launch per forward (layers.TimeProjections), blocks in forward order
        self.time_proj = TimeProjections([r for st in self.down for r in st["resnets"]] + [self.mid_res0, self.mid_res1] +
                                         [r for st in self.up for r in st["resnets"]])
        self.conv_norm_out = GroupNorm(w, "conv_norm_out", groups, eps)
        self.conv_out_w = ops.pack_conv_weight(w.get("conv_out.weight"))
        self.conv_out_b = w.get("conv_out.bias")
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

    def forward(self, sample: torch.Tensor, timestep, class_labels=None, return_dict: bool = True, sampler_table=None,
                step_idx=None):
        """unet_2d.py:249-353.  ``sampler_table`` / ``step_idx``: read the timestep from the device sampler table."""
        if not self._built:
            raise RuntimeError("UNet2DModel: call load_state_dict() first")
        if class_labels is not None:
            raise ValueError("class_labels should not be provided: the model has no class embedding")
        ops.require_hip(sample, "sample")
        c = self.config
        B = sample.shape[0]
        if sampler_table is not None:
            t_emb = ops.timestep_embedding(None, c.block_out_channels[0], batch=B, flip_sin_to_cos=c.flip_sin_to_cos,
                                           shift=float(c.freq_shift), table=sampler_table, step_idx=step_idx)
        else:
            if not torch.is_tensor(timestep):
                timestep = torch.tensor([float(timestep)], dtype=torch.float32)
            t = timestep.to(device=self.device, dtype=torch.float32).reshape(-1)
            if t.numel() == 1:
                t = t.expand(B)
            t_emb = ops.timestep_embedding(t.contiguous(), c.block_out_channels[0], batch=B,
                                           flip_sin_to_cos=c.flip_sin_to_cos, shift=float(c.freq_shift))
        emb = self.time_proj(self.time_embedding(t_emb))     # the resnets below take their columns of this
        x = ops.conv_thin_in(sample.contiguous(), self.conv_in_w, self.conv_in_b, ksize=3, in_nchw=True)
        skips = [x]
        for st in self.down:
            for j, rn in enumerate(st["resnets"]):
                x = rn(x, emb)
                if st["attns"]:
                    x = st["attns"][j](x)
                skips.append(x)
            if st["down"] is not None:
                x = st["down"](x)
                skips.append(x)
        x = self.mid_res0(x, emb)
        if self.mid_attn is not None:
            x = self.mid_attn(x)
        x = self.mid_res1(x, emb)
        for st in self.up:
            for j, rn in enumerate(st["resnets"]):
                x = rn(x, emb, skip=skips.pop())
                if st["attns"]:
                    x = st["attns"][j](x)
            if st["up"] is not None:
                x = st["up"](x)
        x = self.conv_norm_out(x, silu=True)
        out = ops.conv_thin_out(x, self.conv_out_w, self.conv_out_b)
        if not return_dict:
            return (out,)
        return UNet2DOutput(sample=out)
