"""AutoencoderKL.decode on the gfx950 kernels.

Mirrors models/autoencoders/autoencoder_kl.py:199-240 (decode/_decode) and models/autoencoders/vae.py:279-311
(Decoder.forward) with UNetMidBlock2D (unet_2d_blocks.py:736-748) and UpDecoderBlock2D (:2637-2645).  Only the decode
path is on the BASELINE hot path; ``encode`` raises.  Input latents NCHW bf16, output image NCHW bf16, as the reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import ops
from .config_utils import check_to
from .loading import PretrainedMixin
from .layers import GroupNorm, ResnetBlock2D, Upsample2D, Weights
from .unet_2d_condition import FrozenConfig

bf16 = torch.bfloat16


@dataclass
class DecoderOutput:
    sample: torch.Tensor


_DEFAULTS = dict(
    in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",), up_block_types=("UpDecoderBlock2D",),
    block_out_channels=(64,), layers_per_block=1, act_fn="silu", latent_channels=4, norm_num_groups=32, sample_size=32,
    scaling_factor=0.18215, shift_factor=None, latents_mean=None, latents_std=None, force_upcast=True,
    use_quant_conv=True, use_post_quant_conv=True, mid_block_add_attention=True,
)


class VaeAttention:
    """Legacy single-head Attention block of the VAE mid block (attention_processor.py:2696-2787 with a 4-D input:
    GroupNorm, to_q/k/v WITH bias, residual connection).  head_dim = channels (512 for SD/SDXL): handled as
    scores = Q K^T (fp32) -> row softmax -> P V with the MFMA GEMM; a flash-kernel head size uses the flash kernel.
    The V bias is folded into the output bias (softmax rows sum to 1):  W_o (P (V + 1 b_v^T)) + b_o = W_o P V + (W_o b_v + b_o)."""

    def __init__(self, w: Weights, prefix: str, groups: int, eps: float, heads: int = 1):
        self.group_norm = GroupNorm(w, prefix + ".group_norm", groups, eps)
        wq, wk, wv = (w.get(f"{prefix}.to_{n}.weight") for n in "qkv")
        bq, bk, bv = (w.opt(f"{prefix}.to_{n}.bias") for n in "qkv")
        self.inner = wq.shape[0]
        self.heads = heads
        self.head_dim = self.inner // heads
        self.wqk = torch.cat([wq, wk], 0).contiguous()
        self.bqk = torch.cat([bq, bk], 0).contiguous() if bq is not None else None
        self.wv = wv
        wo = w.get(prefix + ".to_out.0.weight")
        bo = w.opt(prefix + ".to_out.0.bias")
        self.wo = wo
        if bv is not None:
            eff = wo.float() @ bv.float()
            if bo is not None:
                eff = eff + bo.float()
            self.bo = eff.to(bf16).contiguous()
        else:
            self.bo = bo
        self.scale = self.head_dim ** -0.5
        self.force_gemm_path = False

    def __call__(self, x):
        B, H, W_, C = x.shape
        S = H * W_
        res = x.view(B * S, C)
        h = self.group_norm(x).view(B * S, C)
        # [B*S][2C] and [C][B*S] (V bias folded into self.bo): one paired launch (layers.Attention)
        qk, vt = ops.linear_pair({"x": h, "w": self.wqk, "bias": self.bqk}, {"x": self.wv, "w": h})
        if self.head_dim in (64, 96, 128, 160) and not self.force_gemm_path:
            o = ops.attention(qk, qk[:, self.inner:], vt, B=B, H=self.heads, D=self.head_dim, Sq=S, Skv=S, Skv_alloc=S,
                              q_row_stride=2 * self.inner, k_row_stride=2 * self.inner,
                              q_batch_stride=S * 2 * self.inner, k_batch_stride=S * 2 * self.inner,
                              vt_ld=B * S, vt_batch_stride=S, scale=self.scale)
        else:
            if self.heads != 1:
                raise ValueError("VaeAttention GEMM path supports a single head")
            # One head of D = 512: scores GEMM (fp32) -> row softmax -> P.V GEMM, in blocks of QBLK query rows so that the score
            # matrix held at any time is QBLK x S (268 MB at S = 16 384 instead of 1 GB + 0.5 GB of probabilities).  Measured
            # (profiles/README.md): the two GEMMs run at ~900 TFLOP/s here; a flash kernel at D = 512 would either recompute Q.K^T per
            # 128-wide output slice (2.5x the flops) or exchange partial scores between waves every tile -- slower than this path.
            o = torch.empty((B * S, C), device=x.device, dtype=bf16)
            QBLK = 4096
            nb = min(QBLK, S)
            scores = torch.empty((nb, S), device=x.device, dtype=torch.float32)
            probs = torch.empty((nb, S), device=x.device, dtype=bf16)
            for b in range(B):
                k = qk[b * S:(b + 1) * S, self.inner:]
                for q0 in range(0, S, nb):
                    n = min(nb, S - q0)
                    q = qk[b * S + q0:b * S + q0 + n, :self.inner]
                    ops.linear(q, k, alpha=self.scale, out_f32=True, out=scores[:n])       # [n][S] fp32
                    ops.softmax_rows(scores[:n], out=probs[:n])                             # [n][S] bf16
                    ops.linear(probs[:n], vt[:, b * S:(b + 1) * S], out=o[b * S + q0:b * S + q0 + n])
        y = ops.linear(o, self.wo, self.bo, residual=res)
        return y.view(B, H, W_, C)


class AutoencoderKL(PretrainedMixin):
    """Drop-in for the reference ``AutoencoderKL`` decode path (inference, bf16, HIP device only)."""

    def __init__(self, **kwargs):
        unknown = set(kwargs) - set(_DEFAULTS)
        if unknown:
            raise TypeError(f"AutoencoderKL: unexpected config keys {sorted(unknown)}")
        cfg = dict(_DEFAULTS)
        cfg.update(kwargs)
        self.config = FrozenConfig(cfg)
        if self.config.act_fn != "silu":
            raise ValueError("AutoencoderKL: only act_fn='silu' is supported")
        for t in self.config.up_block_types:
            if t != "UpDecoderBlock2D":
                raise ValueError(f"{t} does not exist.")
        self.dtype = bf16
        self.device = None
        self._built = False
        self.post_quant_conv = None

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], device="cuda", strict: bool = False):
        """Packs the decoder half (``decoder.*``, ``post_quant_conv.*``) of a reference AutoencoderKL state_dict."""
        c = self.config
        w = Weights(state_dict, device)
        self.device = torch.device(device)
        groups, eps = c.norm_num_groups, 1e-6
        boc = tuple(c.block_out_channels)
        lat = c.latent_channels

        def pad_rows(t, rows):
            if t.shape[0] == rows:
                return t
            out = torch.zeros((rows,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
            out[: t.shape[0]] = t
            return out

        self.latent_pad = lat
        if c.use_post_quant_conv:
            # 1x1 conv lat -> lat; output channels padded to a multiple of 8 for the thin-input kernel
            lat_p = ((lat + 7) // 8) * 8
            pw = w.get("post_quant_conv.weight").reshape(lat, lat)
            self.pqc_w = pad_rows(pw, lat_p).contiguous()
            self.pqc_b = pad_rows(w.get("post_quant_conv.bias"), lat_p).contiguous()
            self.post_quant_conv = True
            self.latent_pad = lat_p
        ci = w.get("decoder.conv_in.weight")  # [C][lat][3][3]
        if self.latent_pad != lat:
            cip = torch.zeros((ci.shape[0], self.latent_pad, 3, 3), device=ci.device, dtype=ci.dtype)
            cip[:, :lat] = ci
            ci = cip
        self.conv_in_w = ops.pack_conv_weight(ci)
        self.conv_in_b = w.get("decoder.conv_in.bias")

        self.mid_res0 = ResnetBlock2D(w, "decoder.mid_block.resnets.0", groups, eps)
        self.mid_attn = VaeAttention(w, "decoder.mid_block.attentions.0", groups, eps) \
            if c.mid_block_add_attention else None
        self.mid_res1 = ResnetBlock2D(w, "decoder.mid_block.resnets.1", groups, eps)

        self.up = []
        n = len(boc)
        for i in range(n):
            pre = f"decoder.up_blocks.{i}"
            stage = {"resnets": [ResnetBlock2D(w, f"{pre}.resnets.{j}", groups, eps)
                                 for j in range(c.layers_per_block + 1)], "up": None}
            if i != n - 1:
                stage["up"] = Upsample2D(w, f"{pre}.upsamplers.0")
            self.up.append(stage)
        self.conv_norm_out = GroupNorm(w, "decoder.conv_norm_out", groups, eps)
        self.conv_out_w = ops.pack_conv_weight(w.get("decoder.conv_out.weight"))
        self.conv_out_b = w.get("decoder.conv_out.bias")
        if strict:
            extra = [k for k in w.unused() if k.startswith("decoder.") or k.startswith("post_quant_conv.")]
            if extra:
                raise RuntimeError(f"unexpected decoder keys: {extra[:8]}")
        self._built = True
        return self

    def to(self, *args, **kwargs):
        return check_to(self, args, kwargs)

    def eval(self):
        return self

    def encode(self, *a, **k):
        raise NotImplementedError("diffusers_amd.AutoencoderKL implements the decode hot path only")

    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None, *, latents_div: float = 1.0,
               latents_add: float = 0.0, postprocess: Optional[str] = None):
        """autoencoder_kl.py:214-240.  ``latents_div`` / ``latents_add`` fuse the pipeline's
        ``latents / scaling_factor (+ shift_factor)`` (pipeline_stable_diffusion_xl.py:1283, pipeline_flux.py:960) into
        the first conv's input read.  ``postprocess`` ("pt" / "np" / "uint8") fuses the pipeline's
        ``image_processor.postprocess`` (image_processor.py:738-786) into the last pass of ``conv_out``: the sample is then the
        finished image ([0, 1] fp32 NCHW / NHWC, or NHWC bytes) instead of the bf16 decoder output in [-1, 1]."""
        if not self._built:
            raise RuntimeError("AutoencoderKL: call load_state_dict() first")
        ops.require_hip(z, "z")
        z = z.contiguous()
        if self.post_quant_conv:
            x = ops.conv_thin_in(z, self.pqc_w, self.pqc_b, ksize=1, in_nchw=True, in_div=latents_div, in_add=latents_add)
            x = ops.conv_thin_in(x, self.conv_in_w, self.conv_in_b, ksize=3, in_nchw=False)
        else:
            x = ops.conv_thin_in(z, self.conv_in_w, self.conv_in_b, ksize=3, in_nchw=True, in_div=latents_div,
                                 in_add=latents_add)
        x = self.mid_res0(x)
        if self.mid_attn is not None:
            x = self.mid_attn(x)
        x = self.mid_res1(x)
        for st in self.up:
            for rn in st["resnets"]:
                x = rn(x)
            if st["up"] is not None:
                x = st["up"](x)
        x = self.conv_norm_out(x, silu=True)
        img = ops.conv_thin_out(x, self.conv_out_w, self.conv_out_b, postprocess=postprocess)
        if not return_dict:
            return (img,)
        return DecoderOutput(sample=img)
