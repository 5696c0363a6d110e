"""AutoencoderKLWan.decode on the gfx950 kernels (the step right after the Wan denoising loop).

Mirrors models/autoencoders/autoencoder_kl_wan.py: ``_decode`` :1187-1217, ``WanDecoder3d`` :788-914, ``WanUpBlock``
:719-785, ``WanMidBlock`` :434-470, ``WanAttentionBlock`` :389-431, ``WanResidualBlock`` :315-386, ``WanResample``
:224-312, ``WanRMS_norm`` :176-206, ``WanCausalConv3d`` :131-173 (Wan 2.1 layout: is_residual=False, patch_size=None).

MI355X-first restatement.  The reference decodes ONE latent frame per decoder call and threads the previous two input
frames of every causal conv through a python-side feature cache -- a schedule made for 24-80 GB cards.  With 288 GB of
HBM the whole clip is resident: activations are channels-last frames ``[T][H][W][C]`` (frames = the batch dimension of
every per-frame op) and

  * a causal 3x3x3 conv is three implicit-GEMM 3x3 launches, one per temporal tap, over frame-shifted views of the
    SAME buffers (tap kt reads frames [0, T-2+kt) and accumulates into output frames [2-kt, T) in place); the zero
    frames in front of the clip are simply the frames a tap does not visit;
  * the temporal upsampler's (3,1,1) conv runs on frames 1.. (the reference's first chunk is "Rep": it bypasses
    time_conv and is not part of its history, :275-296), its 2C output channels are re-laid as frame pairs by one copy
    kernel, and nearest-2x + Conv2d is one fused-gather launch over all frames;
  * per-frame single-head attention (head_dim = C = 384) is QK^T (fp32) -> row softmax -> PV on the MFMA GEMM.

Channel counts that are not a multiple of 64 (96 at full size) are zero-padded to the next multiple once, at load:
padded weights / gammas are zero, so padded channels stay exactly zero through every op.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from .config_utils import check_to
from .loading import PretrainedMixin
from .autoencoder_kl import DecoderOutput
from .layers import Weights
from .unet_2d_condition import FrozenConfig

bf16 = torch.bfloat16

_DEFAULTS = dict(
    base_dim=96, decoder_base_dim=None, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2, attn_scales=(),
    temperal_downsample=(False, True, True), dropout=0.0,
    latents_mean=(-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632,
                  -0.1922, -0.9497, 0.2503, -0.2921),
    latents_std=(2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382,
                 1.1253, 2.8251, 1.9160),
    is_residual=False, in_channels=3, out_channels=3, patch_size=None, scale_factor_temporal=4, scale_factor_spatial=8,
)


# Skip the MFMA steps over zero channel padding (da_gemm_params.k_valid).  Speed only; the switch exists for A/B timing.
K_SKIP = True


def _kv(c: int) -> int:
    return c if K_SKIP else 0


def _pad64(c: int) -> int:
    return (c + 63) // 64 * 64


def _pad_to(t: torch.Tensor, shape) -> torch.Tensor:
    """Zero-pad every dim of ``t`` up to ``shape``."""
    if tuple(t.shape) == tuple(shape):
        return t.contiguous()
    out = torch.zeros(shape, device=t.device, dtype=t.dtype)
    out[tuple(slice(0, s) for s in t.shape)] = t
    return out


class CausalConv3d:
    """WanCausalConv3d with kernel (kt, k, k), k in {1, 3}: one packed [Cout_p][k*k*Cin_p] weight per temporal tap."""

    def __init__(self, w: Weights, prefix: str, cout_pad: Optional[int] = None):
        wt = w.get(prefix + ".weight")                         # [Cout][Cin][kt][k][k]
        b = w.get(prefix + ".bias")
        cout, cin, kt, k, _ = wt.shape
        self.kt, self.k = kt, k
        self.cin, self.cin_p, self.cout_p = cin, _pad64(cin), cout_pad or _pad64(cout)
        wt = _pad_to(wt, (self.cout_p, self.cin_p, kt, k, k))
        self.bias = _pad_to(b, (self.cout_p,))
        self.taps = [ops.pack_conv_weight(wt[:, :, i].contiguous()) if k == 3
                     else wt[:, :, i, 0, 0].contiguous() for i in range(kt)]

    def _one(self, x, wt, bias, residual, out):
        # k_valid: the zero channels that pad Cin up to the K granule (96 -> 128 at full size) are skipped, not multiplied
        if self.k == 3:
            return ops.conv2d_nhwc(x, wt, bias, ksize=3, residual=residual, out=out, k_valid=_kv(self.cin))
        T, H, W_, C = x.shape
        y = ops.linear(x.view(T * H * W_, C), wt, bias,
                       residual=None if residual is None else residual.view(T * H * W_, self.cout_p),
                       out=None if out is None else out.view(T * H * W_, self.cout_p), k_valid=_kv(self.cin))
        return y.view(T, H, W_, self.cout_p)

    def __call__(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: [T][H][W][Cin_p] -> [T][H][W][Cout_p] (+ residual).  The last tap (current frame) carries the bias and the
        residual and writes every output frame; earlier taps accumulate into the frames that have that much history."""
        T = x.shape[0]
        out = self._one(x, self.taps[self.kt - 1], self.bias, residual, None)
        for back in range(1, self.kt):
            if T > back:
                self._one(x[:T - back], self.taps[self.kt - 1 - back], None, out[back:], out[back:])
        return out


class RMSNorm:
    """WanRMS_norm (channel-first in the reference = the last dim here), bias-free; optional fused SiLU."""

    def __init__(self, w: Weights, name: str):
        g = w.get(name).reshape(-1)
        self.real = g.numel()
        self.gamma = _pad_to(g, (_pad64(self.real),))

    def __call__(self, x, silu: bool = False):
        return ops.rmsnorm_channels(x, self.gamma, real_channels=self.real, silu=silu)


class ResidualBlock:
    """WanResidualBlock: RMS-norm -> SiLU -> causal conv, twice, + (1x1x1 conv) shortcut folded into conv2's epilogue."""

    def __init__(self, w: Weights, p: str):
        self.norm1, self.norm2 = RMSNorm(w, p + ".norm1.gamma"), RMSNorm(w, p + ".norm2.gamma")
        self.conv1, self.conv2 = CausalConv3d(w, p + ".conv1"), CausalConv3d(w, p + ".conv2")
        self.shortcut = CausalConv3d(w, p + ".conv_shortcut") if w.has(p + ".conv_shortcut.weight") else None

    def __call__(self, x):
        h = self.shortcut(x) if self.shortcut is not None else x
        y = self.conv1(self.norm1(x, silu=True))
        return self.conv2(self.norm2(y, silu=True), residual=h)


class AttentionBlock:
    """WanAttentionBlock: per-frame, one head of width C.  to_qkv's V bias is folded into the projection bias (softmax
    rows sum to 1), V is produced directly transposed ([C][positions]) as the PV GEMM's weight operand, and the key
    axis is zero-padded to a multiple of 64 (the GEMM's K granule)."""

    def __init__(self, w: Weights, p: str):
        self.norm = RMSNorm(w, p + ".norm.gamma")
        c = self.norm.real
        cp = _pad64(c)
        self.c, self.cp = c, cp
        wqkv = w.get(p + ".to_qkv.weight").reshape(3 * c, c)
        bqkv = w.get(p + ".to_qkv.bias")
        wq, wk, wv = (_pad_to(wqkv[i * c:(i + 1) * c], (cp, cp)) for i in range(3))
        self.wqk = torch.cat([wq, wk], 0).contiguous()
        self.bqk = torch.cat([_pad_to(bqkv[:c], (cp,)), _pad_to(bqkv[c:2 * c], (cp,))]).contiguous()
        self.wv = wv
        wo = w.get(p + ".proj.weight").reshape(c, c)
        bo = w.get(p + ".proj.bias")
        self.bo = _pad_to((wo.float() @ bqkv[2 * c:].float() + bo.float()).to(bf16), (cp,))
        self.wo = _pad_to(wo, (cp, cp))
        self.scale = c ** -0.5

    def __call__(self, x):
        T, H, W_, cp = x.shape
        S = H * W_
        Sp = _pad64(S)
        h = self.norm(x).view(T * S, cp)
        qk = ops.linear(h, self.wqk, self.bqk)                               # [T*S][2 cp]
        o = torch.empty((T * S, cp), device=x.device, dtype=bf16)
        vt = torch.zeros((cp, Sp), device=x.device, dtype=bf16)              # zero key padding, reused per frame
        probs = torch.zeros((S, Sp), device=x.device, dtype=bf16)
        for t in range(T):
            rows = slice(t * S, (t + 1) * S)
            ops.linear(self.wv, h[rows], out=vt[:, :S])                      # V^T of this frame
            scores = ops.linear(qk[rows, :cp], qk[rows, cp:], alpha=self.scale, out_f32=True)    # [S][S] fp32
            ops.softmax_rows(scores, out=probs)
            ops.linear(probs, vt, out=o[rows])
        y = ops.linear(o, self.wo, self.bo, residual=x.view(T * S, cp))
        return y.view(T, H, W_, cp)


class Resample:
    """WanResample 'upsample2d' / 'upsample3d'."""

    def __init__(self, w: Weights, p: str, temporal: bool):
        wt = w.get(p + ".resample.1.weight")                                 # Conv2d(dim, dim // 2, 3, padding=1)
        cout, cin = wt.shape[:2]
        self.cin, self.cin_p, self.cout_p = cin, _pad64(cin), _pad64(cout)
        self.w = ops.pack_conv_weight(_pad_to(wt, (self.cout_p, self.cin_p, 3, 3)))
        self.b = _pad_to(w.get(p + ".resample.1.bias"), (self.cout_p,))
        self.time = None
        if temporal:
            tw = w.get(p + ".time_conv.weight")                              # [2 dim][dim][3][1][1]
            tb = w.get(p + ".time_conv.bias")
            c = tw.shape[1]
            cp = _pad64(c)
            # the two channel halves are the two output frames (:297-299): pad each half separately
            tw2 = torch.cat([_pad_to(tw[j * c:(j + 1) * c], (cp, cp, 3, 1, 1)) for j in range(2)], 0)
            tb2 = torch.cat([_pad_to(tb[j * c:(j + 1) * c], (cp,)) for j in range(2)])
            self.time = CausalConv3d(Weights({"t.weight": tw2, "t.bias": tb2}, tw.device), "t", cout_pad=2 * cp)
            self.cp = cp

    def __call__(self, x):
        T, H, W_, cp = x.shape
        if self.time is not None and T > 1:
            y = self.time(x[1:])                                             # [T-1][H][W][2 cp]
            x2 = torch.empty((2 * T - 1, H, W_, cp), device=x.device, dtype=bf16)
            x2[0].copy_(x[0])                                                # the "Rep" frame passes through
            ops.permute_0213(y.view(T - 1, H * W_, 2, cp), out=x2[1:])       # [T-1][HW][2][cp] -> [T-1][2][HW][cp]
            x = x2
        return ops.conv2d_nhwc(x, self.w, self.b, ksize=3, up=True, k_valid=_kv(self.cin))


class AutoencoderKLWan(PretrainedMixin):
    """Drop-in for the reference ``AutoencoderKLWan`` decode path (inference, bf16, HIP device only)."""

    def __init__(self, **kwargs):
        unknown = set(kwargs) - set(_DEFAULTS)
        if unknown:
            raise TypeError(f"AutoencoderKLWan: unexpected config keys {sorted(unknown)}")
        cfg = dict(_DEFAULTS)
        cfg.update(kwargs)
        self.config = FrozenConfig(cfg)
        if self.config.is_residual or self.config.patch_size is not None:
            raise NotImplementedError("AutoencoderKLWan: the Wan 2.2 residual / patchified layout is not built")
        if self.config.out_channels > 4:
            raise ValueError("AutoencoderKLWan: out_channels <= 4")
        self.dtype = bf16
        self.device = None
        self._built = False

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], device="cuda", strict: bool = False):
        """Packs the decoder half (``decoder.*``, ``post_quant_conv.*``) of a reference AutoencoderKLWan state_dict."""
        c = self.config
        w = Weights(state_dict, device)
        self.device = torch.device(device)
        z = c.z_dim
        mult = list(c.dim_mult)
        t_up = list(c.temperal_downsample)[::-1]

        # post_quant_conv (1x1x1): thin-input kernel on the [z][T*H][W] view of the latents; output padded to 64 channels.
        # ``decode(..., denormalize=True)`` takes the PIPELINE's latents and folds ``latents / (1 / std) + mean``
        # (pipeline_wan.py:653-661) into this conv:  W (s * x + m) + b = (W diag(s)) x + (W m + b).
        pw = w.get("post_quant_conv.weight").reshape(z, z).float()
        pb = w.get("post_quant_conv.bias").float()
        std = torch.tensor(c.latents_std, device=pw.device, dtype=torch.float32)
        mean = torch.tensor(c.latents_mean, device=pw.device, dtype=torch.float32)
        self.zp = 64
        self.pqc_w = _pad_to(pw.to(bf16), (self.zp, z))
        self.pqc_b = _pad_to(pb.to(bf16), (self.zp,))
        self.pqc_w_dn = _pad_to((pw * std[None, :]).to(bf16), (self.zp, z))
        self.pqc_b_dn = _pad_to((pw @ mean + pb).to(bf16), (self.zp,))

        self.conv_in = CausalConv3d(w, "decoder.conv_in")
        self.mid_res0 = ResidualBlock(w, "decoder.mid_block.resnets.0")
        self.mid_attn = AttentionBlock(w, "decoder.mid_block.attentions.0")
        self.mid_res1 = ResidualBlock(w, "decoder.mid_block.resnets.1")
        self.up = []
        for i in range(len(mult)):
            pre = f"decoder.up_blocks.{i}"
            stage = {"resnets": [ResidualBlock(w, f"{pre}.resnets.{j}") for j in range(c.num_res_blocks + 1)], "up": None}
            if i != len(mult) - 1:
                stage["up"] = Resample(w, f"{pre}.upsamplers.0", bool(t_up[i]))
            self.up.append(stage)
        self.norm_out = RMSNorm(w, "decoder.norm_out.gamma")
        self.conv_out = CausalConv3d(w, "decoder.conv_out", cout_pad=4)
        if strict:
            extra = [k for k in w.unused() if k.startswith(("decoder.", "post_quant_conv."))]
            if extra:
                raise RuntimeError(f"unexpected decoder keys: {extra[:8]}")
        self._built = True
        return self

    def to(self, *args, **kwargs):
        return check_to(self, args, kwargs)

    def eval(self):
        return self

    def encode(self, *a, **k):
        raise NotImplementedError("diffusers_amd.AutoencoderKLWan implements the decode path only")

    def _decode_one(self, z: torch.Tensor, denormalize: bool, out_f32: bool) -> torch.Tensor:
        """z: [z_dim][T][H][W] bf16 -> [1][3][1 + 4 (T - 1)][8H][8W]."""
        Z, T, H, W_ = z.shape
        pw, pb = (self.pqc_w_dn, self.pqc_b_dn) if denormalize else (self.pqc_w, self.pqc_b)
        x = ops.conv_thin_in(z.view(1, Z, T * H, W_), pw, pb, ksize=1, in_nchw=True).view(T, H, W_, self.zp)
        x = self.conv_in(x)
        x = self.mid_res0(x)
        x = self.mid_attn(x)
        x = self.mid_res1(x)
        for st in self.up:
            for rn in st["resnets"]:
                x = rn(x)
            if st["up"] is not None:
                x = st["up"](x)
        x = self.conv_out(self.norm_out(x, silu=True))
        return ops.frames_to_ncthw(x, batch=1, channels=self.config.out_channels, lo=-1.0, hi=1.0, out_f32=out_f32)

    def decode(self, z: torch.Tensor, return_dict: bool = True, *, denormalize: bool = False, out_f32: bool = False):
        """autoencoder_kl_wan.py:1219-1241.  z: [B][z_dim][T][H][W]; bf16, or the pipeline's fp32 latents (cast once).
        ``denormalize=True``: z are the pipeline's normalised latents (see load_state_dict)."""
        if not self._built:
            raise RuntimeError("AutoencoderKLWan: call load_state_dict() first")
        ops.require_hip(z, "z", (bf16, torch.float32))
        if z.dim() != 5 or z.shape[1] != self.config.z_dim:
            raise ValueError(f"z must be [B][{self.config.z_dim}][T][H][W]")
        z = z.contiguous()
        if z.dtype == torch.float32:
            z = ops.cast_f32_bf16(z)
        vids = [self._decode_one(z[b], denormalize, out_f32) for b in range(z.shape[0])]
        video = vids[0] if len(vids) == 1 else torch.cat(vids, 0)
        if not return_dict:
            return (video,)
        return DecoderOutput(sample=video)
