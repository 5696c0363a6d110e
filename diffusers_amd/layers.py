"""Engine-side building blocks: each class owns its weights in the packed layout the gfx950 kernels consume and mirrors
one reference nn.Module (cited per class, paths under the reference's src/diffusers/).

Weights are taken from a reference-format ``state_dict`` (same key names and tensor shapes as the reference modules'
``state_dict()``), so checkpoints / seeded reference models load unchanged.  Activations are channels-last bf16:
images [B][H][W][C], tokens [B*S][C].
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from . import _lib as L
from . import ops

bf16 = torch.bfloat16


class Weights:
    """Reference-format state_dict view that moves tensors to the engine device as bf16 and tracks consumption."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device):
        self.sd = state_dict
        self.device = torch.device(device)
        self.used = set()

    def has(self, name: str) -> bool:
        return name in self.sd

    def get(self, name: str) -> torch.Tensor:
        if name not in self.sd:
            raise KeyError(f"missing weight '{name}' in state_dict")
        self.used.add(name)
        return self.sd[name].detach().to(device=self.device, dtype=bf16).contiguous()

    def get_f32(self, name: str) -> torch.Tensor:
        """Parameters the reference keeps in fp32 (``_keep_in_fp32_modules``, e.g. Wan's scale_shift_table)."""
        if name not in self.sd:
            raise KeyError(f"missing weight '{name}' in state_dict")
        self.used.add(name)
        return self.sd[name].detach().to(device=self.device, dtype=torch.float32).contiguous()

    def opt(self, name: str) -> Optional[torch.Tensor]:
        return self.get(name) if name in self.sd else None

    def unused(self):
        return sorted(set(self.sd.keys()) - self.used)


class Linear:
    """nn.Linear; weight [N][K] is already the kernel's layout."""

    def __init__(self, w: Weights, prefix: str):
        self.weight = w.get(prefix + ".weight")
        self.bias = w.opt(prefix + ".bias")

    def __call__(self, x, **kw):
        return ops.linear(x, self.weight, self.bias, **kw)

    @property
    def in_features(self) -> int:          # what reference pipelines read off nn.Linear (e.g. add_embedding.linear_1)
        return self.weight.shape[1]

    @property
    def out_features(self) -> int:
        return self.weight.shape[0]


class Conv3x3:
    """nn.Conv2d(k=3, pad=1) as implicit GEMM (resnet.py:340,:365; downsampling.py:145; upsampling.py:186)."""

    def __init__(self, w: Weights, prefix: str):
        self.weight = ops.pack_conv_weight(w.get(prefix + ".weight"))
        self.bias = w.opt(prefix + ".bias")
        self.ksize = 3

    def __call__(self, x, **kw):
        return ops.conv2d_nhwc(x, self.weight, self.bias, ksize=3, **kw)


class Conv1x1:
    """nn.Conv2d(k=1): the resnet shortcut (resnet.py:373) and SD1.5's proj_in/proj_out (transformer_2d.py:468,:505)."""

    def __init__(self, w: Weights, prefix: str):
        wt = w.get(prefix + ".weight")
        self.weight = wt.reshape(wt.shape[0], -1).contiguous()
        self.bias = w.opt(prefix + ".bias")

    def __call__(self, x, x2=None, **kw):
        if x2 is None:
            B, H, W_, C = x.shape
            y = ops.linear(x.view(B * H * W_, C), self.weight, self.bias, **kw)
            return y.view(B, H, W_, -1)
        return ops.conv2d_nhwc(x, self.weight, self.bias, ksize=1, x2=x2, **kw)


class GroupNorm:
    def __init__(self, w: Weights, prefix: str, groups: int, eps: float):
        self.weight = w.get(prefix + ".weight")
        self.bias = w.get(prefix + ".bias")
        self.groups, self.eps = groups, eps

    def __call__(self, x, silu=False, x2=None):
        return ops.group_norm_nhwc(x, self.weight, self.bias, self.groups, self.eps, silu=silu, x2=x2)


class LayerNorm:
    def __init__(self, w: Weights, prefix: str, eps: float = 1e-5, affine: bool = True):
        self.weight = w.get(prefix + ".weight") if affine else None
        self.bias = w.get(prefix + ".bias") if affine else None
        self.eps = eps

    def __call__(self, x, **kw):
        return ops.layer_norm(x, self.weight, self.bias, self.eps, **kw)


class ResnetBlock2D:
    """models/resnet.py:319-377.  GroupNorm+SiLU -> conv3x3 (+ time_emb_proj(SiLU(temb)) per batch) -> GroupNorm+SiLU
    -> conv3x3 (+ shortcut(x) residual, / output_scale_factor).  The input may be a (x, skip) pair: the U-Net skip
    concat (unet_2d_blocks.py:2444) is never materialised, GroupNorm and the 1x1 shortcut read both sources."""

    def __init__(self, w: Weights, prefix: str, groups: int, eps: float, output_scale_factor: float = 1.0):
        self.norm1 = GroupNorm(w, prefix + ".norm1", groups, eps)
        self.conv1 = Conv3x3(w, prefix + ".conv1")
        self.has_temb = w.has(prefix + ".time_emb_proj.weight")
        if self.has_temb:
            self.time_emb_proj = Linear(w, prefix + ".time_emb_proj")
        self.norm2 = GroupNorm(w, prefix + ".norm2", groups, eps)
        self.conv2 = Conv3x3(w, prefix + ".conv2")
        self.shortcut = Conv1x1(w, prefix + ".conv_shortcut") if w.has(prefix + ".conv_shortcut.weight") else None
        self.out_scale = 1.0 / output_scale_factor

    def __call__(self, x, temb=None, skip=None):
        # temb: [B][temb_channels] bf16 (already the raw embedding; SiLU is fused into the skinny linear), or the projections of
        # every block of the model computed by one launch (TimeProjections below): this block's columns of that tensor
        h = self.norm1(x, silu=True, x2=skip)
        tvec = None
        if self.has_temb and temb is not None:
            if isinstance(temb, ProjectedTemb):
                tvec = temb.all[:, self.temb_off:self.temb_off + self.temb_c]
            elif self.time_emb_proj is None:
                raise ValueError("ResnetBlock2D: this block's time_emb_proj lives in the model's TimeProjections; pass its output")
            else:
                tvec = ops.linear_small_m(temb, self.time_emb_proj.weight, self.time_emb_proj.bias, act_in=L.ACT_SILU)
        h = self.conv1(h, rowvec=tvec)
        h = self.norm2(h, silu=True)
        if self.shortcut is not None:
            res = self.shortcut(x, x2=skip)
        else:
            if skip is not None:
                raise ValueError("ResnetBlock2D: concat input requires a conv_shortcut")
            res = x
        return self.conv2(h, residual=res, out_scale=self.out_scale)


class ProjectedTemb:
    """[B][sum of the blocks' channels] bf16: time_emb_proj(SiLU(temb)) of every ResnetBlock2D of a model, side by side."""

    def __init__(self, t: torch.Tensor):
        self.all = t


class TimeProjections:
    """``time_emb_proj(nonlinearity(temb))`` (resnet.py:345-349) of EVERY ResnetBlock2D of a U-Net as ONE skinny GEMM per forward:
    the blocks' weights stacked along the output dimension [sum C_i][temb_channels].  The embedding is the same vector for all of
    them, so the 19 (SDXL) / 24 (SD1.5) / 34 (ddpm-cat) latency-bound launches of a step become one that streams the same bytes;
    every output column is computed by its own wave from its own weight row, so the values are bit-identical to the per-block
    launches.  Each block reads its columns of the result as the conv's per-batch row vector (`ld_rowvec` = the full width)."""

    def __init__(self, resnets):
        rs = [r for r in resnets if r.has_temb]
        self.weight = self.bias = None
        if not rs or os.environ.get("DIFFUSERS_AMD_TEMB_STACK", "1") == "0":      # A/B knob: the blocks keep their own launches
            return
        biased = [r.time_emb_proj.bias is not None for r in rs]
        if any(biased) and not all(biased):
            return                                   # (no such model in scope: the blocks keep their own launches)
        self.weight = torch.cat([r.time_emb_proj.weight for r in rs], dim=0).contiguous()
        self.bias = torch.cat([r.time_emb_proj.bias for r in rs], dim=0).contiguous() if all(biased) else None
        off = 0
        for r in rs:
            r.temb_off, r.temb_c = off, r.time_emb_proj.weight.shape[0]
            off += r.temb_c
            r.time_emb_proj = None                   # one copy of the weight: the stacked one

    def __call__(self, temb: torch.Tensor):
        if self.weight is None:
            return temb
        return ProjectedTemb(ops.linear_small_m(temb, self.weight, self.bias, act_in=L.ACT_SILU))


class Downsample2D:
    """models/downsampling.py:130-147 (use_conv=True): conv3x3 stride 2 with padding=1, or with padding=0 after a
    (0, 1, 0, 1) zero pad (the DDPM U-Nets, ``downsample_padding=0``)."""

    def __init__(self, w: Weights, prefix: str, padding: int = 1):
        if padding not in (0, 1):
            raise ValueError("Downsample2D: padding must be 0 or 1")
        self.conv = Conv3x3(w, prefix + ".conv")
        self.padding = padding

    def __call__(self, x):
        if self.padding == 1:
            return self.conv(x, stride=2)
        return self.conv(x, stride=2, pad=0, pad_after=1)


class Upsample2D:
    """models/upsampling.py:140-192 (use_conv=True): nearest-2x interpolate fused into the conv's gather."""

    def __init__(self, w: Weights, prefix: str):
        self.conv = Conv3x3(w, prefix + ".conv")

    def __call__(self, x):
        return self.conv(x, up=True)


class CrossKV:
    """Step-invariant cross-attention keys / transposed values of one attention layer (hoisted out of the loop)."""

    __slots__ = ("k", "vt", "skv", "skv_alloc", "batch", "bias", "buf")

    def __init__(self, k, vt, skv, skv_alloc, batch, bias=None, buf=None):
        self.k, self.vt, self.skv, self.skv_alloc, self.batch = k, vt, skv, skv_alloc, batch
        self.buf = buf     # the ONE allocation that holds k then vt (a prefetch hint covers both), or None
        self.bias = bias   # [batch][1][1][ceil64(skv)] additive key mask (encoder_attention_mask), or None


KERNEL_HEAD_DIMS = (64, 96, 128, 160)  # head sizes of the flash kernel (csrc/attention.hip)


def kernel_head_dim(d: int) -> int:
    for k in KERNEL_HEAD_DIMS:
        if d <= k:
            return k
    raise ValueError(f"attention head_dim {d} exceeds the largest flash-kernel head size {KERNEL_HEAD_DIMS[-1]}")


def pad_head_rows(w: torch.Tensor, heads: int, d: int, dp: int) -> torch.Tensor:
    """[heads*d][K] projection rows -> [heads*dp][K] with zero rows after each head (SD1.5 head dims 40 / 80 run on the
    64 / 96 wide kernels: zero q/k channels add nothing to q.k, zero v channels produce zero outputs)."""
    if d == dp:
        return w
    out = torch.zeros((heads, dp) + tuple(w.shape[1:]), device=w.device, dtype=w.dtype)
    out[:, :d] = w.view((heads, d) + tuple(w.shape[1:]))
    return out.view((heads * dp,) + tuple(w.shape[1:])).contiguous()


def pad_head_cols(w: torch.Tensor, heads: int, d: int, dp: int) -> torch.Tensor:
    """[N][heads*d] to_out weight -> [N][heads*dp] with zero columns at the padded channels."""
    if d == dp:
        return w
    out = torch.zeros((w.shape[0], heads, dp), device=w.device, dtype=w.dtype)
    out[:, :, :d] = w.view(w.shape[0], heads, d)
    return out.view(w.shape[0], heads * dp).contiguous()


class Attention:
    """models/attention_processor.py:52-309 + AttnProcessor2_0 (:2696-2787): to_q/to_k/to_v (no bias), SDPA, to_out[0].
    Self-attention fuses Q and K into one GEMM and produces V already transposed (out^T = W_v . X^T) for the flash
    kernel; cross-attention K / V^T depend only on the text embeddings and come from :meth:`precompute_kv`.
    Head dims that are not a kernel size (SD1.5: 40, 80) are zero-padded once, in the packed weights."""

    def __init__(self, w: Weights, prefix: str, heads: int, cross: bool):
        self.heads = heads
        self.cross = cross
        wq = w.get(prefix + ".to_q.weight")
        wk = w.get(prefix + ".to_k.weight")
        wv = w.get(prefix + ".to_v.weight")
        inner = wq.shape[0]
        self.head_dim = inner // heads
        self.kdim = kernel_head_dim(self.head_dim)
        self.inner = heads * self.kdim                      # width of the (padded) q / k / v activations
        d, dp = self.head_dim, self.kdim
        wq, wk, wv = (pad_head_rows(t, heads, d, dp) for t in (wq, wk, wv))
        if cross:
            self.wq, self.wk, self.wv = wq, wk, wv
        else:
            self.wqk = torch.cat([wq, wk], dim=0).contiguous()
            self.wv = wv
            self.wqkv_ln = self.fold1 = None        # set by fold_norm1(): (gamma o [Wq; Wk; Wv]) and its s / c vectors
        self.to_out = Linear(w, prefix + ".to_out.0")
        self.to_out.weight = pad_head_cols(self.to_out.weight, heads, d, dp)
        self.scale = self.head_dim ** -0.5

    def precompute_kv(self, ehs_pad: torch.Tensor, batch: int, skv: int, skv_alloc: int, bias=None) -> CrossKV:
        """ehs_pad: [batch*skv_alloc][cross_dim], zero rows beyond skv in every batch.  ``bias``: the additive key mask
        built from ``encoder_attention_mask`` (see :func:`encoder_mask_bias`); it needs the masked flash kernel (D = 64)."""
        if bias is not None and self.kdim != 64:
            raise ValueError(f"encoder_attention_mask: the masked attention kernel exists for head sizes <= 64 "
                             f"(this layer: {self.head_dim})")
        rows, inner = ehs_pad.shape[0], self.wk.shape[0]
        buf = torch.empty(2 * rows * inner, device=ehs_pad.device, dtype=ehs_pad.dtype)
        k = ops.linear(ehs_pad, self.wk, out=buf[: rows * inner].view(rows, inner))
        vt = ops.linear(self.wv, ehs_pad, out=buf[rows * inner:].view(inner, rows))  # [inner][batch*skv_alloc] = V^T
        return CrossKV(k, vt, skv, skv_alloc, batch, bias, buf)

    def fold_norm(self, norm: "LayerNorm") -> None:
        """Cross-attention only: fold the LayerNorm in front of the block (attention.py:1030) into to_q."""
        self.wq_ln, self.fold = ops.fold_layernorm(self.wq, norm.weight, norm.bias, norm.eps)

    def fold_norm1(self, norm: "LayerNorm") -> None:
        """Self-attention only: fold norm1 (attention.py:986) into ONE fused to_q | to_k | to_v projection whose V block leaves the
        GEMM transposed (ops.linear_qkv): replaces the LayerNorm launch and the paired Q|K + V^T launch.  Needs a column origin
        2 * inner that the transposed-block tiles divide (a multiple of 80: SDXL's 1280 / 640 do; SD1.5's 320-wide level too)."""
        if (2 * self.inner) % 80 != 0:
            return
        wqkv = torch.cat([self.wqk, self.wv], dim=0).contiguous()
        self.wqkv_ln, self.fold1 = ops.fold_layernorm(wqkv, norm.weight, norm.bias, norm.eps)

    def __call__(self, x, batch: int, seq: int, residual, kv: Optional[CrossKV] = None, stats=None, stats_out=None):
        """x: [batch*seq][C], already normalised -- or, with ``stats`` (ops.RowStats of x), the un-normalised tokens of a
        cross-attention whose norm was folded into to_q; returns to_out(attn) + residual (``stats_out``: the row statistics
        of that result, for the next folded norm)."""
        Hh, D = self.heads, self.kdim
        if self.cross and ops.XATTN and kv.bias is None and ops.xattn_eligible(x.shape[0], self.inner, seq, D, kv.skv_alloc) \
                and kv.k.shape[1] == self.inner:
            # to_q and the 77-key attention in ONE launch (da_gemm_params.xa_*): the wave that finishes 32 queries x one head of q
            # runs that head's softmax(q k^T) v on its own registers -- no attention launch, no q round trip
            xa = {"k": kv.k, "vt": kv.vt, "skv": kv.skv, "skv_alloc": kv.skv_alloc, "seq": seq, "scale": self.scale}
            o = ops.linear(x, self.wq_ln, ln=(stats, self.fold), xattn=xa) if stats is not None else ops.linear(x, self.wq, xattn=xa)
        elif self.cross:
            pfk = {"prefetch": kv.buf} if (getattr(ops, "KV_PREFETCH", False) and kv.buf is not None) else {}
            q = ops.linear(x, self.wq_ln, ln=(stats, self.fold), **pfk) if stats is not None else ops.linear(x, self.wq, **pfk)
            o = ops.attention(q, kv.k, kv.vt, B=batch, H=Hh, D=D, Sq=seq, Skv=kv.skv, Skv_alloc=kv.skv_alloc,
                              q_row_stride=self.inner, k_row_stride=self.inner,
                              q_batch_stride=seq * self.inner, k_batch_stride=kv.skv_alloc * self.inner,
                              vt_ld=batch * kv.skv_alloc, vt_batch_stride=kv.skv_alloc, scale=self.scale, bias=kv.bias)
        else:
            if stats is not None:
                # norm1 folded: x is the UN-normalised residual stream, `stats` its row statistics (written by the GEMM that
                # produced it); ONE launch yields [M][2*inner] and the transposed [inner][M]
                qk, vt = ops.linear_qkv(x, self.wqkv_ln, 2 * self.inner, ln=(stats, self.fold1))
            else:
                # [M][2*inner] and [inner][M]: two problems, ONE launch (neither fills the 256 CUs alone at SDXL's sizes:
                # 160 + 80 tiles of 128x256); bit-identical to two launches
                qk, vt = ops.linear_pair({"x": x, "w": self.wqk}, {"x": self.wv, "w": x})
            o = ops.attention(qk, qk[:, self.inner:], vt, B=batch, H=Hh, D=D, Sq=seq, Skv=seq, Skv_alloc=seq,
                              q_row_stride=2 * self.inner, k_row_stride=2 * self.inner,
                              q_batch_stride=seq * 2 * self.inner, k_batch_stride=seq * 2 * self.inner,
                              vt_ld=batch * seq, vt_batch_stride=seq, scale=self.scale)
        return self.to_out(o, residual=residual, stats_out=stats_out)


class FeedForwardGEGLU:
    """models/attention.py:1682-1742 with activation_fn="geglu" (activations.py:93-124): GEGLU fused into the up
    projection's epilogue, bias + residual fused into the down projection's.  ``norm`` (the LayerNorm in front of the
    block, attention.py:1056): folded into the up projection -- ``w1_ln`` is (gamma o W1) in the packed GEGLU row order and
    ``fold`` carries the matching s / c vectors (ops.fold_layernorm), so ``__call__(x, ..., stats=)`` takes the
    UN-normalised tokens plus their row statistics."""

    def __init__(self, w: Weights, prefix: str, norm: Optional["LayerNorm"] = None):
        w1, b1 = w.get(prefix + ".net.0.proj.weight"), w.opt(prefix + ".net.0.proj.bias")
        wp, bp = ops.pack_geglu(w1, b1)
        self.w1, self.b1 = wp, bp
        self.out = Linear(w, prefix + ".net.2")
        self.w1_ln = self.fold = None
        if norm is not None:
            wl, fold = ops.fold_layernorm(w1, norm.weight, norm.bias, norm.eps)
            wl, _ = ops.pack_geglu(wl, None)
            n = w1.shape[0] // 2
            idx = torch.arange(n, device=w1.device).view(n // 32, 32)
            order = torch.cat([idx, idx + n], dim=1).reshape(-1)            # the row order of pack_geglu
            self.w1_ln = wl
            self.fold = ops.LNFold(fold.s.index_select(0, order).contiguous(), fold.c.index_select(0, order).contiguous(),
                                   fold.eps)

    def folds_here(self, rows: int) -> bool:
        """ops.LN_FOLD == 3 (default): fold norm3 only where the folded launch runs on the GEGLU projection's own eight-phase tile
        (ops.geglu_fold_on_k3: the shipped table sends the shape there) -- measured: + 1.4 % on an SDXL image there, a loss on the
        tiles SD1.5's 32 x 32 / 16 x 16 levels use (profiles/r06d_*)."""
        if self.w1_ln is None:
            return False
        return int(ops.LN_FOLD) == 1 or ops.geglu_fold_on_k3(rows, self.w1_ln.shape[0], self.w1_ln.shape[1])

    def __call__(self, x, residual, stats=None, stats_out=None):
        if stats is not None:
            h = ops.linear(x, self.w1_ln, self.b1, act=L.ACT_GEGLU, ln=(stats, self.fold))
        else:
            h = ops.linear(x, self.w1, self.b1, act=L.ACT_GEGLU)
        return self.out(h, residual=residual, stats_out=stats_out)


class BasicTransformerBlock:
    """models/attention.py:960-1080 (norm_type="layer_norm"): x += attn1(LN1 x); x += attn2(LN2 x, ctx); x += FF(LN3 x)."""

    def __init__(self, w: Weights, prefix: str, heads: int):
        self.norm1 = LayerNorm(w, prefix + ".norm1")
        self.attn1 = Attention(w, prefix + ".attn1", heads, cross=False)
        self.norm2 = LayerNorm(w, prefix + ".norm2")
        self.attn2 = Attention(w, prefix + ".attn2", heads, cross=True)
        self.norm3 = LayerNorm(w, prefix + ".norm3")
        # the folded copies (a second GEGLU up-projection, a second to_q, fp32 s / c vectors: ~1.7 GB for SDXL) exist only
        # when the opt-in fold was on at LOAD time; it is part of the packed-cache fingerprint (loading.py)
        self.folded = int(ops.LN_FOLD)          # 0 off, 1 norm2 + norm3, 2 norm2 only, 3 norm2 + norm3 where it pays (ops.LN_FOLD)
        self.ff = FeedForwardGEGLU(w, prefix + ".ff", norm=self.norm3 if self.folded in (1, 3) else None)
        if self.folded:
            self.attn2.fold_norm(self.norm2)
            self.attn1.fold_norm1(self.norm1)
        self.folds_norm1 = bool(self.folded) and self.attn1.wqkv_ln is not None

    def __call__(self, x, batch, seq, kv: CrossKV, stats_in=None, stats_out=None):
        """``stats_in``: row statistics of ``x`` written by the launch that produced it (the previous block's FF-down, or proj_in):
        norm1 is then applied inside the fused Q | K | V projection instead of as a kernel.  ``stats_out``: the FF-down of this
        block writes the statistics of ITS output there (for the next block's norm1)."""
        mode = int(ops.LN_FOLD)
        if mode and (not self.folded or (mode in (1, 3) and self.folded not in (1, 3))):
            raise RuntimeError("ops.LN_FOLD was switched on after this model was loaded: the folded weights are built at "
                               "load_state_dict() time (set DIFFUSERS_AMD_LN_FOLD / ops.LN_FOLD before loading)")
        if not mode:
            x = self.attn1(self.norm1(x), batch, seq, residual=x)
            x = self.attn2(self.norm2(x), batch, seq, residual=x, kv=kv)
            return self.ff(self.norm3(x), residual=x)
        use1 = stats_in is not None and self.folds_norm1
        if mode == 2:
            # norm2 never runs as a kernel: attn1.to_out also writes the row statistics of the residual stream it produces and
            # attn2.to_q applies the normalisation in its epilogue; norm1 likewise when the producer of x left its statistics
            st1 = ops.RowStats(x.shape[0], x.device)
            if use1:
                x = self.attn1(x, batch, seq, residual=x, stats=stats_in, stats_out=st1)
            else:
                x = self.attn1(self.norm1(x), batch, seq, residual=x, stats_out=st1)
            x = self.attn2(x, batch, seq, residual=x, kv=kv, stats=st1)
            return self.ff(self.norm3(x), residual=x, stats_out=stats_out)
        # norm2 and norm3 never run as kernels: attn1.to_out / attn2.to_out also write the row statistics of the
        # residual stream they produce, and attn2.to_q / the GEGLU projection apply the normalisation in their epilogue
        # (ops.linear(ln=)).  norm1 stays a kernel: its consumers are the Q|K projection AND the swapped V^T product,
        # where the normalised tokens are the column operand.
        fold3 = self.ff.folds_here(x.shape[0])      # (mode 3: per shape; mode 1: always)
        st1 = ops.RowStats(x.shape[0], x.device)
        st2 = ops.RowStats(x.shape[0], x.device) if fold3 else None
        if use1:
            x = self.attn1(x, batch, seq, residual=x, stats=stats_in, stats_out=st1)
        else:
            x = self.attn1(self.norm1(x), batch, seq, residual=x, stats_out=st1)
        x = self.attn2(x, batch, seq, residual=x, kv=kv, stats=st1, stats_out=st2)
        if fold3:
            return self.ff(x, residual=x, stats=st2, stats_out=stats_out)
        return self.ff(self.norm3(x), residual=x, stats_out=stats_out)


class Transformer2DModel:
    """models/transformers/transformer_2d.py:324-512 (continuous input).  In channels-last the reference's
    NCHW<->(B,HW,C) permutes vanish: GroupNorm -> proj_in -> blocks -> proj_out (+ residual fused)."""

    def __init__(self, w: Weights, prefix: str, heads: int, layers: int, groups: int):
        self.norm = GroupNorm(w, prefix + ".norm", groups, 1e-6)
        pw = w.get(prefix + ".proj_in.weight")
        self.proj_in_w = pw.reshape(pw.shape[0], -1).contiguous()  # Linear (SDXL) or 1x1 conv (SD1.5): same GEMM
        self.proj_in_b = w.opt(prefix + ".proj_in.bias")
        self.blocks = [BasicTransformerBlock(w, f"{prefix}.transformer_blocks.{i}", heads) for i in range(layers)]
        pw = w.get(prefix + ".proj_out.weight")
        self.proj_out_w = pw.reshape(pw.shape[0], -1).contiguous()
        self.proj_out_b = w.opt(prefix + ".proj_out.bias")

    def precompute_kv(self, ehs_pad, batch, skv, skv_alloc, bias=None):
        return [blk.attn2.precompute_kv(ehs_pad, batch, skv, skv_alloc, bias) for blk in self.blocks]

    def __call__(self, x, kvs):
        B, H, W_, C = x.shape
        res = x.view(B * H * W_, C)
        h = self.norm(x).view(B * H * W_, C)
        # LayerNorm fold (ops.LN_FOLD): the launch that produces a block's input -- proj_in, then each block's FF-down -- also writes
        # the row statistics the block's norm1 needs (applied inside its fused Q | K | V projection)
        chain = bool(ops.LN_FOLD) and ops.LN_FOLD_NORM1 and all(b.folds_norm1 for b in self.blocks)
        st = ops.RowStats(h.shape[0], h.device) if chain else None
        h = ops.linear(h, self.proj_in_w, self.proj_in_b, stats_out=st)
        for i, (blk, kv) in enumerate(zip(self.blocks, kvs)):
            nxt = ops.RowStats(h.shape[0], h.device) if chain and i + 1 < len(self.blocks) else None
            h = blk(h, B, H * W_, kv, stats_in=st, stats_out=nxt)
            st = nxt
        h = ops.linear(h, self.proj_out_w, self.proj_out_b, residual=res)
        return h.view(B, H, W_, C)


class TimestepEmbedding:
    """models/embeddings.py:1262-1308: Linear -> SiLU -> Linear on [B][dim] (skinny-M kernels)."""

    def __init__(self, w: Weights, prefix: str):
        self.l1 = Linear(w, prefix + ".linear_1")
        self.l2 = Linear(w, prefix + ".linear_2")

    def __call__(self, x, residual=None):
        h = ops.linear_small_m(x, self.l1.weight, self.l1.bias, act_out=L.ACT_SILU)
        return ops.linear_small_m(h, self.l2.weight, self.l2.bias, residual=residual)

    @property
    def linear_1(self) -> Linear:          # the reference's attribute names (pipeline_stable_diffusion_xl.py:737)
        return self.l1

    @property
    def linear_2(self) -> Linear:
        return self.l2


def encoder_mask_bias(mask: torch.Tensor, batch: int, skv: int) -> torch.Tensor:
    """``encoder_attention_mask`` of the reference forward -> the attention kernel's additive key bias.
    unet_2d_condition.py:1071-1073 / transformer_2d.py:395-398: a [batch][key_tokens] mask (1 = keep, 0 = discard) becomes
    ``(1 - mask) * -10000`` in the sample dtype, broadcast over heads and queries; a 3-D [batch][1][key_tokens] tensor is
    taken as that bias already.  Returns bf16 [batch][1][1][ceil64(skv)] (one row for every query)."""
    if mask.dim() == 2:
        bias = (1 - mask.to(bf16)) * -10000.0
    elif mask.dim() == 3 and mask.shape[1] == 1:
        bias = mask[:, 0].to(bf16)
    else:
        raise ValueError("encoder_attention_mask must be [batch][key_tokens] or a [batch][1][key_tokens] bias")
    if bias.shape[0] != batch or bias.shape[1] != skv:
        raise ValueError(f"encoder_attention_mask shape {tuple(mask.shape)} does not match ({batch}, {skv}) text tokens")
    out = torch.zeros((batch, 1, 1, (skv + 63) // 64 * 64), device=mask.device, dtype=bf16)
    out[:, 0, 0, :skv] = bias
    return out


def pad_encoder_states(ehs: torch.Tensor):
    """[B][S][C] -> ([B*S_alloc][C] zero padded, S, S_alloc) with S_alloc a multiple of 16 (16-byte aligned V^T rows)."""
    B, S, C = ehs.shape
    s_alloc = ((S + 15) // 16) * 16
    if s_alloc == S:
        return ehs.reshape(B * S, C).contiguous(), S, s_alloc
    pad = torch.zeros((B, s_alloc, C), device=ehs.device, dtype=ehs.dtype)
    pad[:, :S].copy_(ehs)
    return pad.view(B * s_alloc, C), S, s_alloc
