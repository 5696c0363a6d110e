"""TEST INFRASTRUCTURE ONLY: torch-CPU stand-ins for the ``diffusers_amd.ops`` entry points a host-side model class
calls, each with the C ABI's contract (same arguments, layouts, in-place / aliasing behaviour, bf16 rounding at the
output).  They let the -m "not gpu" suite run a model's HOST logic (weight packing, zero padding, frame / tap
scheduling, bias folding) against the reference fixtures; the kernels themselves are only ever tested on the GPU."""
from __future__ import annotations

import torch
import torch.nn.functional as F

bf16 = torch.bfloat16


def _store(val, out, dtype=bf16):
    val = val.to(dtype)
    if out is None:
        return val.contiguous()
    out.copy_(val.reshape(out.shape))
    return out


def conv2d_nhwc(x, w, bias=None, *, ksize=3, x2=None, stride=1, up=False, pad=None, rowvec=None, residual=None,
                out_scale=1.0, act=0, tile=None, staging=None, pad_after=0, out=None):
    assert x2 is None and rowvec is None and act == 0 and pad_after == 0 and x.is_contiguous()
    assert x.shape[-1] % 64 == 0 and w.shape[0] % 4 == 0, "C ABI: conv channels % 64, N % 4"
    B, H, W_, C = x.shape
    co = w.shape[0]
    wt = w.float().view(co, ksize, ksize, C).permute(0, 3, 1, 2)
    xi = x.float().permute(0, 3, 1, 2)
    if up:
        xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
    y = F.conv2d(xi, wt, None if bias is None else bias.float(), stride=stride, padding=(ksize - 1) // 2 if pad is None else pad)
    y = y.permute(0, 2, 3, 1)
    if residual is not None:
        assert tuple(residual.shape) == tuple(y.shape) and residual.is_contiguous()
        y = y + residual.float()
    if out is not None:
        assert tuple(out.shape) == tuple(y.shape) and out.is_contiguous()
    return _store(y * out_scale, out)


def linear(x, w, bias=None, *, act=0, residual=None, rowvec=None, rows_per_batch=0, alpha=1.0, out_scale=1.0, out=None,
           out_f32=False, bias_rows=None, gate=None, tile=None, staging=None):
    assert act == 0 and rowvec is None and bias_rows is None and gate is None
    assert x.stride(1) == 1 and w.stride(1) == 1 and x.shape[1] % 64 == 0 and w.shape[0] % 4 == 0, "C ABI: K % 64, N % 4"
    y = alpha * (x.float() @ w.float().t())
    if bias is not None:
        y = y + bias.float()
    if residual is not None:
        y = y + residual.float()
    return _store(y * out_scale, out, torch.float32 if out_f32 else bf16)


def softmax_rows(scores, out=None):
    M, N = scores.shape
    p = torch.softmax(scores.float(), dim=-1).to(bf16)
    if out is None:
        return p
    out[:, :N] = p
    return out


def rmsnorm_channels(x, gamma, *, real_channels, silu=False):
    assert x.is_contiguous() and x.shape[-1] % 8 == 0
    f = x.float()
    n = (f / f.norm(dim=-1, keepdim=True).clamp_min(1e-12)).to(bf16)
    n = (n.float() * (float(real_channels) ** 0.5)).to(bf16)
    n = (n.float() * gamma.float()).to(bf16)
    return F.silu(n.float()).to(bf16) if silu else n


def permute_0213(x, out=None):
    assert x.dim() == 4 and x.is_contiguous() and x.shape[-1] % 8 == 0
    y = x.permute(0, 2, 1, 3).contiguous()
    if out is None:
        return y
    assert out.is_contiguous() and out.numel() == y.numel()
    out.view(-1).copy_(y.view(-1))
    return out


def frames_to_ncthw(x, *, batch, channels, lo=-1.0, hi=1.0, out_f32=False):
    BT, H, W_, Cs = x.shape
    T = BT // batch
    y = x.float()[..., :channels].clamp(lo, hi).view(batch, T, H, W_, channels).permute(0, 4, 1, 2, 3)
    return y.contiguous().to(torch.float32 if out_f32 else bf16)


def conv_thin_in(x, w, bias, *, ksize, in_nchw, in_div=1.0, in_add=0.0):
    assert ksize == 1 and in_nchw and in_div == 1.0 and in_add == 0.0 and x.shape[1] <= 16 and w.shape[0] % 8 == 0
    y = torch.einsum("bchw,oc->bhwo", x.float(), w.float())
    if bias is not None:
        y = y + bias.float()
    return y.to(bf16).contiguous()


def cast_f32_bf16(x, rep=1):
    return torch.cat([x.to(bf16)] * rep, 0)


def install(monkeypatch, ops_module):
    """Replace the kernels behind ``ops_module`` with the stand-ins above (pack_* helpers are pure torch and stay)."""
    for name in ("conv2d_nhwc", "linear", "softmax_rows", "rmsnorm_channels", "permute_0213", "frames_to_ncthw",
                 "conv_thin_in", "cast_f32_bf16"):
        monkeypatch.setattr(ops_module, name, globals()[name])
