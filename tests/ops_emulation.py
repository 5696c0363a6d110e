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


ACT_NONE, ACT_GEGLU, ACT_GELU_TANH, ACT_SILU, ACT_GELU_ERF, ACT_QUICK_GELU, ACT_GEGLU_TANH = 0, 1, 2, 3, 4, 5, 6


def _act(y, act):
    if act == ACT_NONE:
        return y
    y = y.to(bf16).float()                       # the kernels activate the bf16-rounded projection
    if act == ACT_SILU:
        return F.silu(y)
    if act == ACT_GELU_TANH:
        return F.gelu(y, approximate="tanh")
    if act == ACT_GELU_ERF:
        return F.gelu(y)
    if act == ACT_QUICK_GELU:
        return y * torch.sigmoid((1.702 * y).to(bf16).float()).to(bf16).float()
    raise AssertionError(f"activation {act}")


def conv2d_nhwc(x, w, bias=None, *, ksize=3, x2=None, stride=1, up=False, pad=None, rowvec=None, residual=None,
                out_scale=1.0, act=0, tile=None, staging=None, pad_after=0, out=None, split_k=None, k_valid=0):
    assert x.is_contiguous() and x.shape[-1] % 64 == 0 and w.shape[0] % 4 == 0, "C ABI: conv channels % 64, N % 4"
    if k_valid:   # the contract of da_gemm_params.k_valid: everything past it is zero padding in both operands
        assert x2 is None and 0 < k_valid <= x.shape[-1]
        assert not x[..., k_valid:].any() and not w.view(w.shape[0], ksize * ksize, -1)[..., k_valid:].any()
    if x2 is not None:
        assert x2.is_contiguous() and x2.shape[-1] % 64 == 0 and x2.shape[:3] == x.shape[:3]
        x = torch.cat([x, x2], -1)
    B, H, W_, C = x.shape
    co = w.shape[0]
    wt = w.float().view(co, ksize, ksize, C).permute(0, 3, 1, 2)
    xi = x.float().permute(0, 3, 1, 2)
    if up:
        xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
    pd = (ksize - 1) // 2 if pad is None else pad
    xi = F.pad(xi, (pd, pd + pad_after, pd, pd + pad_after))
    y = F.conv2d(xi, wt, None if bias is None else bias.float(), stride=stride).permute(0, 2, 3, 1)
    if rowvec is not None:
        y = y + rowvec.float()[:, None, None, :]
    y = _act(y, act)
    if residual is not None:
        assert tuple(residual.shape) == tuple(y.shape) and residual.is_contiguous()
        y = y + residual.float()
    if out is not None:
        assert tuple(out.shape) == tuple(y.shape) and out.is_contiguous()
    return _store(y * out_scale, out)


def linear(x, w, bias=None, *, act=0, residual=None, rowvec=None, rows_per_batch=0, alpha=1.0, out_scale=1.0, out=None,
           out_f32=False, bias_rows=None, gate=None, tile=None, staging=None, split_k=None, stats_out=None, ln=None,
           k_valid=0, vt_out=None, xattn=None):
    if xattn is not None:    # da_gemm_params.xa_*: to_q (bf16) -> softmax(scale q k^T) v over the text keys, heads of 64 channels
        assert act == 0 and residual is None and gate is None and rowvec is None and stats_out is None and vt_out is None and not out_f32
        q = linear(x, w, bias, ln=ln, alpha=alpha)
        M, N = q.shape
        seq, skv, sa = xattn["seq"], xattn["skv"], xattn["skv_alloc"]
        assert N % 128 == 0 and seq % 128 == 0 and M % seq == 0 and sa <= 80 and sa % 8 == 0
        return attention(q, xattn["k"], xattn["vt"], B=M // seq, H=N // 64, D=64, Sq=seq, Skv=skv, Skv_alloc=sa, q_row_stride=N,
                         k_row_stride=xattn["k"].stride(0), q_batch_stride=seq * N, k_batch_stride=sa * xattn["k"].stride(0),
                         vt_ld=xattn["vt"].stride(0), vt_batch_stride=sa, scale=xattn["scale"], out=out)
    assert x.stride(1) == 1 and w.stride(1) == 1 and x.shape[1] % 64 == 0 and w.shape[0] % 4 == 0, "C ABI: K % 64, N % 4"
    if k_valid:
        assert 0 < k_valid <= x.shape[1] and not x[:, k_valid:].any() and not w[:, k_valid:].any()
    assert x.stride(0) % 8 == 0 and w.stride(0) % 8 == 0, "C ABI: row strides % 8"
    y = alpha * (x.float() @ w.float().t())
    if ln is not None:      # LayerNorm fold, consumer: rstd * (x W'^T - mu s) + c from the producer's partial sums
        rs, fold = ln
        assert rs.parts > 0 and rs.buf.shape[0] == x.shape[0]
        tot = rs.buf[:, :rs.parts].sum(dim=1)
        mean = tot[:, 0] / x.shape[1]
        rstd = torch.rsqrt((tot[:, 1] / x.shape[1] - mean * mean).clamp_min(0) + fold.eps)
        y = rstd[:, None] * (y - mean[:, None] * fold.s[None, :]) + fold.c[None, :]
    if bias is not None:
        y = y + bias.float()
    if bias_rows is not None:
        y = y + bias_rows.float()[:, None]
    M = y.shape[0]
    bidx = None
    if rowvec is not None or gate is not None:
        assert rows_per_batch > 0
        bidx = torch.arange(M) // rows_per_batch
    if rowvec is not None:
        y = y + rowvec.float()[bidx]
    if act in (ACT_GEGLU, ACT_GEGLU_TANH):        # packed rows: per 64 = [32 value | 32 gate] (ops.pack_geglu)
        assert w.shape[0] % 128 == 0 and residual is None and gate is None and rowvec is None and not out_f32
        g = y.to(bf16).float().view(M, -1, 2, 32)
        gl = F.gelu(g[:, :, 1]) if act == ACT_GEGLU else F.gelu(g[:, :, 1], approximate="tanh")
        y = (g[:, :, 0] * gl.to(bf16).float()).reshape(M, -1)
    else:
        y = _act(y, act)
    if gate is not None:
        gv = gate.float()[bidx]
        y = y.to(bf16).float() * gv
        if gate.dtype == bf16:
            y = y.to(bf16).float()
    if residual is not None:
        y = y + residual.float()
    if vt_out is not None:      # transposed column block (da_gemm_params.vt): columns >= col0 -> vt[n - col0][m], the rest -> C
        vt, col0 = vt_out
        assert act == ACT_NONE and residual is None and gate is None and rowvec is None and stats_out is None and col0 % 16 == 0
        assert vt.shape[0] == y.shape[1] - col0 and vt.shape[1] >= M and vt.dtype == bf16
        vt[:, :M] = y[:, col0:].t().to(bf16)
        y = y[:, :col0]
    res = _store(y * out_scale, out, torch.float32 if out_f32 else bf16)
    if stats_out is not None:   # LayerNorm fold, producer: partial (sum, sum of squares) of the STORED values; the stand-in
        assert act != ACT_GEGLU and not out_f32 and stats_out.buf.shape[0] == M     # writes two parts (column halves)
        r = res.float()
        h = r.shape[1] // 2
        stats_out.parts = 2
        for q, blk in enumerate((r[:, :h], r[:, h:])):
            stats_out.buf[:, q, 0] = blk.sum(dim=1)
            stats_out.buf[:, q, 1] = (blk * blk).sum(dim=1)
    return res


def linear_pair(a, b):
    """da_gemm_pair_bf16: two independent problems, results identical to two linear() calls."""
    return linear(**a), linear(**b)


def linear_qkv(x, wqkv, col0, bias=None, ln=None):
    vt = torch.empty((wqkv.shape[0] - col0, x.shape[0]), dtype=bf16)
    return linear(x, wqkv, bias, ln=ln, vt_out=(vt, col0)), vt


def linear_small_m(x, w, bias=None, *, act_in=0, act_out=0, residual=None, out=None):
    assert x.shape[0] <= 8 and w.is_contiguous()
    xi = x.float()
    if act_in == ACT_SILU:
        xi = F.silu(xi).to(bf16).float()
    else:
        assert act_in == ACT_NONE
    y = xi @ w.float().t()
    if bias is not None:
        y = y + bias.float()
    y = _act(y, act_out)
    if residual is not None:
        y = y + residual.float()
    return _store(y, out)


def attention(q, k, vt, *, B, H, D, Sq, Skv, Skv_alloc, q_row_stride, k_row_stride, q_batch_stride, k_batch_stride,
              vt_ld, vt_batch_stride, scale=None, out=None, ring_slots=0, causal=False, bias=None, q_block=0, pv_delay=0, algo=0):
    assert D in (64, 96, 128, 160) and Skv_alloc % 8 == 0 and Skv_alloc >= Skv
    for s_ in (q_row_stride, k_row_stride, q_batch_stride, k_batch_stride, vt_ld, vt_batch_stride):
        assert s_ % 8 == 0, "C ABI: attention strides % 8"
    qq = q.as_strided((B, H, Sq, D), (q_batch_stride, D, q_row_stride, 1)).float()
    kk = k.as_strided((B, H, Skv, D), (k_batch_stride, D, k_row_stride, 1)).float()
    vv = vt.as_strided((B, H, D, Skv), (vt_batch_stride, D * vt_ld, vt_ld, 1)).float()
    sc = qq @ kk.transpose(2, 3) * (D ** -0.5 if scale is None else scale)
    if bias is not None or causal:
        assert D == 64, "C ABI: the masked attention variant exists for D = 64"
    if bias is not None:
        assert bias.dim() == 4 and bias.stride(3) == 1 and bias.shape[3] >= ((Skv + 63) // 64) * 64 and bias.stride(2) % 4 == 0
        bb = bias[:, :, :Sq, :Skv].float()
        sc = torch.where(bb <= -1e29, torch.full_like(sc, float("-inf")), sc + bb)
    if causal:
        keep = torch.arange(Skv)[None, :] <= torch.arange(Sq)[:, None]
        sc = sc.masked_fill(~keep, float("-inf"))
    p = torch.softmax(sc, dim=-1)
    o = (p @ vv.transpose(2, 3)).permute(0, 2, 1, 3).reshape(B * Sq, H * D)
    return _store(o, out)


def rms_norm(x, gamma, eps):
    """da_rmsnorm_bf16: T5LayerNorm with its two roundings."""
    xf = x.float()
    return (gamma.float() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(bf16).float()).to(bf16)


def group_norm_nhwc(x, gamma, beta, groups, eps, silu=False, x2=None):
    if x2 is not None:
        x = torch.cat([x, x2], -1)
    shp = x.shape
    B, C = shp[0], shp[-1]
    y = F.group_norm(x.float().reshape(B, -1, C).transpose(1, 2), groups, gamma.float(), beta.float(), eps).transpose(1, 2)
    y = y.to(bf16)
    if silu:
        y = F.silu(y.float()).to(bf16)
    return y.reshape(shp).contiguous()


def layer_norm(x, gamma, beta, eps, *, mod_scale=None, mod_shift=None, rows_per_batch=0):
    M, C = x.shape
    y = F.layer_norm(x.float(), (C,), None if gamma is None else gamma.float(), None if beta is None else beta.float(), eps)
    if mod_scale is not None:
        bidx = torch.arange(M) // rows_per_batch
        if mod_scale.dtype == bf16:
            y = y.to(bf16).float()
        y = y * (1 + mod_scale.float()[bidx]) + mod_shift.float()[bidx]
    return y.to(bf16)


def rmsnorm_rope_(x, *, heads, head_dim, col_offsets, weights=None, eps=1e-6, cos=None, sin=None, rope_row0=0,
                  rows_per_batch=0, norm="per_head"):
    rows = x.shape[0]
    rpb = rows_per_batch or rows
    C = heads * head_dim
    for j, off in enumerate(col_offsets):
        blk = x[:, off:off + C].float().view(rows, heads, head_dim)
        w = weights[j] if weights is not None else None
        if norm == "per_head":
            blk = blk * torch.rsqrt(blk.pow(2).mean(-1, keepdim=True) + eps)
            if w is not None:
                blk = blk * w.float().view(1, 1, head_dim)
            blk = blk.to(bf16).float()
        elif norm == "across_heads":
            blk = blk * torch.rsqrt(blk.pow(2).mean((-2, -1), keepdim=True) + eps)
            if w is not None:
                blk = blk * w.float().view(1, heads, head_dim)
            blk = blk.to(bf16).float()
        else:
            assert norm == "none"
        if cos is not None:
            r = rope_row0 + (torch.arange(rows) % rpb)
            c, s_ = cos[r][:, None, :], sin[r][:, None, :]
            ev, od = blk[..., 0::2], blk[..., 1::2]
            o = torch.empty_like(blk)
            o[..., 0::2] = ev * c[..., 0::2] - od * s_[..., 0::2]
            o[..., 1::2] = od * c[..., 1::2] + ev * s_[..., 1::2]
            blk = o
        x[:, off:off + C] = blk.reshape(rows, C).to(bf16)
    return x


def timestep_embedding(t, dim, *, batch, flip_sin_to_cos, shift, scale=1.0, max_period=10000.0, table=None, step_idx=None,
                       out_f32=False):
    import math
    if t is None:
        t = table.view(-1, 8)[int(step_idx), 7].reshape(1)
    t = t.float().reshape(-1).expand(batch) if t.numel() == 1 else t.float().reshape(-1)
    half = dim // 2
    e = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - shift))
    a = scale * t[:, None] * e[None, :]
    emb = torch.cat([torch.sin(a), torch.cos(a)], -1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], -1)
    return emb if out_f32 else emb.to(bf16)


def conv_thin_out(x, w, bias, *, out_f32=False, postprocess=None):
    B, H, W_, C = x.shape
    co = w.shape[0]
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().view(co, 3, 3, C).permute(0, 3, 1, 2),
                 None if bias is None else bias.float(), padding=1)
    y = y.contiguous() if out_f32 else y.to(bf16).contiguous()
    return y if postprocess is None else image_postprocess(y, postprocess)


def bcast_add_f32(a, m):
    return a.float()[None, :] + m.float()


def patchify3d(x, patch):
    B, C, Fr, H, W_ = x.shape
    pt, ph, pw = patch
    y = x.view(B, C, Fr // pt, pt, H // ph, ph, W_ // pw, pw).permute(0, 2, 4, 6, 1, 3, 5, 7)
    return y.reshape(-1, C * pt * ph * pw).contiguous()


def unpatchify3d(tok, shape, patch):
    B, C, Fr, H, W_ = shape
    pt, ph, pw = patch
    y = tok.view(B, Fr // pt, H // ph, W_ // pw, pt, ph, pw, C).permute(0, 7, 1, 4, 2, 5, 3, 6)
    return y.reshape(B, C, Fr, H, W_).contiguous()


def transpose(x, out=None):
    return _store(x.t().float(), out)


def mul_scalar(x, s, rep=1):
    return torch.cat([(x.float() * s).to(x.dtype)] * rep, 0)


# ---- sampler kernels: table-driven fused scheduler steps (rounding follows the tensor dtype like the kernels; the exact
# rounding order is what the GPU tests pin bit for bit, here a faithful-to-tolerance version is enough) ----
def _row(table, step_idx, width=8):
    return table.view(-1, width)[int(step_idx)].float()


def _cfg(eps, cfg, g, n):
    e = eps.reshape(-1)
    if not cfg:
        return e.float()
    u, c = e[:n], e[n:]
    return (u + (g * (c - u)).to(e.dtype)).float() if e.dtype == bf16 else u + g * (c - u)


def euler_scale_model_input(x, table, step_idx, rep=1):
    y = (x.float() / _row(table, step_idx)[3]).to(x.dtype)
    return torch.cat([y] * rep, 0)


def euler_step(eps, x, table, step_idx, *, cfg, guidance, out=None, pred_type=0):
    r = _row(table, step_idx)
    e = _cfg(eps, cfg, guidance, x.numel()).view(x.shape)
    xf = x.float()
    if pred_type == 0:
        pred = xf - (r[0] * e).to(eps.dtype).float()
    elif pred_type == 1:
        pred = (e * r[4]).to(eps.dtype).float() + xf / r[5]
    else:
        pred = e
    prev = xf + ((xf - pred) / r[0]) * r[2]
    return _store(prev, out, x.dtype)


def x0_linear_step(eps, x, noise, table, step_idx, *, cfg, guidance, out=None, noise_step_stride=0, pred_type=0):
    r = _row(table, step_idx)
    e = _cfg(eps, cfg, guidance, x.numel()).view(x.shape)
    xf = x.float()
    if pred_type == 0:
        x0, pe = (xf - r[0] * e) / r[1], e
    elif pred_type == 1:
        x0, pe = r[1] * xf - r[0] * e, r[1] * e + r[0] * xf
    else:
        x0, pe = e, (xf - r[1] * e) / r[0]
    if r[6] > 0:
        x0 = x0.clamp(-r[6], r[6])
    prev = r[2] * x0 + r[3] * pe + r[4] * xf
    if noise is not None and float(r[5]) != 0.0:
        off = int(step_idx) * noise_step_stride
        nz = noise.reshape(-1)[off:off + x.numel()].view(x.shape).float()
        prev = prev + r[5] * nz
    return _store(prev, out, x.dtype)


def flowmatch_step(v, x, table, step_idx, *, cfg=False, guidance=0.0, out=None):
    r = _row(table, step_idx)
    vv = _cfg(v, cfg, guidance, x.numel()).view(x.shape)
    return _store(x.float() + (r[2] * vv).to(v.dtype).float(), out, v.dtype)   # result in the model output's dtype


def unipc_flow_step_(v, x, last, m1, m2, coef, step_idx, *, cfg=False, guidance=0.0):
    r = _row(coef, step_idx, 16)
    vv = _cfg(v, cfg, guidance, x.numel()).view(x.shape)
    xf, l_, a1, a2 = x.float(), last.float(), m1.float(), m2.float()
    mn = xf - (r[0] * vv).to(v.dtype).float()
    xc = xf
    if r[1] != 0:
        inner = r[8] * (mn - a1)
        if int(r[2]) == 2:
            inner = r[7] * ((a2 - a1) / r[6]) + inner
        xc = (r[3] * l_ - r[4] * a1) - r[5] * inner
    xn = r[10] * xc - r[11] * mn
    if int(r[9]) == 2:
        xn = xn - r[12] * (0.5 * ((a1 - mn) / r[13]))
    m2.copy_(m1)
    m1.copy_(mn.to(x.dtype))
    last.copy_(xc.to(x.dtype))
    x.copy_(xn.to(x.dtype))
    return x


def image_postprocess(img, output_type):
    mode = {"pt": 0, "np": 1, "uint8": 2}.get(output_type)
    if mode is None:
        raise ValueError(f"image_postprocess: output_type {output_type!r} (use 'pt', 'np' or 'uint8')")
    v = (img.float() * 0.5 + 0.5).clamp(0, 1)
    if mode == 0:
        return v
    v = v.permute(0, *range(2, img.dim()), 1).contiguous()
    return v if mode == 1 else torch.from_numpy((v.numpy() * 255).round().astype("uint8"))


def advance_step(step_idx):
    step_idx += 1


def require_hip(t, name, dtypes=(bf16,)):
    if t.dtype not in dtypes:
        raise ValueError(f"{name}: dtype {t.dtype}")


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
    assert w.shape[0] % 8 == 0
    xi = x.float() if in_nchw else x.float().permute(0, 3, 1, 2)
    assert xi.shape[1] <= 16
    if in_div != 1.0:
        xi = (x.to(bf16) / in_div).float() if in_nchw else (x.to(bf16) / in_div).float().permute(0, 3, 1, 2)
    if in_add != 0.0:
        xi = (xi.to(bf16) + in_add).float()
    co, cin = w.shape[0], xi.shape[1]
    wt = w.float().view(co, ksize, ksize, cin).permute(0, 3, 1, 2) if ksize == 3 else w.float().view(co, cin, 1, 1)
    y = F.conv2d(xi, wt, None if bias is None else bias.float(), padding=(ksize - 1) // 2)
    return y.permute(0, 2, 3, 1).to(bf16).contiguous()


def cast_f32_bf16(x, rep=1):
    return torch.cat([x.to(bf16)] * rep, 0)


def cfg_rescale(eps2b, guidance, guidance_rescale, out=None):
    """da_cfg_rescale: the reference's op chain on the tensor dtype (pipeline_stable_diffusion.py:69-92, :1054-1059)."""
    u, c = eps2b.chunk(2)
    cfg = u + guidance * (c - u)
    dims = list(range(1, c.ndim))
    resc = cfg * (c.std(dim=dims, keepdim=True) / cfg.std(dim=dims, keepdim=True))
    y = guidance_rescale * resc + (1 - guidance_rescale) * cfg
    if out is not None:
        out.copy_(y)
        return out
    return y


def install(monkeypatch, ops_module):
    """Replace the kernels behind ``ops_module`` with the stand-ins above (pack_* helpers are pure torch and stay)."""
    for name in ("conv2d_nhwc", "linear", "linear_pair", "linear_qkv", "linear_small_m", "rms_norm", "attention", "softmax_rows", "group_norm_nhwc", "layer_norm",
                 "rmsnorm_rope_", "rmsnorm_channels", "timestep_embedding", "permute_0213", "frames_to_ncthw",
                 "conv_thin_in", "conv_thin_out", "bcast_add_f32", "patchify3d", "unpatchify3d", "transpose",
                 "mul_scalar", "cast_f32_bf16", "cfg_rescale", "require_hip", "euler_scale_model_input", "euler_step", "x0_linear_step",
                 "flowmatch_step", "unipc_flow_step_", "advance_step", "image_postprocess"):
        monkeypatch.setattr(ops_module, name, globals()[name])
