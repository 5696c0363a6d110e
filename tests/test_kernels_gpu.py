"""GPU parity tests of every HIP kernel, called through the C ABI (diffusers_amd.ops -> ctypes -> libdiffusers_amd.so)
and checked against plain PyTorch fp32 references of the same op / the oracle restatement / the golden vectors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close_bf16

pytestmark = pytest.mark.gpu

bf16 = torch.bfloat16
DEV = "cuda"


def _ops():
    from diffusers_amd import _lib as L
    from diffusers_amd import ops
    return ops, L


def rnd(shape, seed, scale=1.0, dtype=bf16):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(DEV)


# ----------------------------------------------------------------------------------------------------------------------
# GEMM
# ----------------------------------------------------------------------------------------------------------------------
GEMM_SHAPES = [(256, 256, 256), (154, 1280, 2048), (2048, 1280, 1280), (1000, 324, 320), (128, 64, 64), (8192, 640, 640),
               (96, 132, 192)]


@pytest.mark.parametrize("staging", [0, 1])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_plain(M, N, K, tile, staging):
    ops, L = _ops()
    x = rnd((M, K), 1)
    w = rnd((N, K), 2, scale=K ** -0.5)
    y = ops.linear(x, w, tile=tile, staging=staging)
    ref = x.float() @ w.float().t()
    assert y.shape == (M, N)
    assert_close_bf16(y, ref, f"gemm {M}x{N}x{K} tile={tile} stage={staging}", rtol=8e-3, atol_rms=4e-3)


# every (tile, staging) variant the C ABI exposes: tiles 1..8 x staging 0 (register) / 1..5 (LDS-DMA ring of 2/3/4/6/8
# slots); combinations whose ring does not fit the 160 KiB LDS return DA_ERR_UNSUPPORTED.  28 are valid (25 of round 1 + the
# 8-wave 128x128 tile with 2 / 3 / 4 slots).
ALL_VARIANTS = [(1, 0)] + [(t, s) for s in range(0, 6) for t in range(1, 9) if (t, s) != (1, 0)]
N_VALID_VARIANTS = 28


def _run_variants(fn, what, geglu=False):
    """fn(tile, staging) -> tensor; the first variant is checked by the caller, the others must equal it bit for bit."""
    base, n_ok = None, 0
    for tile, staging in ALL_VARIANTS:
        try:
            y = fn(tile, staging)
        except RuntimeError as e:
            assert "DA_ERR_UNSUPPORTED" in str(e), f"{what} (tile={tile}, staging={staging}): {e}"
            continue
        n_ok += 1
        if base is None:
            base = y
        else:
            assert torch.equal(y, base), f"{what}: variant (tile={tile}, staging={staging}) differs from (1, 0)"
    assert n_ok == (14 if geglu else N_VALID_VARIANTS)  # GEGLU needs an even number of 32-col tiles per wave, f"{what}: {n_ok} variants ran"
    return base


@pytest.mark.parametrize("M,N,K", [(300, 320, 192), (2048, 1280, 1280), (520, 132, 64), (777, 640, 1152)])
def test_gemm_all_variants_bit_identical(M, N, K):
    """All variants walk K in the same order with the same MFMA: outputs must be BIT-identical (and right)."""
    ops, L = _ops()
    x = rnd((M, K), 21)
    w = rnd((N, K), 22, scale=K ** -0.5)
    bias, res = rnd((N,), 23), rnd((M, N), 24)
    ref = x.float() @ w.float().t() + bias.float() + res.float()
    y = _run_variants(lambda t, st: ops.linear(x, w, bias, residual=res, tile=t, staging=st), f"gemm {M}x{N}x{K}")
    assert_close_bf16(y, ref, f"gemm {M}x{N}x{K} all variants", rtol=8e-3, atol_rms=4e-3)


@pytest.mark.parametrize("kind,Cp,Cv", [("linear", 128, 96), ("linear", 64, 32), ("linear", 192, 160), ("conv", 128, 96),
                                        ("conv", 64, 32), ("conv", 128, 80)])
def test_k_valid_skips_the_channel_padding(kind, Cp, Cv):
    """da_gemm_params.k_valid: with zero padding past k_valid in both operands every variant must return what k_valid = 0
    returns.  Where a whole half-slice is padding (valid % 64 == 32) the skipping instantiations really skip the MFMA
    steps over it: junk planted there no longer reaches the output."""
    ops, L = _ops()
    if kind == "linear":
        M, N = 700, 192
        x, w = torch.zeros((M, Cp), device=DEV, dtype=bf16), torch.zeros((N, Cp), device=DEV, dtype=bf16)
        x[:, :Cv], w[:, :Cv] = rnd((M, Cv), 61), rnd((N, Cv), 62, Cv ** -0.5)
        b, r = rnd((N,), 63), rnd((M, N), 64)
        run = lambda t, st, kv, xx=x, ww=w: ops.linear(xx, ww, b, residual=r, tile=t, staging=st, k_valid=kv)  # noqa: E731
        ref = x.float() @ w.float().t() + b.float() + r.float()
    else:
        B, H, W_, N = 2, 20, 24, 96
        x = torch.zeros((B, H, W_, Cp), device=DEV, dtype=bf16)
        x[..., :Cv] = rnd((B, H, W_, Cv), 61)
        w4 = torch.zeros((N, Cp, 3, 3), device=DEV, dtype=bf16)
        w4[:, :Cv] = rnd((N, Cv, 3, 3), 62, (9 * Cv) ** -0.5)
        w, b = ops.pack_conv_weight(w4), rnd((N,), 63)
        run = lambda t, st, kv, xx=x, ww=w: ops.conv2d_nhwc(xx, ww, b, ksize=3, tile=t, staging=st, k_valid=kv)  # noqa: E731
        ref = _conv_ref(x, w4, b)
    full = run(1, 1, 0)
    y = _run_variants(lambda t, st: run(t, st, Cv), f"{kind} C{Cp} k_valid {Cv}")
    assert torch.equal(y, full)
    assert_close_bf16(y, ref, f"{kind} C{Cp} k_valid {Cv}", rtol=8e-3, atol_rms=4e-3)
    if Cv % 64 == 32:   # the padding is exactly the upper half of the last slice of every period
        xj, wj = x.clone(), w.clone()
        if kind == "linear":
            xj[:, Cv:], wj[:, Cv:] = 7.0, 5.0
        else:
            xj[..., Cv:] = 7.0
            wj.view(N, 9, Cp)[..., Cv:] = 5.0
        # the instantiations that skip (gemm_kernel.cuh dispatch, DA_KS): tile / ring depth pairs of the large video convs
        for t, st in [(1, 1), (3, 1), (2, 1), (5, 2), (6, 2), (7, 1), (8, 1), (8, 2)]:
            assert torch.equal(run(t, st, Cv, xj, wj), full), f"tile {t} staging {st}: padding still multiplied"
    with pytest.raises((RuntimeError, ValueError)):
        run(1, 1, Cp + 8)


def test_gemm_pair_launch_is_bit_identical_to_two_launches(monkeypatch):
    """da_gemm_pair_bf16: the fused Q|K projection and the swapped V^T projection of a self-attention layer in ONE launch
    (blocks [0, grid_a) on problem a, the rest on problem b), for every tile / ring depth both problems can run:
    bit-identical to the two separate launches, including ragged shapes and epilogues."""
    ops, L = _ops()
    import ctypes as C
    lib = L.load()
    for (M, K, Nq) in [(2048, 1280, 2560), (520, 192, 260), (8192, 640, 1280)]:
        x = rnd((M, K), 21)
        wqk = rnd((Nq, K), 22, K ** -0.5)
        wv = rnd((Nq // 2, K), 23, K ** -0.5)
        bias = rnd((Nq,), 24)
        ref_qk = ops.linear(x, wqk, bias, tile=L.TILE_128x128, staging=L.STAGE_LDS_DIRECT)
        ref_vt = ops.linear(wv, x, tile=L.TILE_128x128, staging=L.STAGE_LDS_DIRECT)
        n_ok = 0
        for tile in range(1, 9):
            for st in range(1, 6):
                pa, s_ = ops._linear_params(x, wqk, bias, tile=tile, staging=st)
                pb, _ = ops._linear_params(wv, x, tile=tile, staging=st)
                rc = lib.da_gemm_pair_bf16(C.byref(pa), C.byref(pb), s_)
                if rc == 3:        # DA_ERR_UNSUPPORTED (tile, staging) combination
                    continue
                assert rc == 0, (tile, st, rc)
                assert torch.equal(pa._out, ref_qk) and torch.equal(pb._out, ref_vt), (M, K, Nq, tile, st)
                n_ok += 1
        assert n_ok >= 15
    # the tuned front end (table or live tuning decides paired vs separate): same bits either way -- within the first kernel
    # family (a separate launch that the tuner gives to the K2 family differs in the fp32 summation order)
    from diffusers_amd import tuning
    monkeypatch.setattr(tuning, "FAMILY", "1")
    monkeypatch.setattr(tuning, "_table", {})
    monkeypatch.setattr(tuning, "_loaded", True)
    qk, vt = ops.linear_pair({"x": x, "w": wqk, "bias": bias}, {"x": wv, "w": x})
    assert torch.equal(qk, ref_qk) and torch.equal(vt, ref_vt)


@pytest.mark.parametrize("M,N,K", [(2048, 1280, 1280), (2048, 1280, 5120), (300, 320, 1024), (1000, 260, 512)])
def test_gemm_split_k_in_launch_reduction(M, N, K):
    """split_k = 2..4 (in-launch reduction through the workspace, agent-scope flag hand-off): deterministic (every run
    and every tile variant of one split factor gives the same bits), equal to the unsplit result up to the fp32
    summation order (checked against the fp32 reference with the unsplit kernel's own tolerance AND elementwise within
    1 bf16 ulp of the unsplit kernel), epilogue (bias + residual) applied once by the reducer, flags re-armed (the same
    problem runs many times through one workspace), the give-up word never set."""
    ops, L = _ops()
    x, w, b, r = rnd((M, K), 31), rnd((N, K), 32, K ** -0.5), rnd((N,), 33), rnd((M, N), 34)
    base = ops.linear(x, w, b, residual=r, tile=L.TILE_128x128, staging=L.STAGE_LDS_DIRECT)
    ref = x.float() @ w.float().t() + b.float() + r.float()
    ran = 0
    for split in (2, 3, 4):
        first = None
        for tile, st in ((L.TILE_128x128, L.STAGE_LDS_DIRECT), (L.TILE_128x128, L.STAGE_LDS_DIRECT3),
                         (L.TILE_256x128, L.STAGE_LDS_DIRECT3), (L.TILE_128x256, L.STAGE_LDS_DIRECT),
                         (L.TILE_128x64, L.STAGE_LDS_DIRECT3), (L.TILE_64x128, L.STAGE_LDS_DIRECT3)):
            if (K // 64) < split:
                continue
            for rep in range(3):
                try:
                    y = ops.linear(x, w, b, residual=r, tile=tile, staging=st, split_k=split)
                except RuntimeError as e:      # more blocks than can be co-resident for this tile: refused, not run
                    assert "UNSUPPORTED" in str(e)
                    y = None
                    break
                if first is None:
                    first = y.clone()
                assert torch.equal(y, first), f"split {split} tile {tile}/{st} rep {rep}: not deterministic"
            if y is not None:
                ran += 1
        if first is None:
            continue
        assert_close_bf16(first, ref, f"split-K {split} {M}x{N}x{K}")
        ulp = (first.float() - base.float()).abs() / base.float().abs().clamp_min(2.0 ** -10)
        frac = float((first != base).float().mean())
        print(f"[parity] split-K {split} {M}x{N}x{K}: {100 * frac:.2f}% of outputs differ from split 1 by <= {float(ulp.max()):.2e} relative")
        assert float(ulp.max()) <= 2.0 ** -7, "a split-K output is more than one bf16 ulp from the unsplit kernel"
    assert ran >= 3
    assert not ops.splitk_error()


@pytest.mark.parametrize("B,H,W,C1,C2,Co,stride,up", [(2, 32, 32, 1280, 0, 1280, 1, False), (2, 16, 16, 640, 320, 640, 1, False),
                                                     (1, 16, 16, 320, 0, 320, 2, False), (1, 8, 8, 256, 0, 128, 1, True)])
def test_conv_split_k(B, H, W, C1, C2, Co, stride, up):
    """split-K on the implicit GEMM: a block starts in the middle of the (tap, channel) walk (cursor initialised from its
    first K slice), for plain, two-source (skip concat), strided and upsampling convs, with bias + time-embedding row
    vector + residual applied once by the reducer.  Deterministic, and within one bf16 ulp of the unsplit kernel."""
    ops, L = _ops()
    x1 = rnd((B, H, W, C1), 41)
    x2 = rnd((B, H, W, C2), 42) if C2 else None
    K = 9 * (C1 + C2)
    w, b = rnd((Co, K), 43, K ** -0.5), rnd((Co,), 44)
    Ho, Wo = ((2 * H if up else H) // stride, (2 * W if up else W) // stride)
    rv, res = rnd((B, Co), 45), rnd((B, Ho, Wo, Co), 46)
    kw = dict(ksize=3, x2=x2, stride=stride, up=up, rowvec=rv, residual=res)
    base = ops.conv2d_nhwc(x1, w, b, tile=L.TILE_128x128, staging=L.STAGE_LDS_DIRECT, **kw)
    ran = 0
    for split in (2, 3, 4):
        first = None
        for tile, st in ((L.TILE_128x128, L.STAGE_LDS_DIRECT), (L.TILE_128x128, L.STAGE_LDS_DIRECT3),
                         (L.TILE_256x128, L.STAGE_LDS_DIRECT3), (L.TILE_128x256, L.STAGE_LDS_DIRECT3),
                         (L.TILE_128x64, L.STAGE_LDS_DIRECT3), (L.TILE_64x128, L.STAGE_LDS_DIRECT3)):
            for rep in range(2):
                try:
                    y = ops.conv2d_nhwc(x1, w, b, tile=tile, staging=st, split_k=split, **kw)
                except RuntimeError as e:
                    assert "UNSUPPORTED" in str(e)
                    y = None
                    break
                if first is None:
                    first = y.clone()
                assert torch.equal(y, first), f"conv split {split} tile {tile}/{st} rep {rep}: not deterministic"
            ran += y is not None
        if first is not None:
            ulp = (first.float() - base.float()).abs() / base.float().abs().clamp_min(2.0 ** -8)
            assert float(ulp.max()) <= 2.0 ** -7, f"conv split {split}: more than one bf16 ulp from the unsplit kernel"
    assert ran >= 4 and not ops.splitk_error()


@pytest.mark.parametrize("B,H,W,C1,C2,Co", [(2, 8, 8, 1280, 0, 1280), (2, 8, 8, 1280, 1280, 1280), (2, 16, 16, 1280, 0, 1280), (1, 8, 8, 512, 512, 512)])
def test_conv_split_k_wide_factors(B, H, W, C1, C2, Co):
    """The split factors the shipped table uses on the 8 x 8 / 16 x 16 levels of the SD1.5 / DDPM U-Nets, up to DA_SPLITK_MAX (round 6:
    8 -> 24; M = 64 .. 512 rows under K = 4.6 k .. 23 k): every factor the library accepts is deterministic across repeats and tiles,
    within one bf16 ulp of the unsplit kernel and within the kernel tolerance of the fp32 reference; a factor past the maximum is
    refused as INVALID; a launch that could not be co-resident is refused, never run."""
    ops, L = _ops()
    x1 = rnd((B, H, W, C1), 51)
    x2 = rnd((B, H, W, C2), 52) if C2 else None
    K = 9 * (C1 + C2)
    w, b = rnd((Co, K), 53, K ** -0.5), rnd((Co,), 54)
    rv, res = rnd((B, Co), 55), rnd((B, H, W, Co), 56)
    kw = dict(ksize=3, x2=x2, rowvec=rv, residual=res)
    base = ops.conv2d_nhwc(x1, w, b, tile=L.TILE_128x128, staging=L.STAGE_LDS_DIRECT, **kw)
    xin = x1 if x2 is None else torch.cat([x1, x2], -1)
    wf = w.float().view(Co, 3, 3, C1 + C2).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xin.float().permute(0, 3, 1, 2), wf, b.float(), padding=1).permute(0, 2, 3, 1) \
        + rv.float()[:, None, None, :] + res.float()
    ran = 0
    for split in (6, 8, 12, 16, 24):
        first = None
        for tile, st in ((L.TILE_128x64, L.STAGE_LDS_DIRECT3), (L.TILE_64x128, L.STAGE_LDS_DIRECT3), (L.TILE_128x128, L.STAGE_LDS_DIRECT)):
            y = None
            for rep in range(3):
                try:
                    y = ops.conv2d_nhwc(x1, w, b, tile=tile, staging=st, split_k=split, **kw)
                except RuntimeError as e:
                    assert "UNSUPPORTED" in str(e) or "INVALID" in str(e)
                    y = None
                    break
                if first is None:
                    first = y.clone()
                assert torch.equal(y, first), f"conv split {split} tile {tile}/{st} rep {rep}: not deterministic"
            ran += y is not None
        if first is not None:
            assert_close_bf16(first, ref, f"conv split-K {split} M{B * H * W} K{K}")
            ulp = (first.float() - base.float()).abs() / base.float().abs().clamp_min(2.0 ** -8)
            assert float(ulp.max()) <= 2.0 ** -7, f"conv split {split}: more than one bf16 ulp from the unsplit kernel"
    assert ran >= 5 and not ops.splitk_error()
    with pytest.raises(RuntimeError):
        ops.conv2d_nhwc(x1, w, b, tile=L.TILE_128x64, staging=L.STAGE_LDS_DIRECT3, split_k=L.SPLITK_MAX + 1, **kw)


@pytest.mark.parametrize("M,C,N", [(2048, 1280, 1280), (520, 320, 384), (8192, 640, 640)])
def test_layernorm_fold_producer_and_consumers(M, C, N):
    """LayerNorm folded into the GEMMs either side of it (da_gemm_params.stats_out / ln_*):
      * PRODUCER: every tile variant writes per-row partial (sum, sum of squares) of its bf16 output; their sum equals the
        row statistics of the stored tensor (fp32 tolerance), the output itself is bit-identical to the launch without
        statistics, and da_gemm_stats_parts() is the number of partials written;
      * CONSUMER: linear(x, W', ln=...) == linear(layer_norm(x), W) within the bf16 tolerance of one GEMM (the fold skips
        the bf16 rounding of the normalised tensor and rounds gamma o W instead), for a plain epilogue (+bias) and for the
        GEGLU epilogue, for every tile variant; rows with a large mean (|mu| = 8 sigma) included."""
    ops, L = _ops()
    a, wprod, res = rnd((M, 192), 61), rnd((C, 192), 62, 192 ** -0.5), rnd((M, C), 63)
    res = res + 8.0 * (torch.arange(M, device=DEV) % 3 == 0).to(bf16)[:, None]       # a third of the rows: mean ~ 8 sigma
    gamma, beta = rnd((C,), 64) * 0.3 + 1.0, rnd((C,), 65) * 0.2
    w, b = rnd((N, C), 66, C ** -0.5), rnd((N,), 67)
    w1 = rnd((2 * max(N // 128, 1) * 128, C), 68, C ** -0.5)
    b1 = rnd((w1.shape[0],), 69)
    x_plain = ops.linear(a, wprod, residual=res, tile=L.TILE_128x128, staging=1)
    xf = x_plain.float()
    ln_ref = F.layer_norm(xf, (C,), gamma.float(), beta.float(), 1e-5)
    ref = ln_ref @ w.float().t() + b.float()
    g = ln_ref @ w1.float().t() + b1.float()
    h_, g_ = g.chunk(2, dim=-1)
    ref_geglu = h_ * F.gelu(g_)
    wl, fold = ops.fold_layernorm(w, gamma, beta, 1e-5)
    w1l, fold1 = ops.fold_layernorm(w1, gamma, beta, 1e-5)
    w1p, b1p = ops.pack_geglu(w1l, b1)
    n2 = w1.shape[0] // 2
    idx = torch.arange(n2, device=DEV).view(n2 // 32, 32)
    order = torch.cat([idx, idx + n2], dim=1).reshape(-1)
    fold1p = ops.LNFold(fold1.s[order].contiguous(), fold1.c[order].contiguous(), fold1.eps)
    n_prod = 0
    for tile in range(1, 9):
        st = ops.RowStats(M, DEV)
        st.buf.fill_(float("nan"))
        try:
            x = ops.linear(a, wprod, residual=res, tile=tile, staging=1, stats_out=st)
        except RuntimeError:
            continue
        n_prod += 1
        assert torch.equal(x, x_plain), f"tile {tile}: statistics changed the output"
        tot = st.buf[:, :st.parts].sum(dim=1)
        assert torch.isfinite(tot).all() and torch.isnan(st.buf[:, st.parts:]).all(), f"tile {tile}: wrong number of partials"
        assert torch.allclose(tot[:, 0], xf.sum(dim=1), rtol=1e-5, atol=1e-2)
        assert torch.allclose(tot[:, 1], (xf * xf).sum(dim=1), rtol=1e-5, atol=1e-2)
        for ctile in range(1, 9):
            try:
                y = ops.linear(x, wl, b, tile=ctile, staging=1, ln=(st, fold))
            except RuntimeError:
                continue
            assert_close_bf16(y, ref, f"LN fold, producer tile {tile} -> consumer tile {ctile}", rel_rms_max=6e-3)
            if ctile in (L.TILE_128x128, L.TILE_256x128, L.TILE_64x128) and tile == 1:
                yg = ops.linear(x, w1p, b1p, act=L.ACT_GEGLU, tile=ctile, staging=1, ln=(st, fold1p))
                # value * gelu(gate): two bf16-rounded factors, each carrying the rounding of (gamma o W) -- the elementwise
                # bound is 2.5 % (1 element in 2.6 M sat at 1.8 % with the 1.6 % default), the rms bound stays tight
                assert_close_bf16(yg, ref_geglu, f"LN fold + GEGLU, consumer tile {ctile}", rtol=2.5e-2, atol_rms=2.5e-2,
                                  rel_rms_max=8e-3)
    assert n_prod >= 6


def test_conv_all_variants_bit_identical():
    ops, L = _ops()
    B, H, W, C1, C2, Cout = 2, 24, 20, 64, 128, 192
    x1, x2 = rnd((B, H, W, C1), 31), rnd((B, H, W, C2), 32)
    w = rnd((Cout, C1 + C2, 3, 3), 33, scale=(9 * (C1 + C2)) ** -0.5)
    b = rnd((Cout,), 34)
    wp = ops.pack_conv_weight(w)
    xcat = torch.cat([x1, x2], -1).permute(0, 3, 1, 2).float()
    for stride, up in ((1, False), (2, False), (1, True)):
        xin = F.interpolate(xcat, scale_factor=2.0, mode="nearest") if up else xcat
        ref = F.conv2d(xin, w.float(), b.float(), stride=stride, padding=1).permute(0, 2, 3, 1)
        y = _run_variants(lambda t, st: ops.conv2d_nhwc(x1, wp, b, ksize=3, x2=x2, stride=stride, up=up, tile=t,
                                                        staging=st), f"conv s{stride} up{up}")
        assert_close_bf16(y, ref, f"conv s{stride} up{up} all variants", rtol=8e-3, atol_rms=4e-3)


def test_geglu_all_variants_bit_identical():
    ops, L = _ops()
    M, Cc = 700, 256
    x = rnd((M, Cc), 41)
    w = rnd((8 * Cc, Cc), 42, scale=Cc ** -0.5)
    b = rnd((8 * Cc,), 43, scale=0.1)
    wp, bp = ops.pack_geglu(w, b)
    h = (x.float() @ w.float().t() + b.float()).to(bf16).float()
    hv, gate = h.chunk(2, -1)
    y = _run_variants(lambda t, st: ops.linear(x, wp, bp, act=L.ACT_GEGLU, tile=t, staging=st), "geglu", geglu=True)
    assert_close_bf16(y, hv * F.gelu(gate), "geglu all variants", rtol=1.6e-2, atol_rms=8e-3)


def test_gemm_tuner_picks_a_valid_variant(tmp_path, monkeypatch):
    """da_gemm_tune through diffusers_amd.tuning: the chosen variant is recorded, reused, and changes no bit."""
    ops, L = _ops()
    from diffusers_amd import tuning
    monkeypatch.setattr(tuning, "_table", {})
    monkeypatch.setattr(tuning, "_loaded", True)
    monkeypatch.setattr(tuning, "FAMILY", "1")     # first family only: its variants are bit-identical to the reference launch
    x, w = rnd((2048, 640), 51), rnd((1280, 640), 52, scale=640 ** -0.5)
    y_ref = ops.linear(x, w, tile=L.TILE_128x128, staging=L.STAGE_REGISTER)
    y = ops.linear(x, w)  # tunes live
    assert torch.equal(y, y_ref)
    (key, (tile, staging, us, split)), = tuning.table().items()
    assert 1 <= tile <= 8 and 1 <= staging <= 5 and us > 0 and split == 1   # split-K is opt-in (DIFFUSERS_AMD_SPLITK=1)
    print(f"[tune] {key} -> tile {L.TILE_NAMES[tile]} staging {staging}: {us:.1f} us")
    assert torch.equal(ops.linear(x, w), y_ref)  # table hit
    out = tuning.save(tmp_path / "t.json")
    assert "lin:M2048:N1280:K640" in out.read_text()
    # a paired launch is tuned as ONE problem and stored under its own key; same bits as two launches
    wv = rnd((640, 640), 53, scale=640 ** -0.5)
    qk, vt = ops.linear_pair({"x": x, "w": w}, {"x": wv, "w": x})
    assert torch.equal(qk, y_ref) and torch.equal(vt, ops.linear(wv, x, tile=L.TILE_128x128, staging=L.STAGE_REGISTER))
    assert any(k.startswith("pair:") for k in tuning.table())
    # both families competing (the default): the K2 family may win; same result up to the fp32 summation order of K
    monkeypatch.setattr(tuning, "_table", {})
    monkeypatch.setattr(tuning, "FAMILY", "all")
    y2 = ops.linear(x, w)
    (key, (tile, staging, us, split)), = tuning.table().items()
    print(f"[tune] both families: {key} -> tile {L.TILE_NAMES[tile]} staging {staging}: {us:.1f} us")
    assert 1 <= tile < len(L.TILE_NAMES) and torch.equal(ops.linear(x, w), y2)
    scale = torch.maximum(y_ref.float().abs(), y_ref.float().pow(2).mean().sqrt())
    assert float(((y2.float() - y_ref.float()).abs() / scale).max()) <= 2.0 ** -7
    monkeypatch.setattr(tuning, "FAMILY", "k2")
    monkeypatch.setattr(tuning, "_table", {})
    ops.linear(x, w)
    assert next(iter(tuning.table().values()))[0] >= L.FIRST_K2_TILE


@pytest.mark.parametrize("staging", [0, 1])
def test_gemm_epilogues(staging):
    ops, L = _ops()
    M, N, K = 520, 384, 448
    x, w = rnd((M, K), 3), rnd((N, K), 4, scale=K ** -0.5)
    bias, res = rnd((N,), 5), rnd((M, N), 6)
    rowvec = rnd((4, N), 7)
    y = ops.linear(x, w, bias, residual=res, rowvec=rowvec, rows_per_batch=130, out_scale=0.5, staging=staging)
    ref = x.float() @ w.float().t() + bias.float() + rowvec.float().repeat_interleave(130, 0) + res.float()
    assert_close_bf16(y, ref * 0.5, "gemm bias+rowvec+residual+scale", rtol=8e-3, atol_rms=4e-3)
    # fp32 output with alpha (attention scores)
    s = ops.linear(x, w, alpha=0.125, out_f32=True, staging=staging)
    assert s.dtype == torch.float32
    assert_close_bf16(s, 0.125 * (x.float() @ w.float().t()), "gemm f32 out + alpha", rtol=1e-4, atol_rms=1e-4)
    # activations
    for act, fn in ((L.ACT_SILU, F.silu), (L.ACT_GELU_TANH, lambda t: F.gelu(t, approximate="tanh")),
                    (L.ACT_GELU_ERF, F.gelu)):
        y = ops.linear(x, w, bias, act=act, staging=staging)
        assert_close_bf16(y, fn(x.float() @ w.float().t() + bias.float()), f"gemm act {act}", rtol=1.6e-2, atol_rms=8e-3)


@pytest.mark.parametrize("staging", [0, 1])
@pytest.mark.parametrize("tile", [0, 1, 2])
def test_gemm_geglu(tile, staging):
    ops, L = _ops()
    M, C = 300, 128
    x = rnd((M, C), 8)
    w = rnd((8 * C, C), 9, scale=C ** -0.5)
    b = rnd((8 * C,), 10, scale=0.1)
    wp, bp = ops.pack_geglu(w, b)
    y = ops.linear(x, wp, bp, act=L.ACT_GEGLU, tile=tile, staging=staging)
    h = x.float() @ w.float().t() + b.float()
    hv, gate = h.chunk(2, dim=-1)
    assert y.shape == (M, 4 * C)
    assert_close_bf16(y, hv * F.gelu(gate), f"geglu tile={tile}", rtol=1.6e-2, atol_rms=8e-3)


def test_gemm_strided_views_and_transposed_v():
    """Exactly the call patterns of layers.Attention: fused QK output consumed through column views, V^T = W_v . X^T."""
    ops, L = _ops()
    M, C = 512, 128
    x = rnd((M, C), 11)
    wv = rnd((C, C), 12, scale=C ** -0.5)
    vt = ops.linear(wv, x)  # [C][M]
    assert_close_bf16(vt, wv.float() @ x.float().t(), "V^T gemm", rtol=8e-3, atol_rms=4e-3)
    big = rnd((M, 2 * C), 13)
    y = ops.linear(big[:, C:], wv)  # row stride 2C, offset C
    assert_close_bf16(y, big[:, C:].float() @ wv.float().t(), "gemm on strided view", rtol=8e-3, atol_rms=4e-3)


# ----------------------------------------------------------------------------------------------------------------------
# implicit-GEMM conv
# ----------------------------------------------------------------------------------------------------------------------
def _conv_ref(x_nhwc, w_oihw, bias, stride=1, up=False, pad=1):
    x = x_nhwc.float().permute(0, 3, 1, 2)
    if up:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    y = F.conv2d(x, w_oihw.float(), bias.float() if bias is not None else None, stride=stride, padding=pad)
    return y.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("staging", [0, 1])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,up", [
    (2, 16, 16, 64, 128, 1, False), (1, 13, 9, 128, 64, 1, False), (2, 16, 16, 64, 64, 2, False),
    (2, 8, 8, 128, 128, 1, True), (1, 32, 32, 320, 320, 1, False), (1, 17, 17, 64, 68, 2, False)])
def test_conv3x3(B, H, W, Cin, Cout, stride, up, staging):
    ops, L = _ops()
    x = rnd((B, H, W, Cin), 20)
    w = rnd((Cout, Cin, 3, 3), 21, scale=(9 * Cin) ** -0.5)
    b = rnd((Cout,), 22, scale=0.1)
    for tile in (0, 1, 4):
        y = ops.conv2d_nhwc(x, ops.pack_conv_weight(w), b, ksize=3, stride=stride, up=up, tile=tile, staging=staging)
        ref = _conv_ref(x, w, b, stride, up)
        assert y.shape == ref.shape
        assert_close_bf16(y, ref, f"conv3x3 {B}x{H}x{W}x{Cin}->{Cout} s{stride} up{up} tile{tile}", rtol=8e-3, atol_rms=4e-3)


@pytest.mark.parametrize("staging", [0, 1])
def test_conv_concat_rowvec_residual(staging):
    ops, L = _ops()
    B, H, W, C1, C2, Cout = 2, 12, 12, 128, 64, 128
    x1, x2 = rnd((B, H, W, C1), 23), rnd((B, H, W, C2), 24)
    w = rnd((Cout, C1 + C2, 3, 3), 25, scale=(9 * (C1 + C2)) ** -0.5)
    b, tv = rnd((Cout,), 26, scale=0.1), rnd((B, Cout), 27)
    res = rnd((B, H, W, Cout), 28)
    y = ops.conv2d_nhwc(x1, ops.pack_conv_weight(w), b, ksize=3, x2=x2, rowvec=tv, residual=res, out_scale=0.5,
                        staging=staging)
    ref = _conv_ref(torch.cat([x1, x2], -1), w, b) + tv.float()[:, None, None, :] + res.float()
    assert_close_bf16(y, ref * 0.5, "conv3x3 concat+temb+residual", rtol=8e-3, atol_rms=4e-3)
    # 1x1 shortcut over the same two sources
    w1 = rnd((Cout, C1 + C2, 1, 1), 29, scale=(C1 + C2) ** -0.5)
    y1 = ops.conv2d_nhwc(x1, w1.reshape(Cout, -1).contiguous(), b, ksize=1, x2=x2, staging=staging)
    ref1 = _conv_ref(torch.cat([x1, x2], -1), w1, b, pad=0)
    assert_close_bf16(y1, ref1, "conv1x1 concat", rtol=8e-3, atol_rms=4e-3)


# ----------------------------------------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, H):
    from oracle.reference_math import attention
    return attention(q.float().cpu(), k.float().cpu(), v.float().cpu(), H)


@pytest.mark.parametrize("B,H,D,Sq,Skv", [(2, 3, 64, 256, 256), (1, 2, 64, 200, 200), (2, 2, 64, 128, 77),
                                         (1, 5, 64, 1024, 1024), (1, 2, 128, 192, 320), (2, 1, 128, 64, 77)])
def test_flash_attention(B, H, D, Sq, Skv):
    ops, L = _ops()
    C = H * D
    q, k, v = rnd((B, Sq, C), 30), rnd((B, Skv, C), 31), rnd((B, Skv, C), 32)
    skv_alloc = ((Skv + 15) // 16) * 16
    kp = torch.zeros((B, skv_alloc, C), device=DEV, dtype=bf16)
    kp[:, :Skv] = k
    vt = torch.zeros((C, B * skv_alloc), device=DEV, dtype=bf16)
    vt.view(C, B, skv_alloc)[:, :, :Skv] = v.permute(2, 0, 1)
    o = ops.attention(q.view(B * Sq, C), kp.view(B * skv_alloc, C), vt, B=B, H=H, D=D, Sq=Sq, Skv=Skv,
                      Skv_alloc=skv_alloc, q_row_stride=C, k_row_stride=C, q_batch_stride=Sq * C,
                      k_batch_stride=skv_alloc * C, vt_ld=B * skv_alloc, vt_batch_stride=skv_alloc)
    ref = _attn_ref(q, k, v, H).view(B * Sq, C)
    assert_close_bf16(o, ref, f"flash attn B{B} H{H} D{D} Sq{Sq} Skv{Skv}", rtol=1.6e-2, atol_rms=1.6e-2)


@pytest.mark.parametrize("B,H,Sq,Skv", [(2, 3, 256, 256), (1, 2, 200, 333), (1, 5, 1024, 1024), (2, 2, 130, 77)])
def test_flash_attention_query_block_sizes_are_bit_identical(B, H, Sq, Skv):
    """da_attention_params.q_block: 64-query (two-wave) and 128-query (four-wave) workgroups walk the same tiles in the same
    order per wave, so the choice -- made from the grid size when q_block = 0 -- must not change a single bit."""
    ops, L = _ops()
    D, C = 64, H * 64
    q, k, v = rnd((B, Sq, C), 36), rnd((B, Skv, C), 37), rnd((B, Skv, C), 38)
    sa = ((Skv + 15) // 16) * 16
    kp = torch.zeros((B, sa, C), device=DEV, dtype=bf16)
    kp[:, :Skv] = k
    vt = torch.zeros((C, B * sa), device=DEV, dtype=bf16)
    vt.view(C, B, sa)[:, :, :Skv] = v.permute(2, 0, 1)
    def run(qb, **kw):
        return ops.attention(q.view(B * Sq, C), kp.view(B * sa, C), vt, B=B, H=H, D=D, Sq=Sq, Skv=Skv, Skv_alloc=sa,
                             q_row_stride=C, k_row_stride=C, q_batch_stride=Sq * C, k_batch_stride=sa * C,
                             vt_ld=B * sa, vt_batch_stride=sa, q_block=qb, **kw)
    # first generation (algo = 1; q_block = 64 exists only there and pins it)
    o128, o64, oauto = run(128, algo=1), run(64), run(0, algo=1)
    assert torch.equal(o128, o64) and torch.equal(oauto, o128)
    ref = _attn_ref(q, k, v, H).view(B * Sq, C)
    assert_close_bf16(o64, ref, f"flash attn q_block=64 B{B} H{H} Sq{Sq} Skv{Skv}", rtol=1.6e-2, atol_rms=1.6e-2)
    # second generation (the default): 128- and 256-query workgroups, 3- and 4-tile rings -- every wave walks the same tiles in the
    # same order with the same arithmetic
    n128, n256, nauto = run(128, algo=2), run(256, algo=2), run(0)
    assert torch.equal(n128, n256) and torch.equal(nauto, n128)
    assert torch.equal(run(128, algo=2, ring_slots=4), n128) and torch.equal(run(256, algo=2, ring_slots=4), n128)
    assert_close_bf16(n256, ref, f"flash attn v2 q_block=256 B{B} H{H} Sq{Sq} Skv{Skv}", rtol=1.6e-2, atol_rms=1.6e-2)
    # ... and with the softmax shift folded into the first product (algo = 3: Q pre-scaled, one more bf16 rounding)
    a128, a256 = run(128, algo=3), run(256, algo=3)
    assert torch.equal(a128, a256)
    assert_close_bf16(a128, ref, f"flash attn v2 aug B{B} H{H} Sq{Sq} Skv{Skv}", rtol=1.6e-2, atol_rms=1.6e-2)
    # ... with the row sum taken by the matrix pipe (algo = 4: a ones row next to V^T), and with both (algo = 5)
    for algo in (4, 5):
        r128, r256 = run(128, algo=algo), run(256, algo=algo)
        assert torch.equal(r128, r256)
        assert_close_bf16(r128, ref, f"flash attn v2 algo {algo} B{B} H{H} Sq{Sq} Skv{Skv}", rtol=1.6e-2, atol_rms=1.6e-2)
    with pytest.raises(RuntimeError):
        run(32)
    with pytest.raises(RuntimeError):
        run(256, algo=1)


@pytest.mark.parametrize("B,H,D,Sq,Skv", [(2, 3, 64, 256, 256), (1, 2, 64, 200, 333), (1, 5, 64, 1024, 1024), (2, 2, 64, 130, 77),
                                         (1, 2, 64, 96, 64), (1, 2, 128, 192, 320), (1, 2, 128, 520, 1030), (2, 1, 128, 64, 129)])
def test_flash_attention_pv_delay_is_bit_identical(B, H, D, Sq, Skv):
    """da_attention_params.pv_delay: the loops that issue tile j - 1's P.V product (1) and also tile j + 1's Q.K^T product (2) under
    tile j's softmax perform the same operations in the same order per accumulator -- not one bit may differ from the plain loop
    (and it must be right)."""
    ops, L = _ops()
    C = H * D
    q, k, v = rnd((B, Sq, C), 36), rnd((B, Skv, C), 37), rnd((B, Skv, C), 38)
    k[0, Skv // 2] = q[0, 5] * 3.0          # a late jump of the running maximum: the rescale branch runs mid-sequence
    sa = ((Skv + 15) // 16) * 16
    kp = torch.zeros((B, sa, C), device=DEV, dtype=bf16)
    kp[:, :Skv] = k
    vt = torch.zeros((C, B * sa), device=DEV, dtype=bf16)
    vt.view(C, B, sa)[:, :, :Skv] = v.permute(2, 0, 1)
    def run(pd):
        return ops.attention(q.view(B * Sq, C), kp.view(B * sa, C), vt, B=B, H=H, D=D, Sq=Sq, Skv=Skv, Skv_alloc=sa,
                             q_row_stride=C, k_row_stride=C, q_batch_stride=Sq * C, k_batch_stride=sa * C,
                             vt_ld=B * sa, vt_batch_stride=sa, pv_delay=pd)
    plain, delayed = run(-1), run(1)                            # (a pinned pv_delay selects the first-generation kernel)
    assert torch.equal(plain, delayed)
    # pv_delay = 2: the next tile's Q.K^T rides in the softmax slices as well (4-slot ring, the last MFMA of a score tile writes
    # the registers the softmax reads next) -- same operations, same order, same bits
    assert torch.equal(run(2), plain)
    assert_close_bf16(delayed, _attn_ref(q, k, v, H).view(B * Sq, C), f"flash attn pv_delay B{B} H{H} D{D} Sq{Sq} Skv{Skv}",
                      rtol=1.6e-2, atol_rms=1.6e-2)


def test_flash_attention_rescale_branch():
    """Force the online-softmax running max to jump late: one key matches one query far more than the others."""
    ops, L = _ops()
    B, H, D, S = 1, 1, 64, 512
    q, k, v = rnd((B, S, D), 33), rnd((B, S, D), 34), rnd((B, S, D), 35)
    k[0, 400] = q[0, 17] * 4.0
    k[0, 3] = q[0, 100] * 3.0
    vt = v.permute(2, 0, 1).reshape(D, B * S).contiguous()
    ref = _attn_ref(q, k, v, H).view(S, D)
    for algo in (0, 1, 2, 3, 4, 5):   # default, first generation, second generation (deferred maximum), + shift in Q.K^T, + row sum by MFMA, + both
        o = ops.attention(q.view(S, D), k.view(S, D), vt, B=B, H=H, D=D, Sq=S, Skv=S, Skv_alloc=S, q_row_stride=D,
                          k_row_stride=D, q_batch_stride=S * D, k_batch_stride=S * D, vt_ld=S, vt_batch_stride=S, algo=algo)
        assert_close_bf16(o, ref, f"flash attn spiked keys algo {algo}", rtol=1.6e-2, atol_rms=1.6e-2)


@pytest.mark.parametrize("D,S,Skv,lo", [(64, 300, 1000, -40.0), (128, 200, 700, -60.0), (64, 128, 77, 25.0), (128, 96, 320, 0.0)])
def test_flash_attention_v2_deferred_maximum(D, S, Skv, lo):
    """Second-generation kernel: the running shift is only raised when a row outgrows it by more than 2^8.  Scores with a large
    common offset (very negative: the first tile must LOWER the initial shift of the AUG variant; very positive: it must raise
    it at once), a slow upward drift (many small raises that stay under the threshold until they add up) and late spikes."""
    ops, L = _ops()
    B, H = 1, 2
    C = H * D
    q, k, v = rnd((B, S, C), 51), rnd((B, Skv, C), 52), rnd((B, Skv, C), 53)
    k = k.float()
    qn = q.float() / q.float().norm(dim=-1, keepdim=True)
    # every score of query i gets the offset lo * (1 + drift): add a multiple of q_0's direction ... simpler: a shared direction u
    u = torch.nn.functional.normalize(torch.randn(C, device=DEV), dim=0)
    qf = q.float() + 6.0 * u                                      # all queries share a strong component along u
    drift = torch.linspace(0.0, 1.0, Skv, device=DEV)[None, :, None]
    k = k + (lo / 6.0 + 4.0 * drift) * u * (D ** 0.5) / H        # keys walk along u: scores = lo + slow upward drift
    k[0, Skv - 5] += qn[0, 7] * 30.0                              # late spikes
    k[0, Skv // 3] += qn[0, S - 1] * 20.0
    q, k = qf.to(bf16), k.to(bf16)
    sa = ((Skv + 15) // 16) * 16
    kp = torch.zeros((B, sa, C), device=DEV, dtype=bf16)
    kp[:, :Skv] = k
    vt = torch.zeros((C, B * sa), device=DEV, dtype=bf16)
    vt.view(C, B, sa)[:, :, :Skv] = v.permute(2, 0, 1)
    ref = _attn_ref(q, k, v, H).view(B * S, C)
    for algo in (2, 3, 4, 5):
        o = ops.attention(q.view(B * S, C), kp.view(B * sa, C), vt, B=B, H=H, D=D, Sq=S, Skv=Skv, Skv_alloc=sa,
                          q_row_stride=C, k_row_stride=C, q_batch_stride=S * C, k_batch_stride=sa * C, vt_ld=B * sa,
                          vt_batch_stride=sa, algo=algo)
        assert torch.isfinite(o.float()).all()
        # algo 3 / 5 round Q * scale * log2(e) to bf16 once more: with a large COMMON score offset that relative error becomes an
        # absolute one on every score (|s| 2^-9), which the shift-invariance of the softmax does not remove -- the reason the
        # variant is opt-in.  Its bound here is 1.5x the default's.
        tol = 3e-2 if algo in (3, 5) else 2e-2
        assert_close_bf16(o, ref, f"flash attn v2 deferred max D{D} S{S} Skv{Skv} offset {lo} algo {algo}", rtol=tol, atol_rms=tol)


# ----------------------------------------------------------------------------------------------------------------------
# norms
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,HW,C,silu", [(2, 256, 64, True), (2, 1024, 320, True), (1, 4096, 128, False),
                                        (2, 64, 1280, True), (1, 100, 960, True), (2, 16, 2560, False)])
def test_groupnorm(B, HW, C, silu):
    ops, L = _ops()
    x = rnd((B, HW, C), 40, scale=2.0) + 0.5
    g, b = rnd((C,), 41) * 0.1 + 1.0, rnd((C,), 42, scale=0.1)
    y = ops.group_norm_nhwc(x, g, b, 32, 1e-5, silu=silu)
    ref = F.group_norm(x.float().transpose(1, 2), 32, g.float(), b.float(), 1e-5)
    if silu:
        ref = F.silu(ref)
    assert_close_bf16(y, ref.transpose(1, 2), f"groupnorm B{B} HW{HW} C{C} silu{silu}", rtol=1.6e-2, atol_rms=8e-3)


def test_groupnorm_two_sources():
    ops, L = _ops()
    B, HW, C1, C2 = 2, 144, 640, 320
    x1, x2 = rnd((B, HW, C1), 43), rnd((B, HW, C2), 44, scale=3.0)
    g, b = rnd((C1 + C2,), 45) * 0.1 + 1.0, rnd((C1 + C2,), 46, scale=0.1)
    y = ops.group_norm_nhwc(x1, g, b, 32, 1e-5, silu=True, x2=x2)
    ref = F.silu(F.group_norm(torch.cat([x1, x2], -1).float().transpose(1, 2), 32, g.float(), b.float(), 1e-5))
    assert_close_bf16(y, ref.transpose(1, 2), "groupnorm fused concat", rtol=1.6e-2, atol_rms=8e-3)


@pytest.mark.parametrize("B,HW,C1,C2,G", [(2, 1024, 1280, 0, 32), (2, 1024, 1280, 1280, 32), (2, 1024, 1280, 640, 32), (2, 1024, 640, 0, 32),
                                          (2, 4096, 320, 0, 32), (2, 4096, 640, 0, 32), (2, 4096, 640, 320, 32), (2, 64, 1280, 0, 32),
                                          (2, 256, 2560, 0, 32), (1, 1000, 128, 0, 32), (3, 77, 256, 0, 32), (2, 4096, 512, 0, 32),
                                          (1, 100, 960, 0, 32), (2, 1024, 96, 0, 4)])
def test_groupnorm_one_launch_form_against_the_two_kernel_form(B, HW, C1, C2, G, monkeypatch):
    """Round 5: tensors whose per-(batch, group set) slab is small run GroupNorm as ONE launch (gn_fused_kernel: slab parked in LDS,
    or re-read from L2 up to DA_GN_FUSED_KB): against torch in fp32 with the tolerance of the two-kernel form, against that form
    (DA_GN_FUSED=0) to a bf16 ulp of the normalised value -- the statistics are summed in another order --, deterministic; group
    sets of 1 / 2 / 4 groups (cpg 40, 80 / 20, 60 / 10, 30), two sources, large means."""
    ops, L = _ops()
    C = C1 + C2
    x1 = rnd((B, HW, C1), 40, scale=2.0) + 3.0
    x2 = rnd((B, HW, C2), 44, scale=3.0) if C2 else None
    g, b = rnd((C,), 41) * 0.1 + 1.0, rnd((C,), 42, scale=0.1)
    full = torch.cat([x1, x2], -1) if C2 else x1
    ref = F.silu(F.group_norm(full.float().transpose(1, 2), G, g.float(), b.float(), 1e-5)).transpose(1, 2)
    monkeypatch.setenv("DA_GN_FUSED", "0")
    two = ops.group_norm_nhwc(x1, g, b, G, 1e-5, silu=True, x2=x2)
    monkeypatch.setenv("DA_GN_MULTI", "0")          # (the several-workgroup form of round 6 has its own test below)
    for kb in ("0", "4096"):                       # default reach (LDS-resident slabs); everything the one-launch form can take
        monkeypatch.setenv("DA_GN_FUSED", "1")
        monkeypatch.setenv("DA_GN_FUSED_KB", kb)
        monkeypatch.setenv("DA_GN_FUSED_MINWG", "1")
        one = ops.group_norm_nhwc(x1, g, b, G, 1e-5, silu=True, x2=x2)
        what = f"groupnorm one launch B{B} HW{HW} C{C1}+{C2} G{G} reach {kb} KiB"
        assert_close_bf16(one, ref, what, rtol=1.6e-2, atol_rms=8e-3)
        d = (one.float() - two.float()).abs()
        tol = 2.0 ** -6 * two.float().abs() + 1e-3          # (one bf16 ulp of the binade above |value|)
        assert int((d > tol).sum()) == 0, f"{what}: differs from the two-kernel form by more than a bf16 ulp ({float(d.max()):.3e})"
        assert torch.equal(one, ops.group_norm_nhwc(x1, g, b, G, 1e-5, silu=True, x2=x2)), "deterministic"


@pytest.mark.parametrize("B,HW,C1,C2,G", [(2, 4096, 320, 0, 32), (2, 4096, 640, 0, 32), (2, 16384, 320, 0, 32), (2, 4096, 1280, 640, 32),
                                          (2, 1024, 1280, 1280, 32), (2, 1024, 1280, 640, 32), (2, 4096, 640, 320, 32), (2, 4096, 640, 640, 32),
                                          (2, 4001, 640, 0, 32), (1, 5000, 512, 0, 32), (3, 3000, 256, 0, 8)])
def test_groupnorm_one_launch_over_several_workgroups(B, HW, C1, C2, G, monkeypatch):
    """Round 6 (the statistics pass of the mid-size tensors): slabs that exceed one CU's LDS are dealt to 2 .. 32
    workgroups that park their pixels in LDS, exchange fp64 partial sums through the stream's sync buffer and wait for each other --
    ONE launch, the tensor read once.  Against torch in fp32 with the two-kernel form's tolerance; against the two-kernel form to a bf16
    ulp (another summation order); deterministic over repeats that share the arrival counters with launches of OTHER part counts
    (every launch adds 32 to a slab's counter whatever its part count); no part ever timed out waiting."""
    ops, L = _ops()
    monkeypatch.setattr(ops, "GN_MULTI", True)        # (opt-in: the host passes the sync buffer only when the module flag is set)
    C = C1 + C2
    x1 = rnd((B, HW, C1), 40, scale=2.0) + 3.0
    x2 = rnd((B, HW, C2), 44, scale=3.0) if C2 else None
    g, b = rnd((C,), 41) * 0.1 + 1.0, rnd((C,), 42, scale=0.1)
    full = torch.cat([x1, x2], -1) if C2 else x1
    ref = F.silu(F.group_norm(full.float().transpose(1, 2), G, g.float(), b.float(), 1e-5)).transpose(1, 2)
    monkeypatch.setenv("DA_GN_MULTI", "0")
    monkeypatch.setenv("DA_GN_FUSED", "0")
    two = ops.group_norm_nhwc(x1, g, b, G, 1e-5, silu=True, x2=x2)
    monkeypatch.setenv("DA_GN_MULTI", "1")
    monkeypatch.setenv("DA_GN_FUSED", "1")
    one = ops.group_norm_nhwc(x1, g, b, G, 1e-5, silu=True, x2=x2)
    what = f"groupnorm one launch / several workgroups B{B} HW{HW} C{C1}+{C2} G{G}"
    assert_close_bf16(one, ref, what, rtol=1.6e-2, atol_rms=8e-3)
    d = (one.float() - two.float()).abs()
    tol = 2.0 ** -6 * two.float().abs() + 1e-3
    assert int((d > tol).sum()) == 0, f"{what}: differs from the two-kernel form by more than a bf16 ulp ({float(d.max()):.3e})"
    other = rnd((2, 2048, 640), 77)
    go, bo = rnd((640,), 78) * 0.1 + 1.0, rnd((640,), 79, scale=0.1)
    for i in range(6):
        if i % 2:
            ops.group_norm_nhwc(other, go, bo, 32, 1e-5)          # another shape (another part count) on the same counters
        assert torch.equal(one, ops.group_norm_nhwc(x1, g, b, G, 1e-5, silu=True, x2=x2)), f"repeat {i}"
    torch.cuda.synchronize()
    assert not ops.gn_sync_error()


def test_groupnorm_several_workgroups_replays_from_a_hip_graph(monkeypatch):
    """The arrival counters are monotonic and never reset: a captured chain of three GroupNorms of different part counts replays
    twenty times with the eager result."""
    ops, L = _ops()
    monkeypatch.setattr(ops, "GN_MULTI", True)
    shapes = [(2, 4096, 640), (2, 4096, 1280), (2, 1024, 2560)]
    xs = [rnd(sh, 60 + i, scale=1.5) + 0.7 for i, sh in enumerate(shapes)]
    gs = [rnd((sh[2],), 70 + i) * 0.1 + 1.0 for i, sh in enumerate(shapes)]
    bs = [rnd((sh[2],), 80 + i, scale=0.1) for i, sh in enumerate(shapes)]
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        eager = [ops.group_norm_nhwc(x, g, b, 32, 1e-5, silu=True).clone() for x, g, b in zip(xs, gs, bs)]
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            outs = [ops.group_norm_nhwc(x, g, b, 32, 1e-5, silu=True) for x, g, b in zip(xs, gs, bs)]
    torch.cuda.synchronize()
    for i in range(20):
        for o in outs:
            o.zero_()
        gr.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(o, e) for o, e in zip(outs, eager)), f"replay {i}"
    assert not ops.gn_sync_error()


@pytest.mark.parametrize("M,C", [(100, 320), (2048, 640), (513, 1280), (64, 3072), (7, 1536), (33, 64)])
def test_layernorm(M, C):
    ops, L = _ops()
    x = rnd((M, C), 47, scale=1.5) + 0.3
    g, b = rnd((C,), 48) * 0.1 + 1.0, rnd((C,), 49, scale=0.1)
    y = ops.layer_norm(x, g, b, 1e-5)
    assert_close_bf16(y, F.layer_norm(x.float(), (C,), g.float(), b.float(), 1e-5), f"layernorm {M}x{C}", rtol=8e-3, atol_rms=4e-3)
    sc, sh = rnd((2, C), 50, scale=0.3), rnd((2, C), 51, scale=0.3)
    if M % 2 == 0:
        y2 = ops.layer_norm(x, None, None, 1e-6, mod_scale=sc, mod_shift=sh, rows_per_batch=M // 2)
        ln = F.layer_norm(x.float(), (C,), None, None, 1e-6)
        ref = ln * (1 + sc.float().repeat_interleave(M // 2, 0)) + sh.float().repeat_interleave(M // 2, 0)
        assert_close_bf16(y2, ref, f"adaLN {M}x{C}", rtol=1.6e-2, atol_rms=8e-3)


def test_softmax_rows():
    ops, L = _ops()
    s = (torch.randn((300, 4096), generator=torch.Generator().manual_seed(52)) * 3).to(DEV)
    p = ops.softmax_rows(s)
    assert_close_bf16(p, torch.softmax(s.float(), -1), "softmax rows", rtol=8e-3, atol_rms=1e-3)


# ----------------------------------------------------------------------------------------------------------------------
# sampler: bit-exact against trajectories produced by the real reference schedulers (tests/golden/schedulers.npz)
# ----------------------------------------------------------------------------------------------------------------------
def _golden_err(got, ref):
    """max |got - ref| in units of the reference tensor's rms (outputs of a step have cancellation: no ulp metric)."""
    ref = ref.float()
    return float((got.float() - ref).abs().max() / ref.pow(2).mean().sqrt())


@pytest.mark.parametrize("dt_name", ["f32", "bf16"])
def test_sampler_vs_oracle_and_golden(golden, dt_name):
    """Every fused scheduler.step kernel, step by step on identical inputs:
      * BIT-EXACT against the oracle restatement of the reference run on this host (oracle/samplers.py with
        ``device_scalars=True`` = torch's device-kernel scalar semantics, see its docstring);
      * against the frozen reference trajectories in tests/golden/schedulers.npz (produced by the reference's own CPU
        kernels on another host) per step on the golden's inputs: max abs error <= 2^-5 of the step output rms (1 bf16 ulp of a 4-sigma element) for
        bf16 (torch's CPU kernels round a leading 0-d fp32 scalar to bf16 first, a 2^-9 relative perturbation of one
        term), <= 2e-6 rms for fp32 (host libm / vectoriser differences in the schedule tables)."""
    from diffusers_amd import schedulers as S
    from oracle import samplers as OS
    ops, L = _ops()
    gz = golden("schedulers")
    dt = torch.float32 if dt_name == "f32" else bf16
    x0 = torch.from_numpy(gz[f"x0_{dt_name}"]).to(dt)
    eps = torch.from_numpy(gz[f"eps_{dt_name}"]).to(dt)
    tol = 2e-6 if dt_name == "f32" else 2.0 ** -5

    def gold(name):
        return torch.from_numpy(gz[name]).to(dt)

    def cmp(got, want, ref_gold, name):
        got = got.cpu()
        nbad = int((got.float() != want.float()).sum())
        err = _golden_err(got, ref_gold)
        print(f"[parity] {name}: vs same-host oracle mismatches = {nbad}/{got.numel()}; vs golden max err {err:.2e} rms")
        assert nbad == 0, f"{name}: {nbad} elements differ from the oracle"
        assert err <= tol, f"{name}: {err:.3e} (rms units) from the golden reference step"

    # ---- Euler (SDXL): scale_model_input + step ----
    sk = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1,
              timestep_spacing="leading")
    e = S.EulerDiscreteScheduler(**sk)
    e.set_timesteps(5, device=DEV)
    o = OS.EulerOracle(device_scalars=True, **sk)
    o.set_timesteps(5)
    assert torch.equal(e.sigmas, o.sigmas)
    start = ops.mul_scalar(x0.to(DEV), float(e.init_noise_sigma))
    assert torch.equal(start.cpu(), (x0 * o.init_noise_sigma).to(dt))
    g_traj, g_scaled = gold(f"euler_traj_{dt_name}"), gold(f"euler_scaled_{dt_name}")
    x_in = gold(f"euler_start_{dt_name}")
    for i, t in enumerate(e.timesteps):
        o.step_index = i
        e.reset(i)
        cmp(e.scale_model_input(x_in.to(DEV), t), o.scale_model_input(x_in), g_scaled[i], f"euler scale[{i}] {dt_name}")
        e.reset(i)
        cmp(e.step(eps[i].to(DEV), t, x_in.to(DEV)).prev_sample, o.step(eps[i], x_in), g_traj[i],
            f"euler step[{i}] {dt_name}")
        x_in = g_traj[i]

    # ---- DDIM (SD1.5) ----
    dk = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
              set_alpha_to_one=False, steps_offset=1)
    d = S.DDIMScheduler(**dk)
    d.set_timesteps(5, device=DEV)
    od = OS.DDIMOracle(device_scalars=True, **dk)
    od.set_timesteps(5)
    g_traj = gold(f"ddim_traj_{dt_name}")
    x_in = x0
    for i, t in enumerate(d.timesteps):
        d.reset(i)
        cmp(d.step(eps[i].to(DEV), t, x_in.to(DEV)).prev_sample, od.step(eps[i], od.timesteps[i], x_in), g_traj[i],
            f"ddim step[{i}] {dt_name}")
        x_in = g_traj[i]

    # ---- DDPM (ancestral noise from the same CPU generator stream as the reference) ----
    pk = dict(beta_start=0.0001, beta_end=0.02, beta_schedule="linear", variance_type="fixed_small", clip_sample=True)
    p = S.DDPMScheduler(**pk)
    p.set_timesteps(5, device=DEV)
    op_ = OS.DDPMOracle(device_scalars=True, **pk)
    op_.set_timesteps(5)
    g_traj = gold(f"ddpm_traj_{dt_name}")
    g1, g2 = torch.Generator("cpu").manual_seed(9), torch.Generator("cpu").manual_seed(9)
    x_in = x0
    for i, t in enumerate(p.timesteps):
        p.reset(i)
        cmp(p.step(eps[i].to(DEV), t, x_in.to(DEV), generator=g1).prev_sample,
            op_.step(eps[i], op_.timesteps[i], x_in, generator=g2), g_traj[i], f"ddpm step[{i}] {dt_name}")
        x_in = g_traj[i]

    # ---- FlowMatch Euler (Flux) ----
    f = S.FlowMatchEulerDiscreteScheduler(shift=1.0)
    f.set_timesteps(sigmas=np.linspace(1.0, 1 / 5, 5), device=DEV)
    of = OS.FlowMatchOracle(shift=1.0, device_scalars=True)
    of.set_timesteps(sigmas=np.linspace(1.0, 1 / 5, 5))
    g_traj = gold(f"flow_traj_{dt_name}")
    x_in = x0
    for i, t in enumerate(f.timesteps):
        f.reset(i)
        of.step_index = i
        cmp(f.step(eps[i].to(DEV), t, x_in.to(DEV)).prev_sample, of.step(eps[i], x_in), g_traj[i],
            f"flow step[{i}] {dt_name}")
        x_in = g_traj[i]


@pytest.mark.parametrize("dt_name", ["f32", "bf16"])
def test_sampler_round2_cases_vs_oracle_and_golden(golden, dt_name):
    """Prediction types (v_prediction / sample) of the Euler, DDIM and DDPM kernels, DDIM eta > 0, DDPM on spacings whose
    previous timestep is not t - 1000 // n, FlowMatch with an fp32 sample and a bf16 model output: per step on the
    golden's inputs, BIT-EXACT against the oracle with device scalar semantics and within the sampler tolerance of the
    frozen reference run (tests/golden/schedulers_r2.npz; see test_sampler_vs_oracle_and_golden for the bound)."""
    from diffusers_amd import schedulers as S
    from oracle import samplers as OS
    gz = golden("schedulers_r2")
    dt = torch.float32 if dt_name == "f32" else bf16
    # vs the frozen CPU run: fp32 as in test_sampler_vs_oracle_and_golden; bf16 2 ulp of a 4-sigma element (2^-4 rms):
    # the v_prediction / sample forms have TWO leading-scalar products per output (x0 and pred_epsilon), each of which
    # torch's CPU kernels perturb by rounding the 0-d fp32 scalar to bf16 first (oracle/samplers.py docstring) -- the
    # device kernels, like torch's own device kernels, keep the scalar in fp32.  The oracle in device-scalar mode is
    # matched bit for bit either way.
    tol = 2e-6 if dt_name == "f32" else 2.0 ** -4
    N = 6
    load = lambda k: torch.from_numpy(gz[k]).to(dt)  # noqa: E731
    x0, eps, noise, draws = load(f"x0_{dt_name}"), load(f"eps_{dt_name}"), load(f"noise_{dt_name}"), load(f"ddpm_draws_{dt_name}")
    sd = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")

    def cmp(got, want, ref_gold, name):
        got = got.cpu()
        nbad = int((got.float() != want.float()).sum())
        err = _golden_err(got, ref_gold)
        print(f"[parity] {name}: vs same-host oracle mismatches = {nbad}/{got.numel()}; vs golden max err {err:.2e} rms")
        assert nbad == 0, f"{name}: {nbad} elements differ from the oracle"
        assert err <= tol, f"{name}: {err:.3e} (rms units) from the golden reference step"

    for pred in ("v_prediction", "sample"):
        dk = dict(clip_sample=False, set_alpha_to_one=False, steps_offset=1, prediction_type=pred, **sd)
        d, od = S.DDIMScheduler(**dk), OS.DDIMOracle(device_scalars=True, **dk)
        d.set_timesteps(N, device=DEV), od.set_timesteps(N)
        gt, x_in = load(f"ddim_{pred}_{dt_name}"), x0
        for i, t in enumerate(d.timesteps):
            d.reset(i)
            cmp(d.step(eps[i].to(DEV), t, x_in.to(DEV)).prev_sample, od.step(eps[i], od.timesteps[i], x_in), gt[i],
                f"ddim {pred}[{i}] {dt_name}")
            x_in = gt[i]
        ek = dict(steps_offset=1, timestep_spacing="leading", prediction_type=pred, **sd)
        e, oe = S.EulerDiscreteScheduler(**ek), OS.EulerOracle(device_scalars=True, **ek)
        e.set_timesteps(N, device=DEV), oe.set_timesteps(N)
        gt, x_in = load(f"euler_{pred}_{dt_name}"), load(f"euler_{pred}_start_{dt_name}")
        for i, t in enumerate(e.timesteps):
            e.reset(i)
            oe.step_index = i
            cmp(e.step(eps[i].to(DEV), t, x_in.to(DEV)).prev_sample, oe.step(eps[i], x_in), gt[i],
                f"euler {pred}[{i}] {dt_name}")
            x_in = gt[i]
        pk = dict(prediction_type=pred, clip_sample=True)
        p, op_ = S.DDPMScheduler(**pk), OS.DDPMOracle(device_scalars=True, **pk)
        p.set_timesteps(N, device=DEV), op_.set_timesteps(N)
        gt, x_in = load(f"ddpm_{pred}_{dt_name}"), x0
        for i, t in enumerate(p.timesteps):
            p.reset(i)
            cmp(p.step(eps[i].to(DEV), t, x_in.to(DEV), noise=draws[i].to(DEV)).prev_sample,
                op_.step(eps[i], op_.timesteps[i], x_in, noise=draws[i]), gt[i], f"ddpm {pred}[{i}] {dt_name}")
            x_in = gt[i]

    for pred in ("epsilon", "v_prediction"):        # DDIM eta = 0.6, caller-supplied variance noise
        dk = dict(clip_sample=False, set_alpha_to_one=False, steps_offset=1, prediction_type=pred, **sd)
        d, od = S.DDIMScheduler(**dk), OS.DDIMOracle(device_scalars=True, **dk)
        d.set_timesteps(N, device=DEV), od.set_timesteps(N)
        gt, x_in = load(f"ddim_eta_{pred}_{dt_name}"), x0
        for i, t in enumerate(d.timesteps):
            d.reset(i)
            cmp(d.step(eps[i].to(DEV), t, x_in.to(DEV), eta=0.6, variance_noise=noise[i].to(DEV)).prev_sample,
                od.step(eps[i], od.timesteps[i], x_in, eta=0.6, variance_noise=noise[i]), gt[i],
                f"ddim eta {pred}[{i}] {dt_name}")
            x_in = gt[i]

    for spacing in ("linspace", "trailing", "leading"):
        p, op_ = S.DDPMScheduler(timestep_spacing=spacing, clip_sample=True), OS.DDPMOracle(timestep_spacing=spacing, clip_sample=True, device_scalars=True)
        p.set_timesteps(7, device=DEV), op_.set_timesteps(7)
        assert np.array_equal(p.timesteps.cpu().numpy(), gz[f"ddpm_{spacing}7_timesteps"])
        gt, x_in = load(f"ddpm_{spacing}7_{dt_name}"), x0
        for i, t in enumerate(p.timesteps[:N]):
            p.reset(i)
            cmp(p.step(eps[i].to(DEV), t, x_in.to(DEV), noise=draws[i].to(DEV)).prev_sample,
                op_.step(eps[i], op_.timesteps[i], x_in, noise=draws[i]), gt[i], f"ddpm {spacing}7[{i}] {dt_name}")
            x_in = gt[i]


def test_flowmatch_fp32_sample_bf16_output_and_operand_checks(golden):
    """The reference's Wan hand-over: FlowMatchEulerDiscreteScheduler.step(model_output=bf16, sample=fp32) forms the
    update in fp32 and returns the MODEL OUTPUT's dtype (scheduling_flow_match_euler_discrete.py:484,:517) -- bit-exact
    against the frozen reference run; and the operand checks ADVICE r1 asked for (dtype / size / contiguity raise instead
    of reading out of bounds)."""
    from diffusers_amd import schedulers as S
    from oracle import samplers as OS
    ops, L = _ops()
    gz = golden("schedulers_r2")
    f = S.FlowMatchEulerDiscreteScheduler(shift=3.0)
    f.set_timesteps(6, device=DEV)
    of = OS.FlowMatchOracle(shift=3.0, device_scalars=True)
    of.set_timesteps(6)
    assert np.array_equal(f.sigmas.cpu().numpy(), gz["flow_mixed_sigmas"])
    x = torch.from_numpy(gz["x0_f32"])
    v = torch.from_numpy(gz["eps_bf16"]).to(bf16)
    for i, t in enumerate(f.timesteps):
        y = f.step(v[i].to(DEV), t, x.to(DEV)).prev_sample
        of.step_index = i
        want = of.step(v[i], x)                     # device scalar semantics: dt stays fp32 in dt * model_output
        gold = torch.from_numpy(gz["flow_mixed"][i])  # the CPU run rounds dt to bf16 first: within 1 bf16 ulp of it
        assert y.dtype == bf16 and torch.equal(y.cpu(), want), f"step {i}: differs from the oracle"
        assert _golden_err(y.cpu(), gold) <= 2.0 ** -5, f"step {i}: too far from the frozen reference run"
        x = gold.float()
    f.reset(0)
    x = x.to(DEV)
    v = v.to(DEV)
    xb = x.to(bf16)
    with pytest.raises(TypeError):          # bf16 sample with an fp32 model output is not a reference combination
        ops.flowmatch_step(v[0].float(), xb, f.device_table, f.device_step)
    with pytest.raises(ValueError):         # wrong-sized model output with cfg
        ops.flowmatch_step(v[0], xb, f.device_table, f.device_step, cfg=True)
    with pytest.raises(ValueError):         # non-contiguous sample
        ops.flowmatch_step(v[0], xb.transpose(-1, -2), f.device_table, f.device_step)
    with pytest.raises(TypeError):          # `out` must have the model output's dtype
        ops.flowmatch_step(v[0], x, f.device_table, f.device_step, out=torch.empty_like(x))
    with pytest.raises(TypeError):
        ops.euler_step(v[0], x, f.device_table, f.device_step, cfg=False, guidance=0.0)


@pytest.mark.parametrize("dt_name", ["f32", "bf16"])
def test_fused_cfg_step_equals_unfused(golden, dt_name):
    """step_cfg([u;c]) == step(u + g*(c-u)) bit for bit, and both equal the oracle's CFG combine + Euler step."""
    from diffusers_amd import schedulers as S
    from oracle.samplers import EulerOracle, cfg_combine
    gz = golden("schedulers")
    dt = torch.float32 if dt_name == "f32" else bf16
    eps = torch.from_numpy(gz[f"eps_{dt_name}"]).to(dt)
    x0 = torch.from_numpy(gz[f"x0_{dt_name}"]).to(dt)
    u, c = eps[0], eps[1]
    comb = cfg_combine(u, c, 7.5)
    assert torch.equal(comb.float(), torch.from_numpy(gz[f"cfg_{dt_name}"]))
    sk = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="leading",
              steps_offset=1)
    o = EulerOracle(device_scalars=True, **sk)
    o.set_timesteps(5)
    want = o.step(comb, x0)
    e = S.EulerDiscreteScheduler(**sk)
    e.set_timesteps(5, device=DEV)
    got = e.step_cfg(torch.cat([u, c], 0).to(DEV), x0.to(DEV), 7.5)
    e.set_timesteps(5, device=DEV)
    got2 = e.step(comb.to(DEV), e.timesteps[0], x0.to(DEV)).prev_sample
    assert torch.equal(got, got2), "fused CFG+Euler differs from combine-then-step"
    assert torch.equal(got.cpu(), want), "fused CFG+Euler differs from the oracle's order of operations"


# ----------------------------------------------------------------------------------------------------------------------
# misc
# ----------------------------------------------------------------------------------------------------------------------
def test_timestep_embedding():
    from oracle.reference_math import timestep_embedding as ref_te
    ops, L = _ops()
    t = torch.tensor([981.0, 1.0, 500.5, 0.0])
    for dim, flip, shift in ((320, True, 0.0), (256, True, 0.0), (128, False, 1.0)):
        got = ops.timestep_embedding(t.to(DEV), dim, batch=4, flip_sin_to_cos=flip, shift=shift, out_f32=True)
        ref = ref_te(t, dim, flip, shift)
        err = float((got.cpu() - ref).abs().max())
        print(f"[parity] timestep_embedding dim={dim}: max_abs={err:.3e}")
        assert err < 2e-3  # sin/cos of arguments up to 1e3: fp32 argument rounding differs by ulps between libms
        got16 = ops.timestep_embedding(t.to(DEV), dim, batch=4, flip_sin_to_cos=flip, shift=shift)
        assert float((got16.float().cpu() - ref).abs().max()) < 6e-3


@pytest.mark.parametrize("M,N,K", [(2, 1280, 320), (2, 1280, 2816), (1, 320, 1280), (6, 64, 256), (8, 132, 64)])
def test_linear_small_m(M, N, K):
    ops, L = _ops()
    x, w, b = rnd((M, K), 60), rnd((N, K), 61, scale=K ** -0.5), rnd((N,), 62, scale=0.1)
    res = rnd((M, N), 63)
    y = ops.linear_small_m(x, w, b)
    assert_close_bf16(y, x.float() @ w.float().t() + b.float(), f"small-M linear {M}x{N}x{K}", rtol=8e-3, atol_rms=4e-3)
    y = ops.linear_small_m(x, w, b, act_in=L.ACT_SILU)
    assert_close_bf16(y, F.silu(x.float()).to(bf16).float() @ w.float().t() + b.float(), "small-M silu-in", rtol=8e-3, atol_rms=4e-3)
    y = ops.linear_small_m(x, w, b, act_out=L.ACT_SILU, residual=res)
    assert_close_bf16(y, F.silu(x.float() @ w.float().t() + b.float()) + res.float(), "small-M silu-out+res", rtol=1.6e-2, atol_rms=8e-3)


@pytest.mark.parametrize("B,Cin,H,W,Cout,nchw", [(2, 4, 128, 128, 320, True), (1, 3, 64, 64, 128, True), (2, 4, 17, 30, 64, True),
                                                 (1, 4, 9, 5, 16, False), (1, 3, 8, 3, 8, True)])
def test_conv_thin_in_four_pixels_per_thread_is_bit_identical(B, Cin, H, W, Cout, nchw, monkeypatch):
    """Round 6: U-Net conv_in (4 -> 320 at 128 x 128) / the DDPM U-Net's (3 -> 128) with four output pixels of a row per thread:
    the same operations in the same order per output element as the one-pixel kernel -- equal bits, also on ragged widths (W % 4 != 0),
    image borders, channels-last input and with the latent scaling / shift conversions."""
    ops, L = _ops()
    x = rnd((B, Cin, H, W), 90) if nchw else rnd((B, H, W, Cin), 90)
    w, b = rnd((Cout, Cin, 3, 3), 91, scale=0.2), rnd((Cout,), 92)
    wp = ops.pack_conv_weight(w)
    for kw in (dict(), dict(in_div=0.13025), dict(in_div=0.3611, in_add=0.1159)):
        monkeypatch.setenv("DA_CONV_IN_QUAD", "0")
        one = ops.conv_thin_in(x, wp, b, ksize=3, in_nchw=nchw, **kw)
        monkeypatch.setenv("DA_CONV_IN_QUAD", "1")
        four = ops.conv_thin_in(x, wp, b, ksize=3, in_nchw=nchw, **kw)
        assert torch.equal(one, four), f"conv_thin_in quad differs {kw}"
    xr = x.float() if nchw else x.float().permute(0, 3, 1, 2)
    ref = F.conv2d(xr, w.float(), b.float(), padding=1).permute(0, 2, 3, 1)
    assert_close_bf16(four if not kw else ops.conv_thin_in(x, wp, b, ksize=3, in_nchw=nchw), ref, "conv_thin_in quad vs torch", rtol=8e-3, atol_rms=4e-3)


def test_conv_thin_in_out():
    ops, L = _ops()
    B, Cin, H, W, Cout = 2, 4, 16, 16, 64
    x = rnd((B, Cin, H, W), 64)
    w, b = rnd((Cout, Cin, 3, 3), 65, scale=(9 * Cin) ** -0.5), rnd((Cout,), 66, scale=0.1)
    y = ops.conv_thin_in(x, ops.pack_conv_weight(w), b, ksize=3, in_nchw=True)
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1).permute(0, 2, 3, 1)
    assert_close_bf16(y, ref, "conv_thin_in 3x3 nchw", rtol=8e-3, atol_rms=4e-3)
    y = ops.conv_thin_in(x, ops.pack_conv_weight(w), b, ksize=3, in_nchw=True, in_div=0.13025)
    ref = F.conv2d((x.float() / 0.13025).to(bf16).float(), w.float(), b.float(), padding=1).permute(0, 2, 3, 1)
    assert_close_bf16(y, ref, "conv_thin_in with latents/scaling_factor", rtol=8e-3, atol_rms=4e-3)
    w1 = rnd((8, Cin, 1, 1), 67, scale=0.5)
    y = ops.conv_thin_in(x, w1.reshape(8, Cin).contiguous(), None, ksize=1, in_nchw=True)
    assert_close_bf16(y, F.conv2d(x.float(), w1.float()).permute(0, 2, 3, 1), "conv_thin_in 1x1", rtol=8e-3, atol_rms=4e-3)
    # 16 latent channels -> 512 (Flux VAE conv_in): output channels are processed in LDS-sized chunks (+ shift_factor)
    x16 = rnd((1, 16, 12, 10), 91)
    w16, b16 = rnd((512, 16, 3, 3), 92, scale=(9 * 16) ** -0.5), rnd((512,), 93, scale=0.1)
    y = ops.conv_thin_in(x16, ops.pack_conv_weight(w16), b16, ksize=3, in_nchw=True, in_div=0.3611, in_add=0.1159)
    xin = ((x16.float() / 0.3611).to(bf16).float() + 0.1159).to(bf16).float()
    assert_close_bf16(y, F.conv2d(xin, w16.float(), b16.float(), padding=1).permute(0, 2, 3, 1),
                      "conv_thin_in 16->512 chunked, /scale + shift", rtol=8e-3, atol_rms=4e-3)
    xh = rnd((B, H, W, 128), 68)
    for co in (3, 4):
        wo, bo = rnd((co, 128, 3, 3), 69, scale=(9 * 128) ** -0.5), rnd((co,), 70, scale=0.1)
        y = ops.conv_thin_out(xh, ops.pack_conv_weight(wo), bo)
        ref = F.conv2d(xh.float().permute(0, 3, 1, 2), wo.float(), bo.float(), padding=1)
        assert y.shape == ref.shape
        assert_close_bf16(y, ref, f"conv_thin_out Cout={co}", rtol=8e-3, atol_rms=4e-3)
    # enough pixels and whole K slices (Cin % 64 == 0): the same call runs on the MFMA implicit-GEMM kernel with the output
    # channels zero-padded to 16, then nhwc_take_nchw lays out the planes (U-Net conv_out 320 -> 4, VAE conv_out 128 -> 3)
    for (Bb, Hh, Ww, ci, co) in [(2, 72, 64, 320, 4), (1, 96, 80, 128, 3), (1, 64, 64, 64, 1)]:
        xb = rnd((Bb, Hh, Ww, ci), 71)
        wo, bo = rnd((co, ci, 3, 3), 72, scale=(9 * ci) ** -0.5), rnd((co,), 73, scale=0.1)
        wpk = ops.pack_conv_weight(wo)
        y = ops.conv_thin_out(xb, wpk, bo)
        ref = F.conv2d(xb.float().permute(0, 3, 1, 2), wo.float(), bo.float(), padding=1)
        assert y.shape == ref.shape and y.is_contiguous()
        assert_close_bf16(y, ref, f"conv_thin_out via igemm {ci}->{co}", rtol=8e-3, atol_rms=4e-3)
        assert torch.equal(y, ops.conv_thin_out(xb, wpk, bo))          # padded weights come from the cache the second time


# ----------------------------------------------------------------------------------------------------------------------
# Flux-side kernels: gate / row-bias GEMM epilogues, RMSNorm + RoPE
# ----------------------------------------------------------------------------------------------------------------------
def test_gemm_gate_and_row_bias_epilogues():
    ops, L = _ops()
    M, N, K, B = 640, 384, 256, 2
    x, w = rnd((M, K), 61), rnd((N, K), 62, scale=K ** -0.5)
    bias, res, gate = rnd((N,), 63), rnd((M, N), 64), rnd((B, 3 * N), 65)
    g = gate[:, N:2 * N]                                    # a column chunk of the adaLN projection (ld = 3N)
    y = ops.linear(x, w, bias, gate=g, rows_per_batch=M // B, residual=res)
    lin = (x.float() @ w.float().t() + bias.float()).to(bf16).float()
    gated = (lin * g.float().repeat_interleave(M // B, 0)).to(bf16).float()
    assert_close_bf16(y, res.float() + gated, "gemm gate*(xW+b)+residual", rtol=8e-3, atol_rms=4e-3)
    # swapped product with the Linear's bias along M (V^T = W_v . X^T + b_v 1^T), written into a column block
    bm = rnd((M,), 66)
    wide = torch.zeros((M, 2 * N), device=DEV, dtype=bf16)
    ops.linear(x, w, bias_rows=bm, out=wide[:, N:])
    assert_close_bf16(wide[:, N:], x.float() @ w.float().t() + bm.float()[:, None], "gemm row bias", rtol=8e-3, atol_rms=4e-3)
    assert float(wide[:, :N].float().abs().max()) == 0.0


@pytest.mark.parametrize("D,heads", [(64, 2), (128, 24)])
def test_rmsnorm_rope_per_head(D, heads):
    ops, L = _ops()
    rows, St = 80, 16
    Cc = heads * D
    x = rnd((rows, 3 * Cc), 71)
    wq, wk = rnd((D,), 72, scale=0.2) + 1, rnd((D,), 73, scale=0.2) + 1
    gcpu = torch.Generator("cpu").manual_seed(74)
    ang = torch.rand((rows, D // 2), generator=gcpu) * 6.0
    cos = ang.cos().repeat_interleave(2, 1).contiguous().to(DEV)
    sin = ang.sin().repeat_interleave(2, 1).contiguous().to(DEV)
    ref = x.float().clone()

    def ref_part(block, w_, lo, hi):
        v = block.view(-1, heads, D)
        v = F.rms_norm(v, (D,), w_.float(), 1e-6).to(bf16).float()
        c, s = cos[lo:hi].float().cpu()[:, None, :], sin[lo:hi].float().cpu()[:, None, :]
        xr, xi = v.reshape(*v.shape[:-1], -1, 2).unbind(-1)
        rot = torch.stack([-xi, xr], -1).flatten(2)
        return (v * c + rot * s).reshape(-1, Cc)
    xc = x.float().cpu()
    want_q = ref_part(xc[St:, :Cc], wq.cpu(), St, rows)
    want_k = ref_part(xc[St:, Cc:2 * Cc], wk.cpu(), St, rows)
    y = x.clone()
    ops.rmsnorm_rope_(y[St:], heads=heads, head_dim=D, col_offsets=(0, Cc), weights=(wq, wk), eps=1e-6, cos=cos, sin=sin,
                      rope_row0=St)
    assert torch.equal(y[:St], x[:St]) and torch.equal(y[:, 2 * Cc:], x[:, 2 * Cc:]), "rows / columns outside the view changed"
    assert_close_bf16(y[St:, :Cc], want_q, f"rmsnorm+rope q D={D}", rtol=1.6e-2, atol_rms=8e-3)
    assert_close_bf16(y[St:, Cc:2 * Cc], want_k, f"rmsnorm+rope k D={D}", rtol=1.6e-2, atol_rms=8e-3)
    # norm only / rope only
    y2 = x.clone()
    ops.rmsnorm_rope_(y2, heads=heads, head_dim=D, col_offsets=(0,), weights=(wq,), eps=1e-6)
    want = F.rms_norm(xc[:, :Cc].view(-1, heads, D), (D,), wq.float().cpu(), 1e-6).reshape(-1, Cc)
    assert_close_bf16(y2[:, :Cc], want, "rmsnorm only", rtol=1.6e-2, atol_rms=8e-3)


@pytest.mark.parametrize("D,S,Skv", [(96, 320, 320), (160, 256, 256), (160, 64, 77), (96, 1024, 77)])
def test_attention_sd15_head_sizes(D, S, Skv):
    """D = 96 / 160 kernels (SD1.5 head dims 80 -> 96 zero padded, 160) against fp32 SDPA."""
    ops, L = _ops()
    B, H = 2, 8
    inner = H * D
    sa = ((Skv + 15) // 16) * 16
    q = rnd((B * S, inner), 95)
    k = torch.zeros((B * sa, inner), device=DEV, dtype=bf16)
    v = torch.zeros((B * sa, inner), device=DEV, dtype=bf16)
    kk, vv = rnd((B, Skv, inner), 96), rnd((B, Skv, inner), 97)
    k.view(B, sa, inner)[:, :Skv] = kk
    v.view(B, sa, inner)[:, :Skv] = vv
    vt = v.t().contiguous()
    o = ops.attention(q, k, vt, B=B, H=H, D=D, Sq=S, Skv=Skv, Skv_alloc=sa, q_row_stride=inner, k_row_stride=inner,
                      q_batch_stride=S * inner, k_batch_stride=sa * inner, vt_ld=B * sa, vt_batch_stride=sa,
                      scale=80 ** -0.5)
    qh = q.float().view(B, S, H, D).transpose(1, 2)
    kh = kk.float().view(B, Skv, H, D).transpose(1, 2)
    vh = vv.float().view(B, Skv, H, D).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qh.cpu(), kh.cpu(), vh.cpu(), scale=80 ** -0.5).transpose(1, 2).reshape(B * S, inner)
    assert_close_bf16(o, ref, f"attention D={D} S={S} Skv={Skv}", rtol=1.6e-2, atol_rms=1.6e-2)


def test_rmsnorm_across_heads():
    ops, L = _ops()
    rows, heads, D = 70, 12, 128
    Cc = heads * D
    x = rnd((rows, Cc), 81)
    w = rnd((Cc,), 82, scale=0.2) + 1
    y = x.clone()
    ops.rmsnorm_rope_(y, heads=heads, head_dim=D, col_offsets=(0,), weights=(w,), eps=1e-6, norm="across_heads")
    want = F.rms_norm(x.float().cpu(), (Cc,), w.float().cpu(), 1e-6)
    assert_close_bf16(y, want, "rmsnorm across heads", rtol=1.6e-2, atol_rms=8e-3)


# ----------------------------------------------------------------------------------------------------------------------
# Wan-side kernels: patch gather / scatter, fp32 modulation + gate, broadcast add
# ----------------------------------------------------------------------------------------------------------------------
def test_patchify_unpatchify_3d():
    ops, L = _ops()
    B, Cc, Fr, H, W = 2, 16, 3, 8, 6
    x = rnd((B, Cc, Fr, H, W), 101)
    tok = ops.patchify3d(x, (1, 2, 2))
    w = rnd((32, Cc, 1, 2, 2), 102, scale=0.1)
    want = F.conv3d(x.float(), w.float(), stride=(1, 2, 2)).flatten(2).transpose(1, 2).reshape(-1, 32)
    got = tok.float() @ w.float().reshape(32, -1).t()
    assert torch.allclose(got.cpu(), want.cpu(), atol=1e-4), "patchify feature order != Conv3d weight order"
    # unpatchify == the reference's reshape / permute / flatten (transformer_wan.py:727-731)
    y = rnd((B * Fr * (H // 2) * (W // 2), 4 * Cc), 103)
    ref = y.reshape(B, Fr, H // 2, W // 2, 1, 2, 2, -1).permute(0, 7, 1, 4, 2, 5, 3, 6).flatten(6, 7).flatten(4, 5).flatten(2, 3)
    assert torch.equal(ops.unpatchify3d(y, (B, Cc, Fr, H, W), (1, 2, 2)), ref.contiguous())


def test_fp32_modulation_gate_and_bcast_add():
    ops, L = _ops()
    M, Cc, B = 96, 256, 2
    x = rnd((M, Cc), 111)
    tab = torch.randn(6 * Cc, generator=torch.Generator("cpu").manual_seed(112)).to(DEV)
    proj = rnd((B, 6 * Cc), 113, scale=0.3)
    mod = ops.bcast_add_f32(tab, proj)
    want_mod = tab[None, :] + proj.float()
    assert mod.dtype == torch.float32 and torch.equal(mod, want_mod)
    sc, sh, gt = mod[:, Cc:2 * Cc], mod[:, :Cc], mod[:, 2 * Cc:3 * Cc]
    y = ops.layer_norm(x, None, None, 1e-6, mod_scale=sc, mod_shift=sh, rows_per_batch=M // B)
    ln = F.layer_norm(x.float(), (Cc,), None, None, 1e-6)
    ref = ln * (1 + sc.repeat_interleave(M // B, 0)) + sh.repeat_interleave(M // B, 0)
    assert_close_bf16(y, ref, "fp32 LayerNorm * (1 + scale) + shift", rtol=8e-3, atol_rms=4e-3)
    w, bias, res = rnd((Cc, Cc), 114, scale=Cc ** -0.5), rnd((Cc,), 115), rnd((M, Cc), 116)
    z = ops.linear(x, w, bias, gate=gt, rows_per_batch=M // B, residual=res)
    lin = (x.float() @ w.float().t() + bias.float()).to(bf16).float()
    assert_close_bf16(z, res.float() + lin * gt.repeat_interleave(M // B, 0), "gemm fp32 gate + residual", rtol=8e-3,
                      atol_rms=4e-3)


def test_conv_downsample_asymmetric_pad():
    """Downsample2D with padding=0: F.pad(x, (0, 1, 0, 1)) then conv3x3 stride 2 (downsampling.py:139-147)."""
    ops, L = _ops()
    B, H, W, Cc = 2, 16, 12, 64
    x = rnd((B, H, W, Cc), 121)
    w, b = rnd((128, Cc, 3, 3), 122, scale=(9 * Cc) ** -0.5), rnd((128,), 123, scale=0.1)
    y = ops.conv2d_nhwc(x, ops.pack_conv_weight(w), b, ksize=3, stride=2, pad=0, pad_after=1)
    ref = F.conv2d(F.pad(x.float().permute(0, 3, 1, 2), (0, 1, 0, 1)), w.float(), b.float(), stride=2).permute(0, 2, 3, 1)
    assert y.shape == ref.shape == (B, H // 2, W // 2, 128)
    assert_close_bf16(y, ref, "conv3x3 stride 2, pad (0,1,0,1)", rtol=8e-3, atol_rms=4e-3)


# ----------------------------------------------------------------------------------------------------------------------
# Boundary B4 / B3: attention backend function ((B, S, H, D) layout) and attention processor for the reference module
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,Sq,Skv,H,D", [(2, 200, 200, 4, 64), (1, 96, 77, 3, 128), (2, 64, 333, 2, 64)])
def test_attention_backend_bshd(B, Sq, Skv, H, D):
    from diffusers_amd.attention_backend import mi355x_flash_attention
    q, k, v = rnd((B, Sq, H, D), 131), rnd((B, Skv, H, D), 132), rnd((B, Skv, H, D), 133)
    o = mi355x_flash_attention(q, k, v)
    ref = F.scaled_dot_product_attention(q.float().cpu().transpose(1, 2), k.float().cpu().transpose(1, 2),
                                         v.float().cpu().transpose(1, 2)).transpose(1, 2)
    assert o.shape == (B, Sq, H, D)
    assert_close_bf16(o, ref, f"attention backend B{B} Sq{Sq} Skv{Skv} H{H} D{D}", rtol=1.6e-2, atol_rms=1.6e-2)
    with pytest.raises(ValueError):
        mi355x_flash_attention(q, k, v, is_causal=True)
    t_ = rnd((70, 130), 134)
    from diffusers_amd import ops
    assert torch.equal(ops.transpose(t_), t_.t().contiguous())


def test_attention_processor_on_reference_style_module():
    """MI355XAttnProcessor against a duck-typed stand-in of the reference Attention module (to_q/k/v/out as nn.Linear)."""
    import types
    from diffusers_amd.attention_backend import MI355XAttnProcessor
    torch.manual_seed(0)
    C, heads, cross = 128, 2, 64
    mk = lambda i, o, b: torch.nn.Linear(i, o, bias=b).to(DEV, bf16)  # noqa: E731
    attn = types.SimpleNamespace(heads=heads, to_q=mk(C, C, False), to_k=mk(cross, C, False), to_v=mk(cross, C, False),
                                 to_out=[mk(C, C, True)], scale=(C // heads) ** -0.5, residual_connection=False,
                                 rescale_output_factor=1.0, group_norm=None, spatial_norm=None, norm_cross=None)
    x, ctx = rnd((2, 96, C), 141), rnd((2, 77, cross), 142)
    with torch.no_grad():
        y = MI355XAttnProcessor()(attn, x, encoder_hidden_states=ctx)
        f = lambda m, t_: F.linear(t_.float().cpu(), m.weight.float().cpu(), None if m.bias is None else m.bias.float().cpu())  # noqa: E731
        q, k, v = f(attn.to_q, x), f(attn.to_k, ctx), f(attn.to_v, ctx)
        sp = lambda t_: t_.view(2, -1, heads, C // heads).transpose(1, 2)  # noqa: E731
        o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(2, 96, C)
        ref = F.linear(o, attn.to_out[0].weight.float().cpu(), attn.to_out[0].bias.float().cpu())
    assert_close_bf16(y, ref, "attention processor (cross-attention)", rtol=2e-2, atol_rms=2e-2)


# ----------------------------------------------------------------------------------------------------------------------
# UniPC (flow mode) fused step vs the reference trajectories
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,xd,vd", [("f32", torch.float32, torch.float32), ("mixed", torch.float32, bf16),
                                        ("bf16", bf16, bf16)])
def test_unipc_step_vs_reference(golden, name, xd, vd):
    """Bit-exact against the oracle in device-scalar mode (the reference as torch's device kernels evaluate it: 0-d CPU
    fp32 coefficients stay fp32, oracle/samplers.py header); against the frozen CPU run of the reference: bit-exact for
    fp32 latents, within the bf16 coefficient-rounding drift otherwise (rel. RMS 5e-3 mixed, 2e-2 bf16 over 7 steps)."""
    from diffusers_amd import schedulers as S
    from oracle import samplers as OS
    g = golden("unipc")
    n = len(g["timesteps"])
    sch = S.UniPCMultistepScheduler(prediction_type="flow_prediction", use_flow_sigmas=True, flow_shift=3.0)
    sch.set_timesteps(n, device=DEV)
    assert np.array_equal(sch.timesteps.cpu().numpy(), g["timesteps"])
    orc = OS.UniPCFlowOracle(flow_shift=3.0, device_scalars=True)
    orc.set_timesteps(n)
    x = torch.from_numpy(g["x0"]).to(xd).to(DEV)
    xo = torch.from_numpy(g["x0"]).to(xd)
    nbad, worst = 0, 0.0
    tol = {"f32": 0.0, "mixed": 5e-3, "bf16": 2e-2}[name]
    for i, t_ in enumerate(sch.timesteps):
        v = torch.from_numpy(g["v"][i]).to(vd)
        x = sch.step(v.to(DEV), t_, x).prev_sample
        xo = orc.step(v, xo)
        nbad += int((x.cpu() != xo).sum())
        want = torch.from_numpy(g[f"traj_{name}"][i])
        rel = float((x.float().cpu() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
        worst = max(worst, rel)
        assert rel <= tol, f"{name} step {i}: rel rms {rel:.3e} vs the frozen reference run"
    print(f"[parity] unipc {name}: elements differing from the device-scalar oracle over {n} steps = {nbad}; "
          f"worst rel rms vs frozen CPU reference run = {worst:.3e}")
    assert nbad == 0, f"UniPC {name}: {nbad} elements differ from the oracle trajectory"


def test_unipc_cfg_inplace_matches_separate_combine(golden):
    """step_cfg (CFG combine fused, in place) == combine in bf16 then step."""
    from diffusers_amd import schedulers as S
    from oracle import samplers as OS
    g = golden("unipc")
    n = len(g["timesteps"])
    a, b = (S.UniPCMultistepScheduler(prediction_type="flow_prediction", use_flow_sigmas=True, flow_shift=3.0) for _ in range(2))
    a.set_timesteps(n, device=DEV), b.set_timesteps(n, device=DEV)
    xa = torch.from_numpy(g["x0"]).to(DEV)
    xb = xa.clone()
    for i, t_ in enumerate(a.timesteps):
        u = torch.from_numpy(g["v"][i]).to(bf16)
        c = torch.from_numpy(g["v"][(i + 1) % n]).to(bf16)
        a.step_cfg(torch.cat([u, c]).to(DEV), xa, 5.0)
        xb = b.step(OS.cfg_combine(u, c, 5.0).to(DEV), t_, xb).prev_sample
        assert torch.equal(xa, xb), f"step {i}"
    y = rnd((3, 5, 7), 151, dtype=torch.float32)
    from diffusers_amd import ops
    assert torch.equal(ops.cast_f32_bf16(y, rep=2), torch.cat([y, y]).to(bf16))


# ----------------------------------------------------------------------------------------------------------------------
# AutoencoderKLWan kernels: channel RMS-norm (+SiLU), frame-pair permute, NCTHW clamp, and the
# in-place accumulating GEMM / conv launches the temporal taps of a causal Conv3d are made of
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,real,rows,silu", [(64, 24, 1000, True), (64, 48, 257, False), (128, 96, 4097, True),
                                              (192, 192, 333, True), (384, 384, 1024, False), (1024, 1000, 67, True)])
def test_rmsnorm_channels_vs_reference_ops(C, real, rows, silu):
    """WanRMS_norm with the reference's own rounding points (fp32 normalise -> bf16, * sqrt(C) -> bf16, * gamma -> bf16,
    SiLU -> bf16): at most one bf16 ulp from the same chain evaluated by torch."""
    ops, L = _ops()
    x = rnd((rows, C), 31 + C, scale=3.0)
    gm = (1.0 + 0.1 * rnd((C,), 32).float()).to(bf16)
    x[:, real:] = 0
    gm[real:] = 0
    y = ops.rmsnorm_channels(x, gm, real_channels=real, silu=silu)
    n = F.normalize(x.float()[:, :real], dim=1).to(bf16)
    ref = n * (real ** 0.5) * gm[:real]
    if silu:
        ref = F.silu(ref)
    assert y.shape == x.shape and y.dtype == bf16
    assert torch.equal(y[:, real:], torch.zeros_like(y[:, real:])), "zero-padded channels must stay zero"
    d = (y[:, :real].float() - ref.float()).abs()
    ulp = ref.float().abs() * 2.0 ** -7 + 1e-6
    nbad = int((d > ulp).sum())
    print(f"[parity] rmsnorm_channels C={C} real={real} silu={silu}: exact {float((d == 0).float().mean()):.4f}, >1ulp {nbad}")
    assert nbad == 0


def test_permute_0213_and_frames_to_ncthw():
    ops, L = _ops()
    x = rnd((5, 77, 2, 64), 41)
    assert torch.equal(ops.permute_0213(x), x.permute(0, 2, 1, 3).contiguous())
    buf = torch.zeros((1 + 10, 7, 11, 64), device=DEV, dtype=bf16)
    ops.permute_0213(x, out=buf[1:])
    assert torch.equal(buf[1:].reshape(5, 2, 77, 64), x.permute(0, 2, 1, 3)) and float(buf[0].abs().max()) == 0.0
    f = rnd((2 * 5, 6, 10, 4), 42, scale=0.8)
    for f32 in (False, True):
        v = ops.frames_to_ncthw(f, batch=2, channels=3, out_f32=f32)
        ref = f.float()[..., :3].clamp(-1, 1).view(2, 5, 6, 10, 3).permute(0, 4, 1, 2, 3)
        assert v.shape == (2, 3, 5, 6, 10) and v.dtype == (torch.float32 if f32 else bf16)
        assert torch.equal(v.float(), ref.contiguous())


def test_gemm_and_conv_accumulate_in_place():
    """out == residual: each launch adds its product to the buffer exactly once (also on the launch that tunes the
    shape), bit-identical to the out-of-place launch with the same residual."""
    ops, L = _ops()
    x = rnd((3, 12, 20, 64), 51)
    w = ops.pack_conv_weight(rnd((128, 64, 3, 3), 52, scale=(9 * 64) ** -0.5))
    base = rnd((3, 12, 20, 128), 53)
    want = ops.conv2d_nhwc(x, w, None, ksize=3, residual=base)
    acc = base.clone()
    got = ops.conv2d_nhwc(x, w, None, ksize=3, residual=acc, out=acc)
    assert got.data_ptr() == acc.data_ptr() and torch.equal(acc, want)
    # frame-shifted views of one buffer: the temporal-tap pattern
    acc2 = base.clone()
    ops.conv2d_nhwc(x[:2], w, None, ksize=3, residual=acc2[1:], out=acc2[1:])
    assert torch.equal(acc2[0], base[0]) and torch.equal(acc2[1:], ops.conv2d_nhwc(x[:2], w, None, ksize=3, residual=base[1:].contiguous()))
    xl, wl, bl = rnd((777, 128), 54), rnd((192, 128), 55, scale=128 ** -0.5), rnd((777, 192), 56)
    wantl = ops.linear(xl, wl, residual=bl)
    accl = bl.clone()
    ops.linear(xl, wl, residual=accl, out=accl)
    assert torch.equal(accl, wantl)
    # narrow output (conv_out: 3 channels padded to 4)
    w4 = ops.pack_conv_weight(rnd((4, 64, 3, 3), 57, scale=(9 * 64) ** -0.5))
    y4 = ops.conv2d_nhwc(x, w4, rnd((4,), 58), ksize=3)
    ref4 = F.conv2d(x.float().permute(0, 3, 1, 2), w4.float().view(4, 3, 3, 64).permute(0, 3, 1, 2), rnd((4,), 58).float(),
                    padding=1).permute(0, 2, 3, 1)
    assert_close_bf16(y4, ref4, "conv3x3 N=4", rtol=8e-3, atol_rms=4e-3)


@pytest.mark.parametrize("dtype", [bf16, torch.float32])
def test_image_postprocess_matches_reference_chain(dtype):
    """VaeImageProcessor.postprocess: denormalize -> (channels last) -> (x * 255).round() uint8, bit for bit against the
    same chain in torch / numpy (fp32 arithmetic on the decoded values; numpy rounds half to even)."""
    ops, L = _ops()
    x = rnd((2, 3, 37, 53), 71, scale=0.8, dtype=dtype)
    x[0, 0, 0, :8] = torch.tensor([-1.0, 1.0, 0.0, 1.5, -1.5, 1 / 255 - 1, 2 / 255 - 1, 0.00390625], dtype=dtype)
    want = (x.float() * 0.5 + 0.5).clamp(0, 1)
    assert torch.equal(ops.image_postprocess(x, "pt"), want)
    nhwc = want.permute(0, 2, 3, 1).contiguous()
    assert torch.equal(ops.image_postprocess(x, "np"), nhwc)
    u8 = ops.image_postprocess(x, "uint8")
    assert u8.dtype == torch.uint8 and np.array_equal(u8.cpu().numpy(), (nhwc.cpu().numpy() * 255).round().astype("uint8"))
    v = rnd((1, 3, 4, 6, 10), 72, dtype=dtype)                       # video: [B][C][T][H][W] -> [B][T][H][W][C]
    assert torch.equal(ops.image_postprocess(v, "np"), (v.float() * 0.5 + 0.5).clamp(0, 1).permute(0, 2, 3, 4, 1).contiguous())
    with pytest.raises(ValueError):
        ops.image_postprocess(x, "jpeg")


def test_conv_out_postprocess_epilogue_equals_the_separate_pass():
    """`conv_thin_out(..., postprocess=)` (the VAE decoder's conv_out with VaeImageProcessor.postprocess fused into its layout
    pass, image_processor.py:738-786) == conv_thin_out followed by image_postprocess, bit for bit, on the implicit-GEMM route
    (128 -> 3 at 96 x 80) and on the small-image fallback (32 -> 3 at 16 x 16)."""
    ops, L = _ops()
    for (ci, h, w_, seed) in ((128, 96, 80, 81), (32, 16, 16, 83)):
        x = rnd((2, h, w_, ci), seed, scale=2.0)
        wpk = ops.pack_conv_weight(rnd((3, ci, 3, 3), seed + 1, scale=(9 * ci) ** -0.5))
        b = rnd((3,), seed + 2)
        base = ops.conv_thin_out(x, wpk, b)
        assert base.dtype == bf16 and float(base.abs().max()) > 1.0          # some values clamp
        for mode in ("pt", "np", "uint8"):
            got = ops.conv_thin_out(x, wpk, b, postprocess=mode)
            want = ops.image_postprocess(base, mode)
            assert got.dtype == want.dtype and got.shape == want.shape and torch.equal(got, want), (ci, mode)
    with pytest.raises(ValueError):
        ops.conv_thin_out(x, wpk, b, postprocess="pil")


def test_torch_library_ops_run_the_kernels_and_trace():
    """torch.ops.mi355x.* == the ctypes entry points bit for bit, and a function written over them is traceable by
    torch.compile (aot_eager: the tracer needs only the fake kernels; no inductor / Triton code generation involved)."""
    import diffusers_amd.torch_ops  # noqa: F401  (registers the ops)
    ops, L = _ops()
    ns = torch.ops.mi355x
    x, w, b = rnd((256, 320), 1), rnd((640, 320), 2, 320 ** -0.5), rnd((640,), 3)
    assert torch.equal(ns.gemm(x, w, b, L.ACT_SILU), ops.linear(x, w, b, act=L.ACT_SILU))
    xi, wc = rnd((2, 16, 16, 64), 4), rnd((128, 9 * 64), 5, (9 * 64) ** -0.5)
    assert torch.equal(ns.conv2d_nhwc(xi, wc, None, 3, 1, True), ops.conv2d_nhwc(xi, wc, None, ksize=3, up=True))
    g_, be = rnd((64,), 6), rnd((64,), 7)
    assert torch.equal(ns.groupnorm(xi, g_, be, 32, 1e-5, True), ops.group_norm_nhwc(xi, g_, be, 32, 1e-5, silu=True))
    assert torch.equal(ns.layernorm(x, rnd((320,), 8), rnd((320,), 9), 1e-5), ops.layer_norm(x, rnd((320,), 8), rnd((320,), 9), 1e-5))
    q, k, v = rnd((2, 200, 4, 64), 10), rnd((2, 200, 4, 64), 11), rnd((2, 200, 4, 64), 12)
    o = ns.flash_attn(q, k, v, None)
    ref = F.scaled_dot_product_attention(q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2))
    assert_close_bf16(o, ref.transpose(1, 2), "torch.ops.mi355x.flash_attn")

    def block(x, w, b):
        h = torch.ops.mi355x.layernorm(x, None, None, 1e-5)
        return torch.ops.mi355x.gemm(h, w, b, L.ACT_SILU) * 2.0
    want = block(x, w, b)
    got = torch.compile(block, backend="aot_eager", fullgraph=True)(x, w, b)
    assert torch.equal(got, want)


# ----------------------------------------------------------------------------------------------------------------------
# kernels behind the text encoders (diffusers_amd/text_encoders.py)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,S,causal,with_bias,mask", [(2, 2, 77, True, False, False), (1, 12, 77, True, False, False),
                                                         (2, 4, 512, False, True, False), (2, 3, 200, False, True, True),
                                                         (1, 2, 130, True, True, False)])
def test_masked_flash_attention(B, H, S, causal, with_bias, mask):
    """The MASKED variant (D = 64): causal (CLIP), additive bias with scale 1 (T5 relative position bias), key padding
    folded into the bias (UMT5 / Wan), and both at once -- vs fp32 softmax(scale q k^T + bias + mask) v."""
    ops, L = _ops()
    D, inner = 64, H * 64
    g = torch.Generator("cpu").manual_seed(S + H)
    q, k, v = ((torch.randn((B, S, inner), generator=g) * 0.7).to(bf16) for _ in range(3))
    sa = ((S + 15) // 16) * 16
    pad = lambda t: torch.cat([t, torch.zeros((B, sa - S, inner), dtype=bf16)], 1).reshape(B * sa, inner).to(DEV)  # noqa: E731
    qp, kp, vp = pad(q), pad(k), pad(v)
    vt = vp.view(B, sa, inner).permute(2, 0, 1).reshape(inner, B * sa).contiguous()
    scale = 1.0 if with_bias else D ** -0.5
    bias = None
    sc = torch.einsum("bqhd,bkhd->bhqk", q.float().view(B, S, H, D), k.float().view(B, S, H, D)) * scale
    if with_bias:
        ld = ((S + 63) // 64) * 64
        bz = torch.zeros((B if mask else 1, H, S, ld))
        bz[..., :S] = torch.randn((1, H, S, S), generator=g) * 0.5
        if mask:
            keep = torch.ones((B, S), dtype=torch.bool)
            keep[0, S // 2:] = False
            keep[1, S - 7:] = False
            bz[..., :S].masked_fill_(~keep[:, None, None, :], -1e30)
        bias = (bz.to(bf16) if S == 512 else bz).to(DEV)          # both element types of the C ABI
        bb = bias.float().cpu()[..., :S]
        sc = torch.where(bb <= -1e29, torch.full_like(sc, float("-inf")), sc + bb)
    if causal:
        sc = sc.masked_fill(torch.arange(S)[None, :] > torch.arange(S)[:, None], float("-inf"))
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(sc, -1), v.float().view(B, S, H, D)).reshape(B, S, inner)
    for b in range(B):        # padded batches: one launch per batch, as text_encoders._SelfAttention does
        o = torch.zeros((S, inner), device=DEV, dtype=bf16)
        ops.attention(qp[b * sa:], kp[b * sa:], vt[:, b * sa:], B=1, H=H, D=D, Sq=S, Skv=S, Skv_alloc=sa, q_row_stride=inner,
                      k_row_stride=inner, q_batch_stride=sa * inner, k_batch_stride=sa * inner, vt_ld=B * sa, vt_batch_stride=sa,
                      scale=scale, causal=causal, bias=None if bias is None else (bias[b:b + 1] if bias.shape[0] > 1 else bias),
                      out=o)
        assert_close_bf16(o, ref[b], f"masked attention B{B} H{H} S{S} causal={causal} bias={with_bias} mask={mask} [batch {b}]",
                          rel_rms_max=6e-3)


@pytest.mark.parametrize("B,H,Sq,Skv", [(2, 5, 256, 77), (2, 2, 1000, 77), (1, 3, 64, 130)])
def test_cross_attention_key_padding_mask_shared_by_all_queries(B, H, Sq, Skv):
    """bias_row_stride = 0: ONE bias row per batch for every head and query -- UNet2DConditionModel's encoder_attention_mask
    (unet_2d_condition.py:1071-1073: 0 keep, -10000 discard, added to the scaled scores) -- vs fp32 SDPA with that mask."""
    ops, L = _ops()
    from diffusers_amd.layers import encoder_mask_bias
    D, inner = 64, H * 64
    q, k, v = rnd((B, Sq, inner), 71, 0.7), rnd((B, Skv, inner), 72, 0.7), rnd((B, Skv, inner), 73)
    keep = torch.ones((B, Skv), dtype=torch.long, device=DEV)
    keep[0, Skv // 2:] = 0
    keep[-1, Skv - 3:] = 0
    bias = encoder_mask_bias(keep, B, Skv)
    assert bias.shape == (B, 1, 1, (Skv + 63) // 64 * 64)
    sa = ((Skv + 15) // 16) * 16
    kp = torch.zeros((B, sa, inner), device=DEV, dtype=bf16)
    kp[:, :Skv] = k
    vt = torch.zeros((inner, B * sa), device=DEV, dtype=bf16)
    vt.view(inner, B, sa)[:, :, :Skv] = v.permute(2, 0, 1)
    o = ops.attention(q.view(B * Sq, inner), kp.view(B * sa, inner), vt, B=B, H=H, D=D, Sq=Sq, Skv=Skv, Skv_alloc=sa,
                      q_row_stride=inner, k_row_stride=inner, q_batch_stride=Sq * inner, k_batch_stride=sa * inner,
                      vt_ld=B * sa, vt_batch_stride=sa, bias=bias)
    qh, kh, vh = (t.float().cpu().view(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    add = ((1 - keep.float().cpu()) * -10000.0)[:, None, None, :]
    ref = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=add).transpose(1, 2).reshape(B * Sq, inner)
    assert_close_bf16(o, ref, f"cross attention with a key-padding mask B{B} H{H} Sq{Sq} Skv{Skv}", rel_rms_max=6e-3)
    unmasked = ops.attention(q.view(B * Sq, inner), kp.view(B * sa, inner), vt, B=B, H=H, D=D, Sq=Sq, Skv=Skv, Skv_alloc=sa,
                             q_row_stride=inner, k_row_stride=inner, q_batch_stride=Sq * inner, k_batch_stride=sa * inner,
                             vt_ld=B * sa, vt_batch_stride=sa)
    assert not torch.equal(o, unmasked)


def test_text_encoder_epilogues_and_rmsnorm():
    """DA_ACT_QUICK_GELU (CLIP-L MLP), DA_ACT_GEGLU_TANH (T5 gated-GELU feed-forward) and da_rmsnorm_bf16 (T5LayerNorm) vs the
    torch ops of the transformers modules they replace."""
    ops, L = _ops()
    M, K, N = 154, 768, 3072
    x, w, b = rnd((M, K), 71), rnd((N, K), 72, K ** -0.5), rnd((N,), 73)
    y = ops.linear(x, w, b, act=L.ACT_QUICK_GELU)
    pre = (x.float() @ w.float().t() + b.float()).to(bf16).float()
    assert_close_bf16(y, pre * torch.sigmoid(1.702 * pre), "fc1 + quick_gelu")
    wi0, wi1 = rnd((1024, K), 74, K ** -0.5), rnd((1024, K), 75, K ** -0.5)
    wp, _ = ops.pack_geglu(torch.cat([wi1, wi0], 0), None)
    for tile in (L.TILE_128x128, L.TILE_64x128, L.TILE_256x128):
        yg = ops.linear(x, wp, act=L.ACT_GEGLU_TANH, tile=tile, staging=1)
        want = F.gelu(x.float() @ wi0.float().t(), approximate="tanh") * (x.float() @ wi1.float().t())
        assert_close_bf16(yg, want, f"gated tanh-GELU, tile {tile}", rtol=2.5e-2, atol_rms=2.5e-2)
    for C in (768, 4096, 128):
        xr, gm = rnd((77, C), 76, 3.0), rnd((C,), 77) * 0.2 + 1.0
        got = ops.rms_norm(xr, gm, 1e-6)
        xf = xr.float()
        want = gm.float() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(bf16).float()
        assert torch.equal(got, want.to(bf16)) or float((got.float() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max())


def test_cfg_rescale_matches_the_reference_bit_for_bit(golden):
    """da_cfg_rescale (CFG combine + rescale_noise_cfg, pipeline_stable_diffusion.py:69-92, :1054-1059) against the live
    reference's outputs: bf16 bit-exact (every torch op's rounding reproduced; the per-sample std is reduced in fp64 and
    rounded to bf16 as torch rounds its fp32 reduction), fp32 to reduction-order accuracy."""
    ops, L = _ops()
    g = golden("guidance_rescale")
    for case in "abc":
        for dt_name, dt in (("bf16", bf16), ("f32", torch.float32)):
            u = torch.from_numpy(g[f"{case}_{dt_name}_uncond"]).to(dt).to(DEV)
            c = torch.from_numpy(g[f"{case}_{dt_name}_cond"]).to(dt).to(DEV)
            want = torch.from_numpy(g[f"{case}_{dt_name}_out"])
            gs, gr = (float(v) for v in g[f"{case}_{dt_name}_params"])
            y = ops.cfg_rescale(torch.cat([u, c]).contiguous(), gs, gr)
            assert y.shape == u.shape and y.dtype == dt
            if dt == bf16:
                bad = int((y.float().cpu() != want).sum())
                assert bad == 0, f"case {case}: {bad} of {want.numel()} bf16 outputs differ from the reference"
            else:
                assert torch.allclose(y.cpu(), want, rtol=2e-5, atol=2e-6), f"case {case} fp32"
    with pytest.raises(ValueError):
        ops.cfg_rescale(torch.zeros((3, 4, 8, 8), device=DEV, dtype=bf16), 5.0, 0.5)     # odd batch: not (uncond, cond)
