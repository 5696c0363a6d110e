"""GPU tests of the second GEMM / implicit-GEMM family (csrc/gemm2_kernel.cuh, DA_TILE_K2_*): 8 waves = 2 K-groups x 4 waves on
alternate K slices, 16x16x32 MFMA, slice pairs with a mid-pair rendezvous, partial sums exchanged through LDS.

What must hold: (1) right against a plain PyTorch fp32 reference of the same op, with the tolerance of the first family's
tests; (2) every K2 (tile, ring depth) variant BIT-identical to every other one (same per-element summation: (even K
slices) + (odd K slices)); (3) within one bf16 ulp of the first family's result (fp32 summation order is the only
difference); (4) every fused epilogue of da_gemm_params; (5) the implicit-GEMM gather in every conv form the U-Net / VAE use
(3x3, 1x1, stride 2, fused nearest-2x, two-source channel concat), including odd K slice counts and ragged M / N edges."""
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


def k2_variants(L, conv=False, geglu=False):
    out = []
    for t in range(L.FIRST_K2_TILE, len(L.TILE_NAMES)):
        if conv and t in (L.TILE_K2_80x128, L.TILE_K1_256x256, L.TILE_K1_256x320):
            continue
        if geglu and t not in (L.TILE_K2_128x128, L.TILE_K1_256x128, L.TILE_K1_128x256, L.TILE_K1_256x256, L.TILE_K1_256x320, L.TILE_K1_128x320):
            continue
        for st in (L.STAGE_LDS_DIRECT, L.STAGE_LDS_DIRECT3, L.STAGE_PINGPONG, L.STAGE_PINGPONG3):
            out.append((t, st))
    return out


def run_all(fn, L, what, conv=False, geglu=False, min_ok=5):
    """fn(tile, staging) for every variant of gemm2_kernel.cuh; unsupported (tile, ring depth) pairs must say so.  Variants of
    one summation order are BIT-identical: the KG = 2 tiles (k2:*, (even K slices) + (odd K slices)) among themselves, the
    KG = 1 tiles (k1:*, all slices in order) among themselves; the two orders agree within one bf16 ulp.  Returns the k2 result."""
    base, n_ok = {}, 0
    for t, st in k2_variants(L, conv, geglu):
        try:
            y = fn(t, st)
        except RuntimeError as e:
            assert "DA_ERR_UNSUPPORTED" in str(e), f"{what} {L.TILE_NAMES[t]}/{st}: {e}"
            continue
        n_ok += 1
        fam = L.TILE_NAMES[t][:2].replace("k3", "k1")   # the eight-phase tile (gemm3.hip) sums K in the k1 order
        if fam not in base:
            base[fam] = y.clone()
        else:
            assert torch.equal(y, base[fam]), f"{what}: {L.TILE_NAMES[t]}/{st} differs from the first {fam} variant"
    assert n_ok >= min_ok, f"{what}: only {n_ok} variants ran"
    if "k1" in base and "k2" in base:
        if base["k2"].dtype == torch.float32:
            assert torch.allclose(base["k1"], base["k2"], rtol=1e-2, atol=1e-2 * float(base["k2"].abs().max()))
        else:
            one_ulp(base["k1"], base["k2"], f"{what} (k1 vs k2 summation order)", ulps=2 if geglu else 1.25)
    return base.get("k2", base.get("k1"))


def one_ulp(y, base, what, ulps=1):
    """Within one bf16 ulp of the first family's result.  The ulp is that of the LARGER of the output and the tensor's rms:
    an epilogue that adds a residual can cancel (|out| << |gemm term|), and a last-bit difference of the rounded GEMM term is
    then many ulps of the tiny sum -- it is still one ulp of what was rounded."""
    bf = base.float()
    scale = torch.maximum(bf.abs(), bf.pow(2).mean().sqrt())
    d = (y.float() - bf).abs() / scale
    frac = float((y != base).float().mean())
    print(f"[parity] {what}: {100 * frac:.2f}% of outputs differ from the first family by <= {float(d.max()):.2e} relative")
    assert float(d.max()) <= ulps * 2.0 ** -7, f"{what}: more than {ulps} bf16 ulp from the first kernel family"


# (M, N, K): SDXL's projections, odd slice counts (K = 64, 192, 320), ragged M / N edges, a single-tile problem
@pytest.mark.parametrize("M,N,K", [(2048, 1280, 1280), (8192, 640, 640), (300, 320, 192), (520, 132, 64), (777, 644, 1152),
                                   (154, 1280, 2048), (96, 80, 320), (1280, 2048, 1280)])
def test_k2_gemm_variants(M, N, K):
    ops, L = _ops()
    x, w, b, r = rnd((M, K), 21), rnd((N, K), 22, K ** -0.5), rnd((N,), 23), rnd((M, N), 24)
    ref = x.float() @ w.float().t() + b.float() + r.float()
    y = run_all(lambda t, st: ops.linear(x, w, b, residual=r, tile=t, staging=st), L, f"k2 gemm {M}x{N}x{K}", min_ok=8)
    assert_close_bf16(y, ref, f"k2 gemm {M}x{N}x{K}", rtol=8e-3, atol_rms=4e-3)
    one_ulp(y, ops.linear(x, w, b, residual=r, tile=L.TILE_128x128, staging=L.STAGE_LDS_DIRECT), f"k2 gemm {M}x{N}x{K}")
    # repeated launches: the ring / exchange area carry nothing from one launch to the next
    for _ in range(3):
        assert torch.equal(ops.linear(x, w, b, residual=r, tile=L.TILE_K2_128x80, staging=L.STAGE_LDS_DIRECT3), y)


def test_k2_gemm_strided_operands_and_outputs():
    """lda / ldw / ldc / ldr other than the row length: the Q|K layout ([M][2C] viewed as two [M][C] halves)."""
    ops, L = _ops()
    M, N, K = 1024, 640, 640
    xx, ww = rnd((M, 2 * K), 1), rnd((N, 2 * K), 2, K ** -0.5)
    out = torch.zeros((M, 2 * N), device=DEV, dtype=bf16)
    res = rnd((M, 2 * N), 3)
    x, w, r = xx[:, K:], ww[:, :K], res[:, N:]
    ref = x.float() @ w.float().t() + r.float()
    for t, st in k2_variants(L):
        out.zero_()
        try:
            ops.linear(x, w, residual=r, out=out[:, :N], tile=t, staging=st)
        except RuntimeError as e:
            assert "DA_ERR_UNSUPPORTED" in str(e)
            continue
        assert_close_bf16(out[:, :N], ref, f"k2 strided {L.TILE_NAMES[t]}/{st}", rtol=8e-3, atol_rms=4e-3)
        assert float(out[:, N:].abs().max()) == 0.0, "wrote outside its column block"


@pytest.mark.parametrize("act", ["none", "silu", "gelu_tanh", "gelu_erf", "quick_gelu"])
def test_k2_gemm_epilogues_match_first_family(act):
    """bias + per-batch channel vector + activation + residual + output scale, bias_rows, gate (bf16 and fp32), fp32 output:
    the K2 epilogue restates gemm_kernel.cuh's with the same rounding points -- within one bf16 ulp of it everywhere."""
    ops, L = _ops()
    A = {"none": L.ACT_NONE, "silu": L.ACT_SILU, "gelu_tanh": L.ACT_GELU_TANH, "gelu_erf": L.ACT_GELU_ERF,
         "quick_gelu": L.ACT_QUICK_GELU}[act]
    M, N, K, B = 1536, 640, 448, 3
    x, w, b = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5), rnd((N,), 3)
    rv, res, br = rnd((B, N), 4), rnd((M, N), 5), rnd((M,), 6)
    gate_b, gate_f = rnd((B, N), 7), rnd((B, N), 8, dtype=torch.float32)
    cases = {
        "bias+rowvec+act+res+scale": dict(bias=b, rowvec=rv, rows_per_batch=M // B, act=A, residual=res, out_scale=0.5),
        "bias_rows+alpha": dict(bias_rows=br, alpha=0.125, act=A),
        "gate bf16 + res": dict(bias=b, gate=gate_b, rows_per_batch=M // B, residual=res, act=A),
        "gate fp32 + res": dict(bias=b, gate=gate_f, rows_per_batch=M // B, residual=res, act=A),
        "fp32 out": dict(bias=b, out_f32=True, alpha=0.25, act=A),
    }
    for name, kw in cases.items():
        want = ops.linear(x, w, tile=L.TILE_128x128, staging=L.STAGE_LDS_DIRECT, **kw)
        got = run_all(lambda t, st: ops.linear(x, w, tile=t, staging=st, **kw), L, f"k2 epilogue {act} {name}", min_ok=8)
        assert got.dtype == want.dtype
        if want.dtype == torch.float32 and act == "none":
            assert torch.allclose(got, want, rtol=2e-5, atol=2e-5 * float(want.abs().max()))
        else:
            # an activation is applied to the bf16-ROUNDED pre-activation (as the reference's separate op sees it): a last-bit
            # difference there moves the activation's output by up to its slope (< 1.13 for these) times one ulp
            one_ulp(got, want, f"k2 epilogue {act} / {name}", ulps=1 if act == "none" else 1.25)


@pytest.mark.parametrize("M,N2,K", [(2048, 10240, 1280), (300, 256, 192), (8192, 5120, 640)])
def test_k2_geglu(M, N2, K):
    """GEGLU fused into the up projection (activations.py:113-124): packed [32 value | 32 gate] rows, K2 128x128 tile."""
    ops, L = _ops()
    x, w, b = rnd((M, K), 1), rnd((N2, K), 2, K ** -0.5), rnd((N2,), 3)
    wp, bp = ops.pack_geglu(w, b)
    h = (x.float() @ w.float().t() + b.float()).to(bf16).float()
    ref = h[:, : N2 // 2] * F.gelu(h[:, N2 // 2:]).to(bf16).float()
    y = run_all(lambda t, st: ops.linear(x, wp, bias=bp, act=L.ACT_GEGLU, tile=t, staging=st), L, f"k2 geglu {M}x{N2}x{K}",
                geglu=True, min_ok=1)
    assert y.shape == (M, N2 // 2)
    assert_close_bf16(y, ref, f"k2 geglu {M}x{N2}x{K}", rtol=1.6e-2, atol_rms=6e-3)
    base = ops.linear(x, wp, bias=bp, act=L.ACT_GEGLU, tile=L.TILE_128x128, staging=L.STAGE_LDS_DIRECT)
    # the product of two bf16-rounded factors: a last-bit difference in either moves the product by up to two ulps
    one_ulp(y, base, f"k2 geglu {M}x{N2}x{K}", ulps=2)
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        ops.linear(x, wp, bias=bp, act=L.ACT_GEGLU, tile=L.TILE_K2_128x80, staging=L.STAGE_LDS_DIRECT)


def _conv_ref(x, x2, w4, b, stride, up, ksize, rv=None, res=None):
    xin = x if x2 is None else torch.cat([x, x2], dim=-1)
    xn = xin.float().permute(0, 3, 1, 2)
    if up:
        xn = F.interpolate(xn, scale_factor=2.0, mode="nearest")
    y = F.conv2d(xn, w4.float(), b.float(), stride=stride, padding=(ksize - 1) // 2).permute(0, 2, 3, 1)
    if rv is not None:
        y = y + rv.float()[:, None, None, :]
    if res is not None:
        y = y + res.float()
    return y


# B, H, W, C1, C2, Cout, ksize, stride, up  (K slices per tap: 1 .. 30; odd totals; concat boundaries inside and between pairs)
@pytest.mark.parametrize("B,H,W,C1,C2,Co,ks,stride,up", [
    (2, 32, 32, 1280, 0, 1280, 3, 1, False),     # SDXL 32x32 level
    (2, 16, 16, 640, 320, 640, 3, 1, False),     # skip concat, 15 slices per tap
    (2, 24, 24, 320, 0, 320, 3, 1, False),       # 5 slices per tap: pairs straddle taps, 45 slices (odd)
    (1, 20, 20, 64, 0, 96, 3, 1, False),         # 1 slice per tap, 9 slices, ragged N
    (1, 16, 16, 320, 0, 320, 3, 2, False),       # Downsample2D
    (1, 12, 12, 256, 0, 128, 3, 1, True),        # Upsample2D: nearest-2x in the gather
    (2, 16, 16, 640, 0, 1280, 1, 1, False),      # 1x1 shortcut
    (1, 16, 16, 64, 64, 128, 1, 1, False),       # 1x1 over a concat, 2 slices
    (1, 9, 7, 128, 192, 80, 3, 1, False),        # ragged M (63 pixels), concat with unequal halves
])
def test_k2_conv_variants(B, H, W, C1, C2, Co, ks, stride, up):
    ops, L = _ops()
    x = rnd((B, H, W, C1), 41)
    x2 = rnd((B, H, W, C2), 42) if C2 else None
    Ct = C1 + C2
    w4 = rnd((Co, Ct, ks, ks), 43, (ks * ks * Ct) ** -0.5)
    w, b = ops.pack_conv_weight(w4), rnd((Co,), 44)
    Ho, Wo = ((2 * H if up else H) + stride - 1) // stride, ((2 * W if up else W) + stride - 1) // stride
    rv, res = rnd((B, Co), 45), rnd((B, Ho, Wo, Co), 46)
    kw = dict(ksize=ks, x2=x2, stride=stride, up=up, rowvec=rv, residual=res)
    ref = _conv_ref(x, x2, w4, b, stride, up, ks, rv, res)
    what = f"k2 conv{ks} {B}x{H}x{W} {C1}+{C2}->{Co} s{stride} u{int(up)}"
    y = run_all(lambda t, st: ops.conv2d_nhwc(x, w, b, tile=t, staging=st, **kw), L, what, conv=True, min_ok=6)
    assert tuple(y.shape) == tuple(ref.shape)
    assert_close_bf16(y, ref, what, rtol=8e-3, atol_rms=4e-3)
    one_ulp(y, ops.conv2d_nhwc(x, w, b, tile=L.TILE_128x128, staging=L.STAGE_LDS_DIRECT, **kw), what)


def test_k2_conv_in_place_accumulation():
    """residual == out (the temporal taps of a causal Conv3d accumulate in place): each element is read and written by one lane."""
    ops, L = _ops()
    B, H, W, C, Co = 3, 16, 16, 128, 128
    x, w4, b = rnd((B, H, W, C), 1), rnd((Co, C, 3, 3), 2, (9 * C) ** -0.5), rnd((Co,), 3)
    w = ops.pack_conv_weight(w4)
    acc0 = rnd((B, H, W, Co), 4)
    want = ops.conv2d_nhwc(x, w, b, ksize=3, residual=acc0, tile=L.TILE_128x128, staging=L.STAGE_LDS_DIRECT)
    for t in (L.TILE_K2_128x128, L.TILE_K2_128x64):
        acc = acc0.clone()
        ops.conv2d_nhwc(x, w, b, ksize=3, residual=acc, out=acc, tile=t, staging=L.STAGE_LDS_DIRECT)
        one_ulp(acc, want, f"k2 in-place conv {L.TILE_NAMES[t]}")


def test_k2_refuses_what_it_does_not_implement():
    ops, L = _ops()
    x, w = rnd((256, 256), 1), rnd((256, 256), 2)
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        ops.linear(x, w, tile=L.TILE_K2_128x128, staging=L.STAGE_LDS_DIRECT3)      # 3 pairs of 64 KiB do not fit
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        ops.linear(x, w, tile=L.TILE_K2_128x80, staging=L.STAGE_REGISTER)
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        ops.linear(x, w, tile=L.TILE_K2_128x80, staging=L.STAGE_LDS_DIRECT, split_k=2)


@pytest.mark.parametrize("M,C,N", [(2048, 1280, 1280), (520, 320, 400), (8192, 640, 640)])
def test_k2_layernorm_fold_producer_and_consumers(M, C, N):
    """LayerNorm fold on the second kernel family (round 4: da_gemm_params.stats_out / ln_* in gemm2_kernel.cuh, LNF instantiations).
      * PRODUCER (k2:128x80 / k2:128x160, every ring): one (sum, sum of squares) pair per row and 80-column band of the bf16 output;
        their sum equals the row statistics of the stored tensor, the output is bit-identical to the launch without statistics,
        da_gemm_stats_parts() = ceil(N / 80) pairs are written and nothing beyond them;
      * CONSUMER: linear(x, W', ln=...) == linear(layer_norm(x), W) within the bf16 tolerance of one GEMM, plain epilogue
        (k2:128x80 / k2:128x160, + bias) and GEGLU epilogue (k1:128x320, interleaved ownership); rows with a large
        mean (|mu| = 8 sigma) and ragged M / N edges included;
      * statistics from the first family's producers feed the second family's consumers and vice versa."""
    ops, L = _ops()
    a, wprod, res = rnd((M, 192), 61), rnd((C, 192), 62, 192 ** -0.5), rnd((M, C), 63)
    res = res + 8.0 * (torch.arange(M, device=DEV) % 3 == 0).to(bf16)[:, None]
    gamma, beta = rnd((C,), 64) * 0.3 + 1.0, rnd((C,), 65) * 0.2
    w, b = rnd((N, C), 66, C ** -0.5), rnd((N,), 67)
    w1 = rnd((2 * max(N // 128, 1) * 128, C), 68, C ** -0.5)
    b1 = rnd((w1.shape[0],), 69)
    wl, fold = ops.fold_layernorm(w, gamma, beta, 1e-5)
    w1l, fold1 = ops.fold_layernorm(w1, gamma, beta, 1e-5)
    w1p, b1p = ops.pack_geglu(w1l, b1)
    n2 = w1.shape[0] // 2
    idx = torch.arange(n2, device=DEV).view(n2 // 32, 32)
    order = torch.cat([idx, idx + n2], dim=1).reshape(-1)
    fold1p = ops.LNFold(fold1.s[order].contiguous(), fold1.c[order].contiguous(), fold1.eps)
    ref_x = None
    prods = [(L.TILE_K2_128x80, s) for s in (L.STAGE_LDS_DIRECT, L.STAGE_LDS_DIRECT3, L.STAGE_PINGPONG, L.STAGE_PINGPONG3)] + \
            [(L.TILE_K2_128x160, s) for s in (L.STAGE_LDS_DIRECT, L.STAGE_PINGPONG)]
    cons = prods
    gcons = [(L.TILE_K1_128x320, L.STAGE_LDS_DIRECT)]
    stats = []
    for tile, stg in prods:
        x_plain = ops.linear(a, wprod, residual=res, tile=tile, staging=stg)
        st = ops.RowStats(M, DEV)
        st.buf.fill_(float("nan"))
        x = ops.linear(a, wprod, residual=res, tile=tile, staging=stg, stats_out=st)
        what = f"K2 LN fold producer {L.TILE_NAMES[tile]}/{stg} M{M} C{C}"
        assert torch.equal(x, x_plain), f"{what}: statistics changed the output"
        assert st.parts == (C + 79) // 80, f"{what}: {st.parts} partials"
        xf = x.float()
        tot = st.buf[:, :st.parts].sum(dim=1)
        assert torch.isfinite(tot).all() and torch.isnan(st.buf[:, st.parts:]).all(), f"{what}: wrong number of partials"
        assert torch.allclose(tot[:, 0], xf.sum(dim=1), rtol=1e-5, atol=1e-2), what
        assert torch.allclose(tot[:, 1], (xf * xf).sum(dim=1), rtol=1e-5, atol=1e-2), what
        if ref_x is None:
            ref_x = x
        assert torch.equal(x, ref_x)                      # every K2 variant: the same bits
        stats.append(st)
    xf = ref_x.float()
    ln_ref = F.layer_norm(xf, (C,), gamma.float(), beta.float(), 1e-5)
    ref = ln_ref @ w.float().t() + b.float()
    g = ln_ref @ w1.float().t() + b1.float()
    h_, g_ = g.chunk(2, dim=-1)
    ref_geglu = h_ * F.gelu(g_)
    # a first-family producer's statistics of the same tensor (its output differs from the K2 one in last bits only: use the
    # K2 tensor with K2 statistics, and the first-family tensor with its own)
    st1 = ops.RowStats(M, DEV)
    x1 = ops.linear(a, wprod, residual=res, tile=L.TILE_128x128, staging=1, stats_out=st1)
    base = None
    for ctile, cstg in cons:
        for st in (stats[0], stats[-1]):
            y = ops.linear(ref_x, wl, b, tile=ctile, staging=cstg, ln=(st, fold))
            assert_close_bf16(y, ref, f"K2 LN fold consumer {L.TILE_NAMES[ctile]}/{cstg} M{M} C{C} N{N} parts {st.parts}", rel_rms_max=6e-3)
            if base is None:
                base = y
            assert torch.equal(y, base), "K2 consumers: every (tile, ring, statistics source) must give the same bits"
    y1 = ops.linear(x1, wl, b, tile=cons[0][0], staging=cons[0][1], ln=(st1, fold))       # first-family statistics -> K2 consumer
    ref1 = F.layer_norm(x1.float(), (C,), gamma.float(), beta.float(), 1e-5) @ w.float().t() + b.float()
    assert_close_bf16(y1, ref1, "first-family statistics -> K2 consumer", rel_rms_max=6e-3)
    y2 = ops.linear(ref_x, wl, b, tile=L.TILE_128x64, staging=L.STAGE_LDS_DIRECT3, ln=(stats[0], fold))   # K2 statistics -> first family
    assert_close_bf16(y2, ref, "K2 statistics -> first-family consumer", rel_rms_max=6e-3)
    for ctile, cstg in gcons:
        yg = ops.linear(ref_x, w1p, b1p, act=L.ACT_GEGLU, tile=ctile, staging=cstg, ln=(stats[0], fold1p))
        assert_close_bf16(yg, ref_geglu, f"K2 LN fold + GEGLU consumer {L.TILE_NAMES[ctile]} M{M} C{C}", rtol=2.5e-2, atol_rms=2.5e-2,
                          rel_rms_max=8e-3)
    # the automatic choice (tile=None) follows the per-shape table onto the K2 tiles where it lists them
    st = ops.RowStats(M, DEV)
    xa = ops.linear(a, wprod, residual=res, stats_out=st)
    ya = ops.linear(xa, wl, b, ln=(st, fold))
    refa = F.layer_norm(xa.float(), (C,), gamma.float(), beta.float(), 1e-5) @ w.float().t() + b.float()
    assert_close_bf16(ya, refa, "LN fold, automatic variant choice", rel_rms_max=6e-3)


@pytest.mark.parametrize("M,C,inner", [(2048, 1280, 1280), (8192, 640, 640), (520, 320, 320)])
def test_k2_fused_qkv_projection_with_transposed_v_block(M, C, inner):
    """da_gemm_params.vt (round 4): the fused to_q | to_k | to_v projection of a self-attention layer in ONE launch -- columns
    [0, 2 * inner) into C, the V block transposed into vt[channel][token] -- on every variant that carries it, plain and with the
    LayerNorm fold (norm1 applied inside the launch from the producer's statistics); against the separate projections of the
    reference processor (attention_processor.py:2743-2751) in fp32, and bit-identical across variants."""
    ops, L = _ops()
    a, wprod, res = rnd((M, 192), 71), rnd((C, 192), 72, 192 ** -0.5), rnd((M, C), 73)
    gamma, beta = rnd((C,), 74) * 0.3 + 1.0, rnd((C,), 75) * 0.2
    wqkv = rnd((3 * inner, C), 76, C ** -0.5)
    st = ops.RowStats(M, DEV)
    x = ops.linear(a, wprod, residual=res, tile=L.TILE_K2_128x80, staging=L.STAGE_PINGPONG, stats_out=st)
    xf = x.float()
    # plain
    ref = xf @ wqkv.float().t()
    bases = {}                                   # per kernel family: two K-groups (k2) and one (k1) add the K slices in different orders
    n_ok = 0
    for tile, stg in ops.QKV_CANDIDATES:
        if (2 * inner) % ops.QKV_TILE_COLS[tile]:
            continue
        vt = torch.full((inner, M), float("nan"), device=DEV, dtype=bf16)
        try:
            qk = ops.linear(x, wqkv, tile=tile, staging=stg, vt_out=(vt, 2 * inner))
        except RuntimeError as e:
            assert "DA_ERR_UNSUPPORTED" in str(e), e
            continue
        n_ok += 1
        assert qk.shape == (M, 2 * inner)
        assert_close_bf16(qk, ref[:, :2 * inner], f"fused QKV (Q|K block) {L.TILE_NAMES[tile]}/{stg} M{M}", rtol=8e-3, atol_rms=4e-3)
        assert_close_bf16(vt.t(), ref[:, 2 * inner:], f"fused QKV (V^T block) {L.TILE_NAMES[tile]}/{stg} M{M}", rtol=8e-3, atol_rms=4e-3)
        fam = "k1" if tile >= L.TILE_K1_128x320 else "k2"
        if fam not in bases:
            bases[fam] = (qk.clone(), vt.clone())
        assert torch.equal(qk, bases[fam][0]) and torch.equal(vt, bases[fam][1]), f"{fam} variants of the fused projection differ"
    assert n_ok >= 2 and "k2" in bases
    base = bases["k2"]
    if M % 128 == 0 and (2 * inner) % 256 == 0:
        assert "k1" in bases, "the one-round tiles of the fused projection refused an aligned problem"
    # the LayerNorm fold on every variant, each against fp32
    wl_, fold_ = ops.fold_layernorm(wqkv, gamma, beta, 1e-5)
    ln_ref_ = F.layer_norm(xf, (C,), gamma.float(), beta.float(), 1e-5) @ wqkv.float().t()
    for tile, stg in ops.QKV_CANDIDATES:
        if (2 * inner) % ops.QKV_TILE_COLS[tile]:
            continue
        vt = torch.full((inner, M), float("nan"), device=DEV, dtype=bf16)
        try:
            qk = ops.linear(x, wl_, ln=(st, fold_), tile=tile, staging=stg, vt_out=(vt, 2 * inner))
        except RuntimeError as e:
            assert "DA_ERR_UNSUPPORTED" in str(e), e
            continue
        assert_close_bf16(qk, ln_ref_[:, :2 * inner], f"fused QKV + LN fold (Q|K) {L.TILE_NAMES[tile]}/{stg} M{M}", rel_rms_max=6e-3)
        assert_close_bf16(vt.t(), ln_ref_[:, 2 * inner:], f"fused QKV + LN fold (V^T) {L.TILE_NAMES[tile]}/{stg} M{M}", rel_rms_max=6e-3)
    # the separate K2 launches give the same bits as the fused one (same tiles, same summation order)
    sep = ops.linear(x, wqkv[:2 * inner].contiguous(), tile=L.TILE_K2_128x80, staging=L.STAGE_PINGPONG)
    assert torch.equal(sep, base[0])
    # with the LayerNorm fold
    wl, fold = ops.fold_layernorm(wqkv, gamma, beta, 1e-5)
    ln_ref = F.layer_norm(xf, (C,), gamma.float(), beta.float(), 1e-5) @ wqkv.float().t()
    qk, vt = ops.linear_qkv(x, wl, 2 * inner, ln=(st, fold))
    assert_close_bf16(qk, ln_ref[:, :2 * inner], f"fused QKV + LN fold (Q|K) M{M} C{C}", rel_rms_max=6e-3)
    assert_close_bf16(vt.t(), ln_ref[:, 2 * inner:], f"fused QKV + LN fold (V^T) M{M} C{C}", rel_rms_max=6e-3)
    # refusals: a column origin the tiles do not divide, an activation, a residual
    with pytest.raises((RuntimeError, ValueError)):
        ops.linear(x, wqkv, tile=L.TILE_K2_128x80, staging=L.STAGE_PINGPONG, vt_out=(torch.empty((3 * inner - 64, M), device=DEV, dtype=bf16), 64))
    with pytest.raises((RuntimeError, ValueError)):
        ops.linear(x, wqkv, residual=rnd((M, 2 * inner), 77), vt_out=(torch.empty((inner, M), device=DEV, dtype=bf16), 2 * inner))
