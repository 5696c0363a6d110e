"""GPU tests of the eight-phase 256 x 256 tile (csrc/gemm3.hip, DA_TILE_K3_256x256): two wave rows half a phase apart, one
multiplying while the other reads LDS and issues LDS-DMA (nn.Linear: transformer_flux.py:383-412, transformer_wan.py:488-502,
activations.py:113-124).

What must hold: (1) BIT-identical to DA_TILE_K1_256x256 of gemm2_kernel.cuh -- the same K order (slices in order, k-step 0 then 1),
the same MFMA, the same epilogue -- on every shape, ragged edges, odd slice counts and every fused epilogue; since the two kernels
share nothing but that arithmetic, one wrong or stale LDS byte anywhere shows as an inequality; (2) right against a plain PyTorch
fp32 reference; (3) the same bits launch after launch (the schedule places every ds_read by counted vmcnt + barrier: a read that
could overtake its LDS-DMA would come and go with timing) at 256 / 512 / 4096-sized problems; (4) refusals by name."""
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


def k3(ops, L, x, w, *a, **kw):
    return ops.linear(x, w, *a, tile=L.TILE_K3_256x256, staging=L.STAGE_LDS_DIRECT, **kw)


def k1(ops, L, x, w, *a, **kw):
    return ops.linear(x, w, *a, tile=L.TILE_K1_256x256, staging=L.STAGE_LDS_DIRECT, **kw)


# Flux / Wan / SDXL projections (reduced where the fp32 reference would dominate the run time), one tile, one slice (K = 64), odd slice
# counts (K = 192, 320, 1088), ragged M / N edges, fewer rows / columns than a tile
@pytest.mark.parametrize("M,N,K", [(4608, 3072, 3072), (2048, 10240, 1280), (8192, 1280, 640), (256, 256, 64), (256, 256, 128),
                                   (512, 512, 192), (300, 320, 320), (777, 644, 1152), (130, 1284, 1088), (4096, 4096, 256),
                                   (32760 // 8, 5120, 512)])
def test_k3_bit_identical_to_k1(M, N, K):
    ops, L = _ops()
    x, w, b, r = rnd((M, K), 31), rnd((N, K), 32, K ** -0.5), rnd((N,), 33), rnd((M, N), 34)
    y = k3(ops, L, x, w, b, residual=r)
    want = k1(ops, L, x, w, b, residual=r)
    assert torch.equal(y, want), f"k3 {M}x{N}x{K}: {int((y != want).sum())} outputs differ from k1:256x256"
    ref = x.float() @ w.float().t() + b.float() + r.float()
    assert_close_bf16(y, ref, f"k3 gemm {M}x{N}x{K}", rtol=8e-3, atol_rms=4e-3)


@pytest.mark.parametrize("n", [256, 512, 4096])
def test_k3_race_screen(n):
    """The same bits on every one of 25 launches, with other traffic between them (a 64 MiB fill moves the LDS-DMA's arrival times)."""
    ops, L = _ops()
    x, w = rnd((n, n), 41), rnd((n, n), 42, n ** -0.5)
    want = k1(ops, L, x, w)
    noise = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    for it in range(25):
        if it % 3 == 1:
            noise.fill_(it)
        y = k3(ops, L, x, w)
        assert torch.equal(y, want), f"k3 {n}^3 launch {it}: {int((y != want).sum())} outputs differ"


@pytest.mark.parametrize("act", ["none", "gelu_tanh", "silu"])
def test_k3_epilogues(act):
    ops, L = _ops()
    A = {"none": L.ACT_NONE, "silu": L.ACT_SILU, "gelu_tanh": L.ACT_GELU_TANH}[act]
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
        got, want = k3(ops, L, x, w, **kw), k1(ops, L, x, w, **kw)
        assert got.dtype == want.dtype and torch.equal(got, want), f"k3 epilogue {act} / {name} differs from k1:256x256"


def test_k3_strided_operands_and_outputs():
    ops, L = _ops()
    M, N, K = 1024, 640, 640
    xx, ww = rnd((M, 2 * K), 1), rnd((N, 2 * K), 2, K ** -0.5)
    out = torch.zeros((M, 2 * N), device=DEV, dtype=bf16)
    res = rnd((M, 2 * N), 3)
    x, w, r = xx[:, K:], ww[:, :K], res[:, N:]
    k3(ops, L, x, w, residual=r, out=out[:, :N])
    assert_close_bf16(out[:, :N], x.float() @ w.float().t() + r.float(), "k3 strided", rtol=8e-3, atol_rms=4e-3)
    assert float(out[:, N:].abs().max()) == 0.0, "wrote outside its column block"


@pytest.mark.parametrize("M,N2,K", [(2048, 10240, 1280), (300, 256, 192), (8192, 5120, 640)])
def test_k3_geglu(M, N2, K):
    ops, L = _ops()
    x, w, b = rnd((M, K), 1), rnd((N2, K), 2, K ** -0.5), rnd((N2,), 3)
    wp, bp = ops.pack_geglu(w, b)
    h = (x.float() @ w.float().t() + b.float()).to(bf16).float()
    ref = h[:, : N2 // 2] * F.gelu(h[:, N2 // 2:]).to(bf16).float()
    y = k3(ops, L, x, wp, bias=bp, act=L.ACT_GEGLU)
    assert y.shape == (M, N2 // 2)
    assert torch.equal(y, k1(ops, L, x, wp, bias=bp, act=L.ACT_GEGLU)), "k3 GEGLU differs from k1:256x256"
    assert_close_bf16(y, ref, f"k3 geglu {M}x{N2}x{K}", rtol=1.6e-2, atol_rms=6e-3)


def test_k3_refuses_what_it_does_not_implement():
    ops, L = _ops()
    x, w = rnd((512, 256), 1), rnd((512, 256), 2)
    for st in (L.STAGE_LDS_DIRECT3, L.STAGE_PINGPONG, L.STAGE_REGISTER):
        with pytest.raises(RuntimeError, match="UNSUPPORTED"):
            ops.linear(x, w, tile=L.TILE_K3_256x256, staging=st)
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        ops.linear(x, w, tile=L.TILE_K3_256x256, staging=L.STAGE_LDS_DIRECT, split_k=2)
    xc, wc = rnd((1, 16, 16, 64), 3), rnd((64, 9 * 64), 4)
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        ops.conv2d_nhwc(xc, wc, None, ksize=3, tile=L.TILE_K3_256x256, staging=L.STAGE_LDS_DIRECT)


# ---- the GEGLU projection's own eight-phase tile (DA_TILE_K3_256x320: 4 x 2 waves of 64 x 160, value / gate tiles paired in a wave) ----
def k3g(ops, L, x, w, *a, **kw):
    return ops.linear(x, w, *a, tile=L.TILE_K3_256x320, staging=L.STAGE_LDS_DIRECT, **kw)


# SDXL's two GEGLU projections, one tile / one slice, odd slice counts (K = 192, 320), several tiles in both directions
# (N2 = packed value + gate columns: whole 320-column tiles AND whole 64-column [32 value | 32 gate] groups -> multiples of 640)
@pytest.mark.parametrize("M,N2,K", [(2048, 10240, 1280), (8192, 5120, 640), (256, 640, 64), (256, 640, 192), (512, 1280, 320),
                                    (1024, 1280, 2560), (768, 1920, 128)])
@pytest.mark.parametrize("tanh", [False, True])
def test_k3_geglu_tile_bit_identical_to_k1(M, N2, K, tanh):
    ops, L = _ops()
    act = L.ACT_GEGLU_TANH if tanh else L.ACT_GEGLU
    x, w, b = rnd((M, K), 51), rnd((N2, K), 52, K ** -0.5), rnd((N2,), 53)
    wp, bp = ops.pack_geglu(w, b)
    y = k3g(ops, L, x, wp, bias=bp, act=act)
    want = ops.linear(x, wp, bias=bp, act=act, tile=L.TILE_K1_128x320, staging=L.STAGE_LDS_DIRECT)
    assert y.shape == (M, N2 // 2)
    assert torch.equal(y, want), f"k3:256x320 {M}x{N2}x{K}: {int((y != want).sum())} outputs differ from k1:128x320"
    h = (x.float() @ w.float().t() + b.float()).to(bf16).float()
    ref = h[:, : N2 // 2] * F.gelu(h[:, N2 // 2:], approximate="tanh" if tanh else "none").to(bf16).float()
    assert_close_bf16(y, ref, f"k3:256x320 geglu {M}x{N2}x{K}", rtol=1.6e-2, atol_rms=6e-3)
    # no bias, alpha, a strided output block
    out = torch.zeros((M, N2), device=DEV, dtype=bf16)
    k3g(ops, L, x, wp, act=act, alpha=0.5, out=out[:, : N2 // 2])
    assert torch.equal(out[:, : N2 // 2], ops.linear(x, wp, act=act, alpha=0.5, tile=L.TILE_K1_128x320, staging=L.STAGE_LDS_DIRECT))
    assert float(out[:, N2 // 2:].abs().max()) == 0.0, "wrote outside its column block"


# Round 6: the tile as the CONSUMER of a folded LayerNorm (norm3 -> ff.net.0.proj, attention.py:1056-1080): statistics partials of
# the producing launch (16 per row at C 1280, 8 at C 640, 3 / 24 for the slot masks), mean / rstd formed in the prologue, the fold applied
# to the accumulators in front of the bias.
@pytest.mark.parametrize("M,C,N2", [(2048, 1280, 10240), (8192, 640, 5120), (256, 192, 640), (512, 1920, 1280), (256, 320, 640)])
@pytest.mark.parametrize("tanh", [False, True])
def test_k3_geglu_tile_layernorm_fold_bit_identical_to_k1(M, C, N2, tanh):
    ops, L = _ops()
    act = L.ACT_GEGLU_TANH if tanh else L.ACT_GEGLU
    a, wprod, res = rnd((M, 192), 71), rnd((C, 192), 72, 192 ** -0.5), rnd((M, C), 73)
    res = res + 8.0 * (torch.arange(M, device=DEV) % 3 == 0).to(bf16)[:, None]      # rows with a large mean
    gamma, beta = rnd((C,), 74) * 0.3 + 1.0, rnd((C,), 75) * 0.2
    w1, b1 = rnd((N2, C), 76, C ** -0.5), rnd((N2,), 77)
    w1l, fold1 = ops.fold_layernorm(w1, gamma, beta, 1e-5)
    w1p, b1p = ops.pack_geglu(w1l, b1)
    n2 = N2 // 2
    idx = torch.arange(n2, device=DEV).view(n2 // 32, 32)
    order = torch.cat([idx, idx + n2], dim=1).reshape(-1)
    fold1p = ops.LNFold(fold1.s[order].contiguous(), fold1.c[order].contiguous(), fold1.eps)
    st = ops.RowStats(M, DEV)
    x = ops.linear(a, wprod, residual=res, tile=L.TILE_K2_128x80, staging=L.STAGE_PINGPONG, stats_out=st)
    assert st.parts == (C + 79) // 80
    y = k3g(ops, L, x, w1p, b1p, act=act, ln=(st, fold1p))
    want = ops.linear(x, w1p, b1p, act=act, tile=L.TILE_K1_128x320, staging=L.STAGE_LDS_DIRECT, ln=(st, fold1p))
    assert torch.equal(y, want), f"k3:256x320 + LayerNorm fold {M}x{N2}x{C}: {int((y != want).sum())} outputs differ from k1:128x320"
    ln_ref = F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5)
    g = ln_ref @ w1.float().t() + b1.float()
    h_, g_ = g.chunk(2, dim=-1)
    ref = h_ * F.gelu(g_, approximate="tanh" if tanh else "none")
    assert_close_bf16(y, ref, f"k3:256x320 LN fold + GEGLU {M}x{N2}x{C}", rtol=2.5e-2, atol_rms=2.5e-2, rel_rms_max=8e-3)
    # the same launch again and again with other traffic in between (the statistics loads and the s / c pieces are the OLDEST loads of
    # the prologue: the counted waits of the loop must never see them), and without a bias / with alpha
    noise = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
    for it in range(6):
        if it % 2:
            noise.fill_(it)
        assert torch.equal(k3g(ops, L, x, w1p, b1p, act=act, ln=(st, fold1p)), want), f"launch {it} differs"
    y2 = k3g(ops, L, x, w1p, act=act, alpha=0.5, ln=(st, fold1p))
    assert torch.equal(y2, ops.linear(x, w1p, act=act, alpha=0.5, tile=L.TILE_K1_128x320, staging=L.STAGE_LDS_DIRECT, ln=(st, fold1p)))


def test_k3_geglu_tile_layernorm_fold_is_the_automatic_choice():
    """ops.linear(ln=) without a pinned tile follows the shipped table onto k3:256x320 for SDXL's GEGLU projections (round 6; rounds 4-5
    sent the folded launch to k1:128x320) -- same bits either way."""
    ops, L = _ops()
    from diffusers_amd import tuning
    M, C, N2 = 2048, 1280, 10240
    a, wprod, res = rnd((M, 192), 81), rnd((C, 192), 82, 192 ** -0.5), rnd((M, C), 83)
    gamma, beta = rnd((C,), 84) * 0.3 + 1.0, rnd((C,), 85) * 0.2
    w1, b1 = rnd((N2, C), 86, C ** -0.5), rnd((N2,), 87)
    w1l, fold1 = ops.fold_layernorm(w1, gamma, beta, 1e-5)
    w1p, b1p = ops.pack_geglu(w1l, b1)
    idx = torch.arange(N2 // 2, device=DEV).view(N2 // 64, 32)
    order = torch.cat([idx, idx + N2 // 2], dim=1).reshape(-1)
    fold1p = ops.LNFold(fold1.s[order].contiguous(), fold1.c[order].contiguous(), fold1.eps)
    st = ops.RowStats(M, DEV)
    x = ops.linear(a, wprod, residual=res, stats_out=st)
    ent = tuning.table().get(f"lin:M{M}:N{N2}:K{C}:a{L.ACT_GEGLU}:f0:r0")
    if ent is None or ent[0] != L.TILE_K3_256x320:
        pytest.skip("the table in use does not send this shape to k3:256x320")
    y = ops.linear(x, w1p, b1p, act=L.ACT_GEGLU, ln=(st, fold1p))
    assert torch.equal(y, k3g(ops, L, x, w1p, b1p, act=L.ACT_GEGLU, ln=(st, fold1p)))


def test_k3_geglu_tile_race_screen():
    ops, L = _ops()
    M, N2, K = 2048, 10240, 1280
    x, w, b = rnd((M, K), 61), rnd((N2, K), 62, K ** -0.5), rnd((N2,), 63)
    wp, bp = ops.pack_geglu(w, b)
    want = ops.linear(x, wp, bias=bp, act=L.ACT_GEGLU, tile=L.TILE_K1_128x320, staging=L.STAGE_LDS_DIRECT)
    noise = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    for it in range(25):
        if it % 3 == 1:
            noise.fill_(it)
        y = k3g(ops, L, x, wp, bias=bp, act=L.ACT_GEGLU)
        assert torch.equal(y, want), f"k3:256x320 launch {it}: {int((y != want).sum())} outputs differ"


def test_k3_geglu_tile_refusals():
    ops, L = _ops()
    x, w = rnd((256, 128), 1), rnd((640, 128), 2)
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        k3g(ops, L, x, w)                                               # not a GEGLU epilogue
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        k3g(ops, L, rnd((300, 128), 3), w, act=L.ACT_GEGLU)             # M is not whole tiles
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        k3g(ops, L, x, rnd((256, 128), 4), act=L.ACT_GEGLU)             # N is not whole tiles
