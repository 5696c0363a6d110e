"""Cross-attention in the epilogue of attn2.to_q (round 5: da_gemm_params.xa_*, the XA instantiation of csrc/gemm2_kernel.cuh).

One launch computes q = to_q(x) (optionally with norm2 folded in) and softmax(scale q k^T) v against the step-invariant K / V^T of the
text embeddings -- what the reference does as `attn.to_q` + `F.scaled_dot_product_attention` (attention_processor.py:2743-2777 with
encoder_hidden_states of 77 CLIP tokens).  Checked against (a) the two-launch path of the same library (to_q, then da_attention_bf16)
and (b) the fp32 torch computation, at the SDXL shapes (M 2048 x 1280 / 20 heads; M 8192 x 640 / 10 heads) and at edge shapes (one
batch, one head pair, 8 / 80 keys)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close_bf16, rel_rms

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16
DEV = "cuda"


def rnd(shape, seed, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(bf16).to(DEV)


def _setup(B, seq, C, heads, cross, skv, seed=0):
    from diffusers_amd import layers
    inner = heads * 64
    x = rnd((B * seq, C), seed + 1)
    wq, wk, wv = (rnd((inner, c_), seed + 2 + i, c_ ** -0.5) for i, c_ in enumerate((C, cross, cross)))
    ehs = rnd((B, skv, cross), seed + 7)
    pad, s, sa = layers.pad_encoder_states(ehs)
    return x, wq, wk, wv, ehs, pad, s, sa, inner


def _ref(x, wq, wk, wv, ehs, B, seq, heads, bq=None, ln=None):
    xf = x.float()
    if ln is not None:
        xf = F.layer_norm(xf, (xf.shape[1],), ln[0].float(), ln[1].float(), 1e-5)
    q = (xf @ wq.float().t() + (0 if bq is None else bq.float())).to(bf16).float().view(B, seq, heads, 64).transpose(1, 2)
    k = (ehs.float() @ wk.float().t()).to(bf16).float().view(B, -1, heads, 64).transpose(1, 2)
    v = (ehs.float() @ wv.float().t()).to(bf16).float().view(B, -1, heads, 64).transpose(1, 2)
    return F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B * seq, heads * 64)


@pytest.mark.parametrize("B,seq,C,heads,cross,skv", [(2, 1024, 1280, 20, 2048, 77), (2, 4096, 640, 10, 2048, 77), (1, 128, 128, 2, 64, 77),
                                                    (3, 256, 256, 4, 128, 80), (1, 384, 128, 2, 64, 8), (2, 128, 192, 6, 64, 33)])
def test_cross_attention_in_the_to_q_epilogue(B, seq, C, heads, cross, skv):
    from diffusers_amd import _lib as L, ops
    x, wq, wk, wv, ehs, pad, s, sa, inner = _setup(B, seq, C, heads, cross, skv)
    k = ops.linear(pad, wk)
    vt = ops.linear(wv, pad)
    scale = 64 ** -0.5
    xa = {"k": k, "vt": vt, "skv": s, "skv_alloc": sa, "seq": seq, "scale": scale}
    want = _ref(x, wq, wk, wv, ehs, B, seq, heads)
    q = ops.linear(x, wq)
    two = ops.attention(q, k, vt, B=B, H=heads, D=64, Sq=seq, Skv=s, Skv_alloc=sa, q_row_stride=inner, k_row_stride=inner,
                        q_batch_stride=seq * inner, k_batch_stride=sa * inner, vt_ld=B * sa, vt_batch_stride=sa, scale=scale)
    base = None
    for stg in (L.STAGE_PINGPONG, L.STAGE_LDS_DIRECT):
        got = ops.linear(x, wq, xattn=xa, tile=L.TILE_K2_128x128, staging=stg)
        what = f"fused to_q + cross-attention B{B} S{seq} C{C} H{heads} keys {skv} staging {stg}"
        assert got.shape == (B * seq, inner) and got.dtype == bf16
        assert_close_bf16(got, want, what + " vs fp32", rtol=1.6e-2, atol_rms=1.6e-2, rel_rms_max=6e-3)
        r2 = rel_rms(got, two)
        print(f"[parity] {what}: rel_rms vs the two-launch path {r2:.3e}; two-launch vs fp32 {rel_rms(two, want):.3e}")
        assert r2 < 6e-3
        if base is None:
            base = got
        assert torch.equal(got, base), "the two ring schedules must give the same bits"
        assert torch.equal(got, ops.linear(x, wq, xattn=xa, tile=L.TILE_K2_128x128, staging=stg)), "deterministic"
    # with a bias on to_q
    bq = rnd((inner,), 91)
    gotb = ops.linear(x, wq, bq, xattn=xa)
    assert_close_bf16(gotb, _ref(x, wq, wk, wv, ehs, B, seq, heads, bq=bq), "fused, to_q bias", rtol=1.6e-2, atol_rms=1.6e-2, rel_rms_max=6e-3)
    # refusals: anything behind q but the attention, more than 80 keys, a row tile across two batches
    with pytest.raises(ValueError):
        ops.linear(x, wq, xattn=xa, residual=q)
    with pytest.raises(ValueError):
        ops.linear(x, wq, xattn=dict(xa, skv_alloc=96))
    with pytest.raises(ValueError):
        ops.linear(x, wq, xattn=dict(xa, seq=seq + 64))


@pytest.mark.parametrize("B,seq,C,heads", [(2, 1024, 1280, 20), (2, 4096, 640, 10)])
def test_cross_attention_epilogue_with_the_layernorm_fold(B, seq, C, heads):
    """norm2 folded into the same launch (attention.py:1030: LN -> attn2): statistics from the producer's epilogue (attn1.to_out)."""
    from diffusers_amd import ops
    x0, wq, wk, wv, ehs, pad, s, sa, inner = _setup(B, seq, C, heads, 2048, 77, seed=20)
    a, wprod = rnd((B * seq, 192), 31), rnd((C, 192), 32, 192 ** -0.5)
    gamma, beta = rnd((C,), 33) * 0.3 + 1.0, rnd((C,), 34) * 0.2
    st = ops.RowStats(B * seq, DEV)
    x = ops.linear(a, wprod, residual=x0, stats_out=st)          # the residual stream and its row statistics
    wl, fold = ops.fold_layernorm(wq, gamma, beta, 1e-5)
    k, vt = ops.linear(pad, wk), ops.linear(wv, pad)
    xa = {"k": k, "vt": vt, "skv": s, "skv_alloc": sa, "seq": seq, "scale": 64 ** -0.5}
    got = ops.linear(x, wl, ln=(st, fold), xattn=xa)
    want = _ref(x, wq, wk, wv, ehs, B, seq, heads, ln=(gamma, beta))
    assert_close_bf16(got, want, f"fused LN-fold to_q + cross-attention M{B * seq} C{C}", rtol=2.5e-2, atol_rms=3e-2, rel_rms_max=8e-3)
    q = ops.linear(x, wl, ln=(st, fold))
    two = ops.attention(q, k, vt, B=B, H=heads, D=64, Sq=seq, Skv=s, Skv_alloc=sa, q_row_stride=inner, k_row_stride=inner,
                        q_batch_stride=seq * inner, k_batch_stride=sa * inner, vt_ld=B * sa, vt_batch_stride=sa, scale=64 ** -0.5)
    print(f"[parity] fused LN-fold to_q + cross-attention M{B * seq}: vs two launches {rel_rms(got, two):.3e}, vs fp32 {rel_rms(got, want):.3e} "
          f"(two launches vs fp32 {rel_rms(two, want):.3e})")
    assert rel_rms(got, two) < 6e-3


def test_attention_layer_takes_the_fused_launch():
    """layers.Attention (cross) routes eligible shapes through ONE launch and the rest through to_q + da_attention_bf16; both agree."""
    from diffusers_amd import layers, ops
    B, seq, C, heads = 2, 256, 256, 4
    sd = {"a.to_q.weight": rnd((C, C), 1, C ** -0.5), "a.to_k.weight": rnd((C, 64), 2, 0.125), "a.to_v.weight": rnd((C, 64), 3, 0.125),
          "a.to_out.0.weight": rnd((C, C), 4, C ** -0.5), "a.to_out.0.bias": rnd((C,), 5)}
    att = layers.Attention(layers.Weights(sd, DEV), "a", heads, cross=True)
    pad, s, sa = layers.pad_encoder_states(rnd((B, 77, 64), 6))
    kv = att.precompute_kv(pad, B, s, sa)
    x = rnd((B * seq, C), 7)
    calls = []
    orig = ops.attention
    ops.attention = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    saved = ops.XATTN
    try:
        ops.XATTN = True
        y1 = att(x, B, seq, residual=x, kv=kv)
        n_fused = len(calls)
        ops.XATTN = False
        y2 = att(x, B, seq, residual=x, kv=kv)
    finally:
        ops.attention, ops.XATTN = orig, saved
    assert n_fused == 0 and len(calls) == 1
    assert rel_rms(y1, y2) < 4e-3
