"""Key-split tail of the flash kernel (round 6; include/diffusers_amd.h da_attention_params.split_ws, csrc/attention2.hip).

Replaces nothing new in the reference -- it is the same F.scaled_dot_product_attention call (models/attention_processor.py:2767,
models/attention_dispatch.py:3709) -- but changes HOW a launch covers the chip: the query blocks of the last, partial round of CUs are
split over the keys, their (O, m, l) partials meet in a workspace and the last arriver of a block combines them in unit order.

CPU: the host-side plan (`da_attention_split_plan`, a pure function of the shape) for the BASELINE shapes.  GPU: the split launches
against the unsplit kernel and against fp32 SDPA, pinned split factors, ragged key counts, the deferred-maximum stress inside one
unit's key range, run-to-run bit-identity under other traffic (the combine order does not depend on who arrives last) and HIP-graph
replays (the ticket counters are re-armed by the kernel)."""
import ctypes as C

import pytest
import torch

from conftest import rel_rms

bf16 = torch.bfloat16
DEV = "cuda"


def _plan(B, H, Sq, Skv, D, q_block=0, kv_split=0):
    from diffusers_amd import _lib as L
    p = L.AttentionParams()
    p.B, p.H, p.Sq, p.Skv, p.Skv_alloc, p.D, p.q_block, p.kv_split = B, H, Sq, Skv, (Skv + 7) // 8 * 8, D, q_block, kv_split
    full, tail, s = C.c_int(), C.c_int(), C.c_int()
    need = L.load().da_attention_split_plan(C.byref(p), C.byref(full), C.byref(tail), C.byref(s))
    return int(need), full.value, tail.value, s.value


def test_split_plan_of_the_baseline_shapes():
    """256 CUs (the plan falls back to MI355X's count without a device).  SDXL S = 1024: 320 blocks = 256 whole + 64 split four
    ways (every CU: one whole block of 16 tiles + one unit of 4); S = 4096 (20 (batch, head) pairs: not a multiple of the XCD count
    -- the balanced block mapping gives every XCD 80 blocks): 640 = 512 + 128 split in two; Flux: 432 blocks of 256 queries = 256 +
    176 split in four (three rounds of 18 tiles instead of one of 72); Wan's 3072 blocks are whole rounds: no split."""
    from diffusers_amd import _lib as L
    cb = L.ATTN_SPLIT_COUNTER_BYTES
    assert _plan(2, 20, 1024, 1024, 64) == (cb + 64 * 4 * 4 * (64 * 128 + 512), 256, 64, 4)
    assert _plan(2, 10, 4096, 4096, 64) == (cb + 128 * 2 * 4 * (64 * 128 + 512), 512, 128, 2)
    assert _plan(1, 24, 4608, 4608, 128)[3] == 1                   # D = 128 never splits on its own (measured: a loss) ...
    assert _plan(1, 24, 4608, 4608, 128, kv_split=4) == (cb + 176 * 4 * 8 * (128 * 128 + 512), 256, 176, 4)      # ... only pinned
    assert _plan(2, 12, 32760, 32760, 128)[3] == 1 and _plan(2, 12, 32760, 32760, 128)[0] == 0
    assert _plan(2, 20, 1024, 77, 64)[3] == 1                      # cross-attention: two key tiles, nothing to split
    assert _plan(2, 20, 1024, 1024, 64, kv_split=1)[3] == 1        # switched off
    assert _plan(2, 20, 1024, 1024, 64, kv_split=2)[1:] == (256, 64, 2)
    assert _plan(1, 3, 1000, 1024, 64)[1:] == (0, 24, 4)           # 24 blocks (a multiple of 8): no whole round at all, every block split
    assert _plan(1, 3, 1100, 1024, 64)[3] == 1                     # 27 blocks: not a multiple of the XCD count -> legacy mapping, whole
    # units are whole tiles, at least four of them (256 keys), none empty: 16 tiles cut in 8 would be units of two
    assert _plan(2, 20, 1024, 1024, 64, kv_split=8)[3] == 1
    assert _plan(2, 20, 1024, 512, 64, kv_split=2)[1:] == (256, 64, 2)
    assert _plan(2, 20, 1024, 448, 64)[3] == 1                     # 7 tiles: too short to split


def _sdpa(q, k, v, B, H, D, Sq, Skv):
    """fp32 reference on the device: q [B*Sq][H*D], k / v [B][Skv][H*D]."""
    qh = q.float().view(B, Sq, H, D).transpose(1, 2)
    kh = k.float().view(B, Skv, H, D).transpose(1, 2)
    vh = v.float().view(B, Skv, H, D).transpose(1, 2)
    o = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh)
    return o.transpose(1, 2).reshape(B * Sq, H * D)


def _case(B, H, D, Sq, Skv, seed=0, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    C_ = H * D
    q = (torch.randn((B * Sq, C_), generator=g) * scale).to(bf16).to(DEV)
    k = (torch.randn((B, Skv, C_), generator=g) * scale).to(bf16).to(DEV)
    v = torch.randn((B, Skv, C_), generator=g).to(bf16).to(DEV)
    sa = (Skv + 15) // 16 * 16
    kp = torch.zeros((B, sa, C_), device=DEV, dtype=bf16)
    kp[:, :Skv] = k
    vt = torch.zeros((C_, B * sa), device=DEV, dtype=bf16)
    vt.view(C_, B, sa)[:, :, :Skv] = v.permute(2, 0, 1)
    return q, k, v, kp, vt, sa


def _run(ops, q, kp, vt, B, H, D, Sq, Skv, sa, **kw):
    C_ = H * D
    return ops.attention(q, kp.view(B * sa, C_), vt, B=B, H=H, D=D, Sq=Sq, Skv=Skv, Skv_alloc=sa, q_row_stride=C_, k_row_stride=C_,
                         q_batch_stride=Sq * C_, k_batch_stride=sa * C_, vt_ld=B * sa, vt_batch_stride=sa, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,D,Sq,Skv", [(2, 20, 64, 1024, 1024), (2, 10, 64, 4096, 4096), (1, 24, 128, 4608, 4608),
                                         (2, 20, 64, 1024, 1000), (2, 8, 64, 1280, 700), (1, 16, 128, 2560, 1100)])
def test_split_launch_vs_unsplit_and_fp32(B, H, D, Sq, Skv):
    from diffusers_amd import ops
    q, k, v, kp, vt, sa = _case(B, H, D, Sq, Skv, seed=Sq + Skv)
    # D = 64: the library's own choice; D = 128 does not split on its own (measured: a loss, profiles/r06_attention.jsonl): pinned
    pin0 = 0 if D == 64 else 4
    need, full, tail, s = _plan(B, H, Sq, Skv, D, kv_split=pin0)
    assert s > 1, "the case is meant to split"
    ref = _sdpa(q, k, v, B, H, D, Sq, Skv)
    whole = _run(ops, q, kp, vt, B, H, D, Sq, Skv, sa, kv_split=1)
    auto = _run(ops, q, kp, vt, B, H, D, Sq, Skv, sa, kv_split=pin0)
    r_whole, r_auto, r_between = rel_rms(whole, ref), rel_rms(auto, ref), rel_rms(auto, whole)
    same = float((auto == whole).float().mean())
    print(f"[split] B{B} H{H} D{D} Sq{Sq} Skv{Skv}: {full} whole blocks + {tail} x {s} units; rel-rms vs fp32 SDPA: whole {r_whole:.3e}, "
          f"split {r_auto:.3e}; split vs whole {r_between:.3e}, {100 * same:.2f} % of the outputs bit-equal")
    assert torch.isfinite(auto.float()).all()
    assert r_auto < 4e-3 and r_auto <= 1.05 * r_whole + 1e-5          # the same kernel's accuracy (fp32 partials, one more fp32 combine)
    # (split and whole blocks round P = 2^(s - m) to bf16 under DIFFERENT running shifts m, so a split block's outputs differ from
    # the unsplit launch's in the last bf16 bit about as often as not -- as two flash kernels with different tile sizes do.  Two
    # results that each sit r from fp32 with independent roundings sit up to sqrt(2) r from each other: measured 1.1e-3 .. 2.7e-3,
    # the latter when EVERY block of the launch is split)
    assert r_between < 1.5 * r_whole
    # whole blocks are untouched by the split: their rows are the unsplit launch's bits
    QT = 256 if (D == 128 and B * H * ((Sq + 255) // 256) >= 192) else 128
    qtiles = (Sq + QT - 1) // QT
    rows_equal = (auto.view(B, Sq, H, D) == whole.view(B, Sq, H, D)).all(dim=-1)        # [B][Sq][H]
    n_whole_checked = 0
    nb8 = B * H * qtiles // 8
    for L_ in range(0, full, max(1, full // 64)):          # physical block id -> logical block (balanced mapping of attention2.hip)
        xcd, slot = L_ & 7, L_ >> 3
        lb = xcd * nb8 + slot
        pair, qt = lb // qtiles, lb % qtiles
        b, h = pair // H, pair % H
        assert bool(rows_equal[b, qt * QT:(qt + 1) * QT, h].all()), f"whole block {L_} differs from the unsplit launch"
        n_whole_checked += 1
    assert n_whole_checked > 0 or full == 0
    for pin in (2, 3, 4, 8):
        if _plan(B, H, Sq, Skv, D, kv_split=pin)[3] != pin:
            continue
        o = _run(ops, q, kp, vt, B, H, D, Sq, Skv, sa, kv_split=pin)
        assert rel_rms(o, ref) < 4e-3, f"kv_split={pin}"
        assert rel_rms(o, whole) < 1.5 * r_whole, f"kv_split={pin}"


@pytest.mark.gpu
def test_split_launch_is_run_to_run_bit_identical_under_other_traffic():
    """Which unit of a block arrives last changes with timing; the combine walks the units in index order, so the result may not.
    25 launches with a streaming copy and an unrelated GEMM in between, two shapes."""
    from diffusers_amd import ops
    junk = torch.randn(64 << 20, device=DEV)
    a = torch.randn(2048, 2048, device=DEV).to(bf16)
    for (B, H, D, Sq) in ((2, 20, 64, 1024), (1, 24, 128, 4608)):
        q, k, v, kp, vt, sa = _case(B, H, D, Sq, Sq, seed=3)
        pin = 0 if D == 64 else 4
        assert _plan(B, H, Sq, Sq, D, kv_split=pin)[3] > 1
        first = _run(ops, q, kp, vt, B, H, D, Sq, Sq, sa, kv_split=pin).clone()
        for i in range(25):
            if i % 3 == 0:
                junk.mul_(1.0001)
            if i % 2 == 0:
                a @ a
            o = _run(ops, q, kp, vt, B, H, D, Sq, Sq, sa, kv_split=pin)
            assert torch.equal(o, first), f"launch {i} of B{B} H{H} D{D} S{Sq} differs"


@pytest.mark.gpu
def test_split_launch_replays_from_a_hip_graph():
    """The ticket counters are re-armed by the kernel: twenty replays of a captured chain of split launches (two shapes sharing
    the stream's workspace back to back) give the eager result every time."""
    from diffusers_amd import ops
    c1 = _case(2, 20, 64, 1024, 1024, seed=5)
    c2 = _case(2, 10, 64, 4096, 4096, seed=6)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        e1 = _run(ops, c1[0], c1[3], c1[4], 2, 20, 64, 1024, 1024, c1[5]).clone()
        e2 = _run(ops, c2[0], c2[3], c2[4], 2, 10, 64, 4096, 4096, c2[5]).clone()
        o1, o2 = torch.empty_like(e1), torch.empty_like(e2)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(3):
                _run(ops, c1[0], c1[3], c1[4], 2, 20, 64, 1024, 1024, c1[5], out=o1)
                _run(ops, c2[0], c2[3], c2[4], 2, 10, 64, 4096, 4096, c2[5], out=o2)
    torch.cuda.synchronize()
    for i in range(20):
        o1.zero_()
        o2.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(o1, e1) and torch.equal(o2, e2), f"replay {i}"


@pytest.mark.gpu
@pytest.mark.parametrize("algo", [2, 3, 4, 5])
def test_split_units_with_spiked_keys_and_score_offsets(algo):
    """The partials of a block's units carry DIFFERENT running shifts m (each unit has only seen its own keys; the deferred
    maximum lets a unit's shift lag its true maximum).  Keys with a strong common offset that drifts upward along the sequence, one
    spike inside the second unit's range and one inside the last: the combine must weight the units by 2^(m_i - max m)."""
    from diffusers_amd import ops
    B, H, D, Sq, Skv = 2, 20, 64, 1024, 1024
    g = torch.Generator("cpu").manual_seed(11)
    C_ = H * D
    q = torch.randn((B * Sq, C_), generator=g).to(DEV)
    k = torch.randn((B, Skv, C_), generator=g).to(DEV)
    v = torch.randn((B, Skv, C_), generator=g).to(bf16).to(DEV)
    u = torch.nn.functional.normalize(torch.randn(C_, generator=g), dim=0).to(DEV)
    q = q + 6.0 * u
    drift = torch.linspace(-1.0, 1.0, Skv, device=DEV)[None, :, None]
    k = k + (3.0 * drift) * u * (D ** 0.5) / H
    qn = q / q.norm(dim=-1, keepdim=True)
    k[0, 300] += qn[5] * 25.0                     # inside unit 1 of the four (keys 256..511)
    k[1, 1000] += qn[Sq + 77] * 30.0              # inside the last unit
    q, k = q.to(bf16), k.to(bf16)
    kp = k.contiguous()
    vt = torch.empty((C_, B * Skv), device=DEV, dtype=bf16)
    vt.view(C_, B, Skv).copy_(v.permute(2, 0, 1))
    ref = _sdpa(q, k, v, B, H, D, Sq, Skv)
    whole = _run(ops, q, kp, vt, B, H, D, Sq, Skv, Skv, kv_split=1, algo=algo)
    split = _run(ops, q, kp, vt, B, H, D, Sq, Skv, Skv, kv_split=4, algo=algo)
    rw, rs = rel_rms(whole, ref), rel_rms(split, ref)
    print(f"[split] spiked / drifting keys, algo {algo}: rel-rms vs fp32 whole {rw:.3e}, split x4 {rs:.3e}")
    assert torch.isfinite(split.float()).all()
    assert rs <= 1.1 * rw + 2e-4 and rs < 3e-2


# ---- CPU model of the kernel's block decode (attention2.hip, "which logical block, which key tiles") -------------------------------
def _decode_block(bid, qtiles, ntiles, plan):
    """Python restatement of the kernel's scalar block decode for the balanced mapping: -> (pair, query tile, first key tile, end key
    tile, unit id or -1, tail-block id, XCD the block runs on = physical id & 7)."""
    full, tail, s, nb8 = plan
    tps = (ntiles + s - 1) // s if s > 1 else ntiles
    xcd, slot = bid & 7, bid >> 3
    t0, t1, unit, rblk = 0, ntiles, -1, 0
    if s > 1 and bid >= full:
        u = bid - full
        v, xcd = u >> 3, u & 7
        vb, si = v // s, v % s
        rblk = (vb << 3) | xcd
        unit = rblk * s + si
        slot = (full >> 3) + vb
        t0, t1 = si * tps, min(ntiles, si * tps + tps)
    lb = xcd * nb8 + slot
    return lb // qtiles, lb % qtiles, t0, t1, unit, rblk, (bid & 7)


@pytest.mark.parametrize("B,H,Sq,Skv,D,pin", [(2, 20, 1024, 1024, 64, 0), (2, 10, 4096, 4096, 64, 0), (1, 24, 4608, 4608, 128, 4),
                                             (2, 8, 1280, 700, 64, 0), (1, 16, 2560, 1100, 128, 4), (1, 3, 1000, 1024, 64, 0),
                                             (2, 20, 1024, 1000, 64, 3), (2, 5, 2048, 2048, 64, 0), (2, 12, 32760, 32760, 128, 0),
                                             (4, 10, 4096, 520, 64, 2), (2, 20, 1024, 77, 64, 0)])
def test_every_query_block_and_key_tile_is_covered_exactly_once(B, H, Sq, Skv, D, pin):
    """The grid the library launches (whole blocks, then `s` units per tail block) walks every (batch, head, query tile, key tile)
    exactly once; the units of one tail block have distinct unit ids, share its tail-block id (their ticket counter) and run on the
    XCD that owns the block; every XCD owns the same number of logical blocks."""
    need, full, tail, s = _plan(B, H, Sq, Skv, D, kv_split=pin)
    QT = 256 if (D == 128 and B * H * ((Sq + 255) // 256) >= 192) else 128
    qtiles, ntiles = (Sq + QT - 1) // QT, (Skv + 63) // 64
    nb = B * H * qtiles
    if nb % 8:
        pytest.skip("legacy mapping (block count not a multiple of the XCD count): whole blocks only")
    if s <= 1:
        full, tail, s = nb, 0, 1
    plan = (full, tail, s, nb // 8)
    seen, per_xcd, units_of, xcd_of = {}, [0] * 8, {}, {}
    for bid in range(full + tail * s):
        pair, qt, t0, t1, unit, rblk, xcd = _decode_block(bid, qtiles, ntiles, plan)
        assert 0 <= pair < B * H and 0 <= qt < qtiles and 0 <= t0 < t1 <= ntiles, (bid, pair, qt, t0, t1)
        for t in range(t0, t1):
            assert (pair, qt, t) not in seen, f"key tile {t} of block ({pair}, {qt}) walked twice (blocks {seen[(pair, qt, t)]} and {bid})"
            seen[(pair, qt, t)] = bid
        if unit < 0:
            per_xcd[xcd] += 1
        else:
            units_of.setdefault(rblk, []).append(unit)
            assert xcd_of.setdefault((pair, qt), xcd) == xcd, "the units of one block run on different XCDs"
            assert unit // s == rblk
    assert len(seen) == B * H * qtiles * ntiles
    assert len(set(per_xcd)) == 1, per_xcd                       # whole blocks: the same count on every XCD
    assert len(units_of) == tail and all(sorted(u) == list(range(r * s, r * s + s)) for r, u in units_of.items())
    if tail:
        assert max(units_of) < 4096                              # ticket counters: DA_ATTN_SPLIT_COUNTER_BYTES / 4
