// Flash attention forward, second kernel generation ("v2") for gfx950: unmasked, D = 64 / 128, bf16 in / bf16 out.
//
// Same call sites as attention.hip (F.scaled_dot_product_attention at diffusers models/attention_processor.py:2767 and
// models/attention_dispatch.py:3709) and the same swapped products
//
//   S^T[kv][q] = K[kv][:] . Q[q][:]^T      A = K tile (LDS), B = Q (registers)         v_mfma_f32_32x32x16_bf16
//   O^T[d][q]  = V^T[d][kv] . P^T[kv][q]   A = V^T tile (LDS), B = P^T (registers, straight from the S^T accumulators)
//
// so that a lane owns ONE query.  What changed against the round-1..3 kernel (profiles/r03d_pmc_sq_counters.md: 49 % of the
// wave cycles parked, LDS conflicts 0.35 of LDS-active, ~250 issued instructions around 16 MFMAs per 64-key tile):
//
//  * K rows are fed to the Q.K^T product PERMUTED (A-row i takes key pi(i), pi = swap of bits 2 and 3): the C/D layout of the
//    32 x 32 MFMA then leaves every lane with eight CONSECUTIVE keys per register octet, the P^T B-fragment of the P.V product
//    is in natural key order, and a V^T A-fragment is ONE conflict-free ds_read_b128 (it was two 2-way-conflicting
//    ds_read_b64 plus register shuffles).
//  * K / V^T tiles arrive by BUFFER-addressed LDS-DMA: per-lane byte offsets are loop constants, the tile advance is one
//    scalar offset, chunks past the end of the sequence set bit 31 of the offset (the range check writes zeros): ~12
//    instructions per tile where the pointer-select form issued ~50.
//  * Deferred maximum: the O / l rescale runs only when some row's score exceeds the running shift by more than 2^THR
//    (wave vote); otherwise P = exp2(s * c - m) with the OLD shift, bounded by 2^THR.  The first tile always takes the
//    branch (m starts at -1e30).  The decision is taken before tile j's exponentials and the O rescale is applied after
//    tile j - 1's P.V product has been added (the order of the previous kernel).
//  * The row sum is kept in four partial accumulators inside the softmax slices (the compiler used to sink all 32 adds
//    into one dependent chain behind the last MFMA); the cross-half exchanges are v_permlane32_swap, not LDS permutes.
//  * O leaves through LDS as whole rows, 16 bytes per lane (the 8-byte-per-lane MFMA layout touched 32 lines per store).
//  * NW = 4 or 8 waves per workgroup (128 / 256 queries share one K / V^T stream): D = 128 gets two waves per SIMD.
//
// Numerics: fp32 scores, fp32 softmax statistics, P rounded to bf16 for the second product, fp32 O accumulation -- as before;
// results differ from the previous kernel in the last bits (different shift m, different summation order of l).
#include <type_traits>

#include <cstdlib>

#include "common.cuh"

namespace da_attn2 {

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, size_t bytes) {
  const uint64_t v = (uint64_t)base;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  const int n = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffffull ? 0x7fffffffull : bytes));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}
#endif

constexpr float kDeferLog2 = 8.0f;   // THR: P stays <= 2^8 between rescales

// Block -> (batch, head, query tile) mapping and the key-split tail (round 6; include/diffusers_amd.h da_attention_params.split_ws).
// nb8 > 0 ("balanced" mapping; nb = B * H * qtiles is a multiple of 8): the grid is exactly nb blocks (+ the split units), the
// LOGICAL blocks are numbered pair-major (pair * qtiles + query tile) and XCD x (= physical block id & 7) owns the contiguous run
// [x * nb8, (x + 1) * nb8): every XCD gets the same number of blocks and sees at most nb8 / qtiles + 2 different (batch, head)
// pairs' K / V^T in its L2.  (The legacy mapping, nb8 == 0, gives every XCD whole pairs: with 20 pairs -- SDXL's 64 x 64 level -- four
// XCDs run three pairs and four run two, and a fifth of the grid's blocks exit at once.)
// Key split: physical blocks [0, full) are whole logical blocks (slot b >> 3 of XCD b & 7); behind them come `s` UNITS for each
// of the `tail` remaining logical blocks, unit si of a block walking key tiles [si * tps, min(ntiles, (si + 1) * tps)).  Unit
// u = blockIdx.x - full sits on XCD u & 7, the XCD that owns the logical block it belongs to.  s == 1: no split.
struct SplitPlan {
  int full, tail, s, tps, nb8;
};
constexpr int kSplitCounterBytes = DA_ATTN_SPLIT_COUNTER_BYTES;
template <int D, int NW>
constexpr int split_unit_bytes() { return NW * (D * 128 + 512); }   // per wave: 32 queries x D fp32 of O, then (m, l) per lane

template <int D>
struct Cfg {
  static constexpr int CPR = D / 8;              // 16-byte chunks per K row
  static constexpr int KBYTES = 64 * D * 2;      // K tile  [64][D]
  static constexpr int VBYTES = D * 128;         // V^T tile [D][64]
  static constexpr int STAGE = KBYTES + VBYTES;
};

// 16-byte-slot XOR of K row `row` (source side): conflict-free ds_read_b128 for the permuted rows of every lane group
template <int D>
__device__ __forceinline__ int k_swz(int row) {
  return D == 64 ? (row >> 1) & 7 : row & 15;
}

// AUG: the shift -m of the online softmax rides in the Q.K^T product as one extra k-step (A = a ones column, B = -m in bf16),
// with Q pre-multiplied by scale * log2(e): the scores leave the MFMA as s * c - m and the 32 v_fma_f32 per tile in front of the
// exponentials disappear (two more MFMAs per tile instead).  m is kept bf16-exact; the softmax is invariant to the shift, so
// its rounding is harmless; the pre-multiplied Q is rounded to bf16 once more than the reference's (scores within bf16 noise).
// RSM: the row sum l of the online softmax comes out of the matrix pipe: one more A fragment of ones next to the V^T fragments of every
// P.V k-step accumulates sum_k P^T[k][q] into a 17th / 33rd accumulator tile (every row of it holds the row sums), so the 32 v_add_f32
// per tile disappear (4 more MFMAs per tile instead).  l is then the sum of the bf16-ROUNDED probabilities -- exactly the weights
// the second product applies to V -- and it is rescaled with O when the running shift moves.
// PRIO = 1 (round 5): one s_setprio pair per tile -- priority 1 from the end of the Q.K^T block to the end of the tile's softmax
// slices (which carry the previous tile's P.V MFMAs), priority 0 for the rescale, the rendezvous, the K-fragment reads, the LDS-DMA
// issue and the Q.K^T MFMAs.  With three workgroups per CU a SIMD holds three waves of three workgroups; the arbitration then lets a
// wave that is inside its softmax / P.V phase run ahead of a wave that is starting a tile, which takes the co-resident waves out of
// step: one wave's transcendental-heavy VALU phase sits beside another's MFMA block instead of beside the same phase of its peers.
// Measured (profiles/r05f_attention_setprio.jsonl, chained launches, off / on / off / on on one box): S 1024 24.6 -> 22.1 us, S 4096
// 126.4 -> 118.4 us at D = 64 (another box: 24.2 -> 21.6, 121.6 -> 112.7),
// nothing at D = 128 (one eight-wave workgroup per CU); eleven other placements (P.V MFMAs alone, VALU slices alone, graded by
// slice, the memory head raised, static per-workgroup levels) gain less or lose.  Speed only: the same operations in the same order.
template <int D, int NW, int NS, bool AUG, bool RSM, int PRIO = 0>
// min waves per SIMD: D = 64 / four waves: three workgroups per CU (<= 168 registers); eight waves: one workgroup = two per SIMD
__global__ __launch_bounds__(64 * NW, (D == 64 && NW == 4) ? 3 : (NW == 8 ? 2 : 1)) void attn2_fwd_kernel(const da_attention_params p, const SplitPlan sp) {
#if defined(__HIP_DEVICE_COMPILE__)
  using C = Cfg<D>;
  static_assert(D == 64 || D == 128, "head sizes of the v2 kernel");
  static_assert(NW == 4 || NW == 8, "four or eight waves");
  static_assert(NS == 3 || NS == 4, "ring of 3 or 4 tiles (tile j - 1's V^T stays resident while tile j is consumed)");
  constexpr int QT = 32 * NW;                    // queries per workgroup
  constexpr int PD = NS - 2;                     // tiles in flight ahead of the one being consumed
  constexpr int KCH = C::CPR / NW;               // K pieces (1 KiB) per wave per tile
  constexpr int VCH = (D / 8) / NW;              // V^T pieces per wave per tile
  static_assert(KCH >= 1 && VCH >= 1 && KCH * NW == C::CPR && VCH * NW == D / 8, "pieces divide among the waves");
  constexpr int LOADS = KCH + VCH;
  constexpr int DT = D / 32, NPV = 4 * DT, NQK = 2 * (D / 16), PER = NPV / 8;
  constexpr int RING = NS * C::STAGE;
  static_assert(RING <= 160 * 1024, "K / V^T ring exceeds the LDS of a CU");
  static_assert(NW * 32 * (2 * D + 16) <= RING, "output staging does not fit the ring");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  const int qtiles = (p.Sq + QT - 1) / QT;
  const int ntiles = (p.Skv + 63) >> 6;
  // which logical block, which key tiles (all wave-uniform scalars)
  int t_begin = 0, t_end = ntiles, unit = -1, rblk = 0, pair, qt;
  if (sp.nb8 > 0) {
    int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    if (sp.s > 1 && bid >= sp.full) {
      const int u = bid - sp.full, v = u >> 3, vb = v / sp.s, si = v - vb * sp.s;
      xcd = u & 7;
      rblk = (vb << 3) | xcd;
      unit = rblk * sp.s + si;
      slot = (sp.full >> 3) + vb;
      t_begin = si * sp.tps;
      t_end = min(ntiles, t_begin + sp.tps);
    }
    const int lb = xcd * sp.nb8 + slot;
    pair = lb / qtiles;
    qt = lb - pair * qtiles;
  } else {
    // legacy mapping: all query tiles of one (batch, head) pair go to ONE XCD (attention.hip)
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    pair = (slot_id / qtiles) * 8 + xcd;
    qt = slot_id % qtiles;
    if (pair >= p.B * p.H) return;
  }
  const int b = pair / p.H, h = pair - b * p.H;
  const int q0 = qt * QT + wave * 32;

  const uint16_t* __restrict__ Q = (const uint16_t*)p.q + (size_t)b * p.q_batch_stride + (size_t)h * D;
  const uint16_t* __restrict__ K = (const uint16_t*)p.k + (size_t)b * p.k_batch_stride + (size_t)h * D;
  const uint16_t* __restrict__ VT = (const uint16_t*)p.vt + (size_t)h * D * p.vt_ld + (size_t)b * p.vt_batch_stride;
  uint16_t* __restrict__ O = (uint16_t*)p.out + (size_t)b * p.o_batch_stride + (size_t)h * D;
  const float sl2 = p.scale * 1.4426950408889634f;

  // ---- staging: buffer-addressed LDS-DMA, one 1 KiB piece (64 lanes x 16 B) per wave instruction ----
  // piece pi = i * NW + wave; K: chunk pch = 64 pi + lane -> (row = pch / CPR, slot = pch % CPR), source chunk slot ^ k_swz(row);
  // V^T: (d = pch / 8, slot = pch % 8), source chunk slot ^ ((d >> 1) & 7).  Offsets are bytes from the (batch, head) base;
  // the tile advance is the scalar offset.  Only the LAST tile can reach past Skv_alloc: its offsets (vl_*) carry bit 31 on
  // the chunks that do (beyond num_records: the range check writes zeros to LDS without touching memory).
  const __amdgpu_buffer_rsrc_t rs_k = uniform_rsrc(K, 0x7fffffff);
  const __amdgpu_buffer_rsrc_t rs_v = uniform_rsrc(VT, 0x7fffffff);
  int vo_k[KCH], vl_k[KCH], vo_v[VCH], vl_v[VCH];
  {
    const int rem_last = p.Skv_alloc - (ntiles - 1) * 64;      // keys of the last tile that exist
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int pch = (i * NW + wave) * 64 + lane;
      const int row = pch / C::CPR, slot = pch % C::CPR;
      vo_k[i] = (row * p.k_row_stride + ((slot ^ k_swz<D>(row)) << 3)) * 2;
      vl_k[i] = vo_k[i] | (row >= rem_last ? (int)0x80000000 : 0);
    }
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
      const int pch = (i * NW + wave) * 64 + lane;
      const int d = pch >> 3, slot = pch & 7;
      const int c = slot ^ ((d >> 1) & 7);
      vo_v[i] = (d * p.vt_ld + c * 8) * 2;
      vl_v[i] = vo_v[i] | (c * 8 >= rem_last ? (int)0x80000000 : 0);   // Skv_alloc % 8 == 0: a chunk is all in or all out
    }
  }
#define DA_LDS(ptr) ((__attribute__((address_space(3))) void*)(ptr))
  auto issue = [&](int tile, int buf_off) {
    unsigned char* kb = smem + buf_off;
    unsigned char* vb = kb + C::KBYTES;
    const int so_k = tile * 64 * p.k_row_stride * 2, so_v = tile * 128;
    if (tile == ntiles - 1) {                                   // wave-uniform
#pragma unroll
      for (int i = 0; i < KCH; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, DA_LDS(kb + (i * NW + wave) * 1024), 16, vl_k[i], so_k, 0, 0);
#pragma unroll
      for (int i = 0; i < VCH; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, DA_LDS(vb + (i * NW + wave) * 1024), 16, vl_v[i], so_v, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < KCH; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, DA_LDS(kb + (i * NW + wave) * 1024), 16, vo_k[i], so_k, 0, 0);
#pragma unroll
      for (int i = 0; i < VCH; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, DA_LDS(vb + (i * NW + wave) * 1024), 16, vo_v[i], so_v, 0, 0);
    }
  };
#undef DA_LDS
  // ---- prologue: tiles 0 .. PD - 1 in flight before anything else ----
#pragma unroll
  for (int t0 = 0; t0 < PD; ++t0)
    if (t_begin + t0 < t_end) issue(t_begin + t0, t0 * C::STAGE);

  // ---- Q fragments (MFMA B operand): lane (q = l31, hi) holds Q[q][16 ks + 8 hi + 0..7] ----
  bf16x8_t qf[D / 16];
  {
    const int q = q0 + l31;
    const bool ok = q < p.Sq;
    const uint16_t* qp = Q + (size_t)(ok ? q : 0) * p.q_row_stride + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      uint4 v = *(const uint4*)(qp + 16 * ks);
      if (!ok) v = make_uint4(0, 0, 0, 0);
      if constexpr (AUG) {
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= sl2;
        v = pack8(f);
      }
      qf[ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  }
  // AUG: the extra k-step.  A = K side: 1.0 at k = 0 for every key row; B = Q side: -m of the lane's query at k = 0.
  // (the A side is all ones: the B side is zero everywhere but at k = 0 of the hi = 0 lanes, so the step still contributes -m only;
  //  the same fragment feeds the row-sum product of RSM)
  bf16x8_t qx = __builtin_bit_cast(bf16x8_t, make_uint4(0u, 0u, 0u, 0u));
  const bf16x8_t ones8 = __builtin_bit_cast(bf16x8_t, make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u));

  f32x16_t o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  f32x16_t osum;             // RSM: row sums, accumulated by the matrix pipe
#pragma unroll
  for (int r = 0; r < 16; ++r) osum[r] = 0.f;
  float m_run = AUG ? 0.f : -1e30f, l_run = 0.f;

  // fragment addresses (bytes inside a ring slot): 4 K and 4 V^T lane-constant offsets, the rest are immediates
  const int pi_row = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);   // key row this lane feeds as A-row l31
  const int ksw = k_swz<D>(pi_row);
  const int kbase = pi_row * (2 * D);
  const int vsw = (l31 >> 1) & 7;
  const int vbase = C::KBYTES + l31 * 128;
  auto kfrag = [&](const unsigned char* sb, int k) {            // Q.K^T MFMA k = 2 ks + st
    const int ks = k >> 1, st = k & 1;
    return *(const bf16x8_t*)(sb + kbase + st * (32 * 2 * D) + (((2 * ks + hi) ^ ksw) << 4));
  };
  auto vfrag = [&](const unsigned char* sb, int k) {            // P.V MFMA k = u * DT + dt: keys 16 u + 8 hi + 0..7 of row 32 dt + l31
    const int u = k / DT, dt = k % DT;
    return *(const bf16x8_t*)(sb + vbase + dt * 4096 + (((2 * u + hi) ^ vsw) << 4));
  };
  auto xhalf_max = [](float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  };

  bf16x8_t pf[4];            // P^T fragments of the previous tile, consumed one iteration late
  int cur = 0, prv = (NS - 1) * C::STAGE, nxt = PD * C::STAGE;   // ring offsets of tile j, tile j - 1, tile j + PD
  // One iteration: [rendezvous, DMA of tile j + PD, Q.K^T(j), softmax(j) in eight slices with P.V(j - 1) riding in them].
  auto iter = [&](int j, auto has_prev_c) __attribute__((always_inline)) {
    constexpr bool HAS_PREV = decltype(has_prev_c)::value;
    // issued so far: tiles t_begin .. min(t_end, j + PD) - 1; tile j must have landed, the later ones may stay in flight
    if (PD >= 2 && j + 1 < t_end) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // past the rendezvous every wave has finished iteration j - 1, i.e. its reads of tile j - 2's V^T: that slot is free
    const unsigned char* sb = smem + cur;
    const unsigned char* sbp = smem + prv;
    f32x16_t s[2];
    {
      // K fragments in batches of KB (all of them where the registers allow): the reads the first MFMAs wait for go out first
      constexpr int KB = (AUG && RSM && D == 64 && NW == 4) ? NQK / 2 : (D == 128 ? NQK / 2 : NQK);
      bf16x8_t kf[KB];
#pragma unroll
      for (int k = 0; k < KB; ++k) kf[k] = kfrag(sb, k);
      if (j + PD < t_end) issue(j + PD, nxt);

      const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if constexpr (AUG) {
        s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, qx, zero16, 0, 0, 0);
        s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, qx, zero16, 0, 0, 0);
      }
#pragma unroll
      for (int k0 = 0; k0 < NQK; k0 += KB) {
#pragma unroll
        for (int k = k0; k < k0 + KB; ++k) {
          s[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[k - k0], qf[k >> 1], (!AUG && k < 2) ? zero16 : s[k & 1], 0, 0, 0);
          if (k + KB < NQK) kf[k - k0] = kfrag(sb, k + KB);       // the next batch's fragment takes the register just consumed
        }
      }
    }
    // PRIO: from here to the end of the tile (softmax slices with tile j - 1's P.V MFMAs riding in them) the wave outranks waves that
    // are still in their rendezvous / K-fragment / Q.K^T head -- see the template comment
    if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_sched_barrier(0);
    // ragged last tile: keys past Skv never win the maximum and exponentiate to exactly 0 (wave-uniform branch)
    const int kv0 = j * 64;
    if (kv0 + 64 > p.Skv) {
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + 32 * st + 16 * (r >> 3) + 8 * hi + (r & 7);
          s[st][r] = (kv >= p.Skv) ? -1e30f : s[st][r];
        }
    }
    float alpha = 1.f;
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
    auto pack_p = [&](int u) {
      const int b8 = 8 * (u & 1);
      const uint4 pk = make_uint4(pack_bf2(s[u >> 1][b8 + 0], s[u >> 1][b8 + 1]), pack_bf2(s[u >> 1][b8 + 2], s[u >> 1][b8 + 3]),
                                  pack_bf2(s[u >> 1][b8 + 4], s[u >> 1][b8 + 5]), pack_bf2(s[u >> 1][b8 + 6], s[u >> 1][b8 + 7]));
      pf[u] = __builtin_bit_cast(bf16x8_t, pk);
    };
    auto slice = [&](int sl) {
      if (sl == 0) {
        float mx0 = fmaxf(s[0][0], s[0][1]), mx1 = fmaxf(s[1][0], s[1][1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) {
          mx0 = fmaxf(fmaxf(mx0, s[0][r]), s[0][r + 1]);
          mx1 = fmaxf(fmaxf(mx1, s[1][r]), s[1][r + 1]);
        }
        const float mxr = xhalf_max(fmaxf(mx0, mx1));            // row maximum of this tile
        if constexpr (AUG) {
          // scores are already s * c - m.  Deferred maximum: keep the shift while no row of the wave outgrows it by 2^THR;
          // the first tile always sets it (m = 0 so far: the scores may sit far below as well as above).
          if (__builtin_amdgcn_ballot_w64(!HAS_PREV || mxr > kDeferLog2) != 0) {
            const float delta = HAS_PREV ? fmaxf(mxr, 0.f) : mxr;
            const float m_new = bf2f(f2bf(m_run + delta));       // bf16-exact: it rides in the next tiles' MFMA
            const float d_eff = m_new - m_run;
            alpha = __builtin_amdgcn_exp2f(-d_eff);
            m_run = m_new;
            qx = __builtin_bit_cast(bf16x8_t, make_uint4(hi == 0 ? (uint32_t)f2bf(-m_new) : 0u, 0u, 0u, 0u));
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
              for (int r = 0; r < 16; ++r) s[st][r] -= d_eff;
          }
        } else {
          const float mx = mxr * sl2;                            // scaled (sl2 > 0)
          // deferred maximum: keep the old shift while no row of the wave outgrows it by more than 2^THR (m starts at -1e30:
          // the first tile always takes the branch)
          if (__builtin_amdgcn_ballot_w64(mx - m_run > kDeferLog2) != 0) {
            const float m_new = fmaxf(m_run, mx);
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
          }
        }
      } else if (sl >= 1 && sl <= 4) {
        const int st = (sl - 1) >> 1, r0 = 8 * ((sl - 1) & 1);
#pragma unroll
        for (int r = r0; r < r0 + 8; ++r) {
          const float e = AUG ? __builtin_amdgcn_exp2f(s[st][r]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[st][r], sl2, -m_run));
          s[st][r] = e;
          if constexpr (!RSM) ps[r & 3] += e;
        }
        if constexpr (!RSM) {
#pragma unroll
          for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(ps[u]));   // the adds stay in this slice
        }
      } else if (sl >= 5 && sl <= 6) {
        // P^T of tile j is packed straight into `pf`: fragment u of tile j - 1 is consumed by the P.V MFMAs of slices 2u and 2u + 1,
        // so fragments 0 / 1 are free from slice 4 on, fragment 2 from slice 6 on, fragment 3 behind slice 7 (packed after the loop)
#pragma unroll
        for (int u = (sl == 5 ? 0 : 2); u < (sl == 5 ? 2 : 3); ++u) pack_p(u);
      } else if (sl == 7) {
        if constexpr (!RSM) l_run = l_run * alpha + ((ps[0] + ps[1]) + (ps[2] + ps[3]));
      }
    };
    if constexpr (HAS_PREV) {
      bf16x8_t av[PER], an[PER];
#pragma unroll
      for (int q = 0; q < PER; ++q) av[q] = vfrag(sbp, q);
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) {
#pragma unroll
        for (int q = 0; q < PER; ++q)
          if (sl < 7) an[q] = vfrag(sbp, (sl + 1) * PER + q);     // next slice's V^T fragments fly under this slice
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          const int k = sl * PER + q;
          o[k % DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[q], pf[k / DT], o[k % DT], 0, 0, 0);
          if constexpr (RSM) {
            if (k % DT == 0) osum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, pf[k / DT], osum, 0, 0, 0);
          }
        }
        slice(sl);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < PER; ++q) av[q] = an[q];
      }
    } else {
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) {
        slice(sl);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    pack_p(3);
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(pf[u]));
    __builtin_amdgcn_sched_barrier(0);
    // rescale AFTER tile j - 1's product has been added: O_j-1 complete, then * alpha_j, then (next iteration) + P_j V_j.
    // alpha is exactly 1.0f unless the deferred-maximum branch ran (x * 1.0f is exact: same bits either way).
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      if constexpr (RSM) {
#pragma unroll
        for (int r = 0; r < 16; ++r) osum[r] *= alpha;
      }
    }
    prv = cur;
    cur = (cur + C::STAGE == RING) ? 0 : cur + C::STAGE;
    nxt = (nxt + C::STAGE == RING) ? 0 : nxt + C::STAGE;
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  iter(t_begin, F_{});
  for (int j = t_begin + 1; j < t_end; ++j) iter(j, T_{});
  {                                                               // the last tile's product (its slot was not refilled)
    const unsigned char* sbl = smem + prv;
#pragma unroll
    for (int k = 0; k < NPV; ++k) o[k % DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(sbl, k), pf[k / DT], o[k % DT], 0, 0, 0);
    if constexpr (RSM) {
#pragma unroll
      for (int u = 0; u < 4; ++u) osum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, pf[u], osum, 0, 0, 0);
    }
  }

  // ---- epilogue: O^T registers -> LDS (per wave) -> whole rows, 16 bytes per lane ----
  float l_tot;
  if constexpr (RSM) {
    l_tot = osum[0];           // every row of the tile holds the sums over BOTH halves' keys (the MFMA contracts all 16 k of a step)
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_tot = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                   // every wave is out of the ring
  asm volatile("" ::: "memory");
  if (unit >= 0) {
    // ---- key-split tail: publish this unit's (O, m, l), draw a ticket, the last arriver of the block combines all s units ----
    // Hand-off R1 of cdna_hip_programming.md Guideline 16: write-through (sc1) payload stores, every storing wave drains its stores,
    // workgroup barrier, ONE relaxed agent-scope ticket; the combiner reads with sc1 loads.  No polling: a unit that is not last
    // leaves.  The combine runs over the units in index order whichever arrives last (its own partial is re-read like the others),
    // so the block's result is a fixed function of the inputs.
    constexpr int WAVE_BYTES = D * 128 + 512, UNIT_BYTES = NW * WAVE_BYTES;
    static_assert(UNIT_BYTES == split_unit_bytes<D, NW>(), "unit size");
    unsigned char* data = (unsigned char*)p.split_ws + kSplitCounterBytes;
    int* cnt = (int*)p.split_ws;
    {
      const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(data + (size_t)unit * UNIT_BYTES + wave * WAVE_BYTES, WAVE_BYTES);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x4_t v;
          v.x = __float_as_uint(o[dt][4 * g + 0]); v.y = __float_as_uint(o[dt][4 * g + 1]);
          v.z = __float_as_uint(o[dt][4 * g + 2]); v.w = __float_as_uint(o[dt][4 * g + 3]);
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((dt * 4 + g) * 64 + lane) * 16, 0, 16);   // lane-linear KiB pieces, sc1
        }
      typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
      u32x2_t ml;
      ml.x = __float_as_uint(m_run); ml.y = __float_as_uint(l_tot);
      __builtin_amdgcn_raw_buffer_store_b64(ml, rs, D * 128 + lane * 8, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // EVERY storing wave drains before the ticket
    __syncthreads();
    int* tick = (int*)smem;
    if (t == 0) *tick = __hip_atomic_fetch_add(cnt + rblk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(*tick);
    if (ticket != sp.s - 1) return;
    if (t == 0) __hip_atomic_store(cnt + rblk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
    __syncthreads();                                               // `tick` is read; the ring is staging space again
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
    const unsigned char* blk = data + (size_t)(rblk * sp.s) * UNIT_BYTES + wave * WAVE_BYTES;
    float m_all = -3.0e38f;
    for (int i = 0; i < sp.s; ++i) {
      const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(blk + (size_t)i * UNIT_BYTES, WAVE_BYTES);
      const u32x2_t ml = __builtin_amdgcn_raw_buffer_load_b64(rs, D * 128 + lane * 8, 0, 16);
      m_all = fmaxf(m_all, __uint_as_float(ml.x));
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    l_tot = 0.f;
    for (int i = 0; i < sp.s; ++i) {
      const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(blk + (size_t)i * UNIT_BYTES, WAVE_BYTES);
      const u32x2_t ml = __builtin_amdgcn_raw_buffer_load_b64(rs, D * 128 + lane * 8, 0, 16);
      const float w = __builtin_amdgcn_exp2f(__uint_as_float(ml.x) - m_all);
      l_tot = __builtin_fmaf(w, __uint_as_float(ml.y), l_tot);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        u32x4_t v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((dt * 4 + g) * 64 + lane) * 16, 0, 16);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          o[dt][4 * g + 0] = __builtin_fmaf(w, __uint_as_float(v[g].x), o[dt][4 * g + 0]);
          o[dt][4 * g + 1] = __builtin_fmaf(w, __uint_as_float(v[g].y), o[dt][4 * g + 1]);
          o[dt][4 * g + 2] = __builtin_fmaf(w, __uint_as_float(v[g].z), o[dt][4 * g + 2]);
          o[dt][4 * g + 3] = __builtin_fmaf(w, __uint_as_float(v[g].w), o[dt][4 * g + 3]);
        }
      }
    }
  }
  const float inv = 1.0f / l_tot;
  constexpr int OROW = 2 * D + 16;                                // bytes per staged row (pad: the 8-byte writes spread over the banks)
  unsigned char* stg = smem + wave * (32 * OROW);
  if ((p.o_row_stride & 7) == 0 && ((size_t)p.out & 15) == 0 && (p.o_batch_stride & 7) == 0) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack_bf2(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
        pk.y = pack_bf2(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        *(uint2*)(stg + l31 * OROW + (32 * dt + 8 * g + 4 * hi) * 2) = pk;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // wave-private region: no barrier
    constexpr int CPRO = D / 8;                                   // 16-byte pieces per output row
#pragma unroll
    for (int i = 0; i < (32 * CPRO) / 64; ++i) {
      const int pc = i * 64 + lane;
      const int row = pc / CPRO, c = pc % CPRO;
      const uint4 v = *(const uint4*)(stg + row * OROW + c * 16);
      if (q0 + row < p.Sq) *(uint4*)(O + (size_t)(q0 + row) * p.o_row_stride + c * 8) = v;
    }
  } else {
    const int q = q0 + l31;
    if (q < p.Sq) {
      uint16_t* op = O + (size_t)q * p.o_row_stride;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 pk;
          pk.x = pack_bf2(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
          pk.y = pack_bf2(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
          *(uint2*)(op + 32 * dt + 8 * g + 4 * hi) = pk;
        }
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// The key-split plan of a launch with QT queries per workgroup and `unit_bytes` of workspace per unit: how many blocks run whole,
// how many are split and how.  Cost model in key tiles per CU: the whole blocks take (full / CUs) * ntiles everywhere; the tail adds
// ceil(tail * s / CUs) rounds of ceil(ntiles / s) tiles, plus OVH tiles once for a split (prologue, publish, combine).  s = 1 is the
// unsplit launch (its tail costs a whole extra round of ntiles).  Units are whole tiles, at least four of them, and none is empty.
inline int cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}
inline SplitPlan split_plan(const da_attention_params& p, int QT, int unit_bytes, long long* ws_bytes) {
  if (ws_bytes) *ws_bytes = 0;
  const int npairs = p.B * p.H, qtiles = (p.Sq + QT - 1) / QT, ntiles = (p.Skv + 63) >> 6;
  const long long nbl = (long long)npairs * qtiles;
  if ((nbl & 7) || nbl > 0x3fffffff) return SplitPlan{0, 0, 1, 0, 0};         // legacy mapping, whole blocks
  const int nb = (int)nbl;
  const SplitPlan none = {nb, 0, 1, 0, nb >> 3};                              // balanced mapping, whole blocks
  if (p.kv_split == 1 || ntiles < 8) return none;
  const int cus = cu_count();
  int tail = nb % cus, full = nb - tail;
  auto valid = [&](int s) { const int tps = (ntiles + s - 1) / s; return s >= 2 && s <= 8 && tps >= 4 && (s - 1) * tps < ntiles; };   // units of >= 256 keys
  // Experiment (round 6, second session; DA_ATTN_SPLIT_ALL=<s>): a D = 64 launch with at most two 128-query workgroups per CU (SDXL's
  // S = 1024: 320 blocks, 1.25 waves per SIMD -- a lone wave runs a key tile in 0.80 us against 0.61 at three waves per SIMD) splits
  // the keys of EVERY block over s units: s times the waves, each walking 1 / s of the keys, the partials combined by the last arriver.
  // MEASURED, a loss (profiles/r06g_attention_split_all.jsonl, chained launches): S = 1024, 2 x 20 heads 20.0 us (tail split) -> 23.5 / 27.9 /
  // 31.3 us at s = 2 / 3 / 4; SD1.5's 2 x 8 heads at S = 4096 79.3 -> 104 / 89 / 97; SDXL image 0.9965 / 1.0016 -> 0.9909 / 0.9935 (s = 2),
  // 0.9692 / 0.9739 (s = 4) -- the partials' round trip through memory (35 KB per unit, write-through, read back by the last arriver)
  // costs more than the extra waves per SIMD buy.  Off by default; what it would take instead is a combine inside the workgroup.
  static const int split_all = [] { const char* v = getenv("DA_ATTN_SPLIT_ALL"); return v ? atoi(v) : 0; }();
  if (split_all >= 2 && p.kv_split == 0 && p.D == 64 && QT == 128 && nb <= 2 * cus && !(cus & 7) && valid(split_all) &&
      (size_t)nb * 4 <= (size_t)kSplitCounterBytes) {
    const long long need_all = (long long)kSplitCounterBytes + (long long)nb * split_all * unit_bytes;
    if (ws_bytes) *ws_bytes = need_all;
    if (p.split_ws && p.split_ws_bytes >= need_all) return SplitPlan{0, nb, split_all, (ntiles + split_all - 1) / split_all, nb >> 3};
  }
  if (tail == 0 || (cus & 7) || (size_t)tail * 4 > (size_t)kSplitCounterBytes) return none;
  int best = 1;
  if (p.kv_split >= 2) {
    if (!valid(p.kv_split)) return none;
    best = p.kv_split;
  } else if (p.D != 64) {
    // D = 128 (eight-wave workgroups, one per CU, 135 KB of partials per unit): measured on Flux's joint attention (432 blocks: 256 + 176)
    // every split factor that fits the workspace LOSES -- 251-255 us whole, 278 / 284 / 269 us at 2 / 3 / 4 units
    // (profiles/r06_attention.jsonl): the partials' round trip costs more than the shorter last round saves.  Pinned splits only.
    return none;
  } else {
    const double OVH = 2.0;
    double best_cost = (double)ntiles;                      // s = 1: one more round of whole blocks
    for (int s = 2; s <= 8; ++s) {
      if (!valid(s)) continue;
      const int tps = (ntiles + s - 1) / s, rounds = (tail * s + cus - 1) / cus;
      const double cost = (double)rounds * tps + OVH;
      if (cost < best_cost - 0.5) best_cost = cost, best = s;
    }
    if (best == 1) return none;
  }
  const long long need = (long long)kSplitCounterBytes + (long long)tail * best * unit_bytes;
  if (ws_bytes) *ws_bytes = need;
  if (!p.split_ws || p.split_ws_bytes < need) return none;
  return SplitPlan{full, tail, best, (ntiles + best - 1) / best, nb >> 3};
}

template <int D, int NW, int NS, bool AUG, bool RSM, int PRIO>
int launch_prio(const da_attention_params& p, hipStream_t s) {
  using C = Cfg<D>;
  const size_t lds = (size_t)NS * C::STAGE;
  auto kern = attn2_fwd_kernel<D, NW, NS, AUG, RSM, PRIO>;
  static bool attr_set = false;   // per instantiation
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DA_ERR_LAUNCH;
    attr_set = true;
  }
  const int qtiles = (p.Sq + 32 * NW - 1) / (32 * NW), rounds = (p.B * p.H + 7) / 8;
  const SplitPlan sp = split_plan(p, 32 * NW, split_unit_bytes<D, NW>(), nullptr);
  const int grid = sp.nb8 > 0 ? sp.full + sp.tail * sp.s : 8 * rounds * qtiles;
  DA_LAUNCH(kern, dim3(grid), dim3(64 * NW), lds, s, p, sp);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// D = 64: the priority pair of the template comment (default on; DA_ATTN2_PRIO=0 switches it off for A/B runs).
template <int D, int NW, int NS, bool AUG, bool RSM>
int launch(const da_attention_params& p, hipStream_t s) {
  if constexpr (D == 64) {
    static const int prio = [] { const char* v = getenv("DA_ATTN2_PRIO"); return v ? atoi(v) : 1; }();
    if (prio == 1) return launch_prio<D, NW, NS, AUG, RSM, 1>(p, s);
  }
  return launch_prio<D, NW, NS, AUG, RSM, 0>(p, s);
}

}  // namespace da_attn2

// Entry used by da_attention_bf16 (attention.hip).  Returns DA_ERR_UNSUPPORTED for anything this generation does not cover
// (masked / biased attention, D = 96 / 160, 32-bit offset overflow): the caller then runs the first-generation kernel.
int da_attn2_dispatch(const da_attention_params& p, hipStream_t s) {
  if (p.bias || p.causal) return DA_ERR_UNSUPPORTED;
  if (p.D != 64 && p.D != 128) return DA_ERR_UNSUPPORTED;
  if (p.ring_slots != 0 && p.ring_slots != 3 && p.ring_slots != 4) return DA_ERR_UNSUPPORTED;
  if (p.q_block != 0 && p.q_block != 128 && p.q_block != 256) return DA_ERR_UNSUPPORTED;
  // 31-bit byte offsets of the buffer-addressed staging
  if ((size_t)p.Skv_alloc * (size_t)p.k_row_stride * 2 + 64ull * p.k_row_stride * 2 >= 0x7fffffffull) return DA_ERR_UNSUPPORTED;
  if ((size_t)p.D * (size_t)p.vt_ld * 2 + (size_t)p.Skv_alloc * 2 >= 0x7fffffffull) return DA_ERR_UNSUPPORTED;
  if (p.kv_split < 0 || p.kv_split > 8) return DA_ERR_INVALID;
  int ns = p.ring_slots ? p.ring_slots : 3;
  // experiment (round 6): a fourth ring slot (two key tiles in flight ahead instead of one) for D = 64 launches with at most two
  // 128-query workgroups per CU -- LDS then admits two workgroups per CU, which is all such a launch has.  DA_ATTN2_RING4=1 enables.
  if (p.ring_slots == 0 && p.D == 64) {
    static const int ring4 = [] { const char* v = getenv("DA_ATTN2_RING4"); return v ? atoi(v) : 0; }();
    if (ring4 && (long long)p.B * p.H * ((p.Sq + 127) / 128) <= 2LL * da_attn2::cu_count()) ns = 4;
  }
  // Queries per workgroup, measured (profiles/r04a_attention_v2.md).  D = 128 (one workgroup per CU either way): eight waves put two
  // waves on every SIMD and halve the K / V^T stream per flop -- 1.35x over four (Flux 332 -> 245 us, Wan 9.3 -> 7.8 ms) whenever
  // there are enough 256-query workgroups to cover the chip.  D = 64: three 128-query workgroups per CU (three waves per SIMD, by
  // registers) beat one 256-query workgroup (two per SIMD) as soon as the launch has more than two workgroups per CU, and tie below.
  int qb = p.q_block;
  if (qb == 0) {
    const long long blocks256 = (long long)p.B * p.H * ((p.Sq + 255) / 256);
    qb = (p.D == 128 && blocks256 >= 192) ? 256 : 128;
  }
  const bool aug = p.algo == 3 || p.algo == 5, rsm = p.algo == 4 || p.algo == 5;
#define DA_A2V(D_, NW_, NS_)                                                                                          \
  return aug ? (rsm ? da_attn2::launch<D_, NW_, NS_, true, true>(p, s) : da_attn2::launch<D_, NW_, NS_, true, false>(p, s)) \
             : (rsm ? da_attn2::launch<D_, NW_, NS_, false, true>(p, s) : da_attn2::launch<D_, NW_, NS_, false, false>(p, s))
#define DA_A2(D_, NW_)                                                                                               \
  do {                                                                                                               \
    if (ns == 3) { DA_A2V(D_, NW_, 3); } else { DA_A2V(D_, NW_, 4); }                                                \
  } while (0)
  if (p.D == 64) {
    if (qb == 256) DA_A2(64, 8);
    DA_A2(64, 4);
  }
  if (qb == 256) DA_A2(128, 8);
  DA_A2(128, 4);
#undef DA_A2V
#undef DA_A2
}

// include/diffusers_amd.h: the split a launch would take (for sizing the workspace and for tests)
extern "C" long long da_attention_split_plan(const da_attention_params* pp, int* full_blocks, int* tail_blocks, int* units_per_tail_block) {
  if (full_blocks) *full_blocks = 0;
  if (tail_blocks) *tail_blocks = 0;
  if (units_per_tail_block) *units_per_tail_block = 1;
  if (!pp || pp->bias || pp->causal || (pp->D != 64 && pp->D != 128) || pp->B <= 0 || pp->H <= 0 || pp->Sq <= 0 || pp->Skv <= 0) return 0;
  da_attention_params p = *pp;
  int qb = p.q_block;
  if (qb == 0) {
    const long long blocks256 = (long long)p.B * p.H * ((p.Sq + 255) / 256);
    qb = (p.D == 128 && blocks256 >= 192) ? 256 : 128;
  }
  if (qb != 128 && qb != 256) return 0;
  const int nw = qb / 32;
  const int unit = nw * (p.D * 128 + 512);
  // size query: plan as if a large enough workspace were there
  static int dummy;
  p.split_ws = &dummy;
  p.split_ws_bytes = (long long)1 << 60;
  long long need = 0;
  const da_attn2::SplitPlan sp = da_attn2::split_plan(p, qb, unit, &need);
  if (sp.s <= 1) return 0;
  if (full_blocks) *full_blocks = sp.full;
  if (tail_blocks) *tail_blocks = sp.tail;
  if (units_per_tail_block) *units_per_tail_block = sp.s;
  return need;
}
