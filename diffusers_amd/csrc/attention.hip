// Flash-style scaled-dot-product attention forward for gfx950 (MI355X), bf16 in / bf16 out, fp32 softmax.
//
// Replaces F.scaled_dot_product_attention at diffusers models/attention_processor.py:2767 (AttnProcessor2_0, UNet
// self/cross attention) and models/attention_dispatch.py:3709 (_native_attention, Flux / Wan).  No mask, no dropout,
// not causal -- exactly what the five BASELINE configs use.
//
// Layout: q/k/out are token-major [B][S][H*D] views with arbitrary row strides (so the fused QK projection output
// can be consumed in place); V is consumed TRANSPOSED, vt[(h*D+d)][b*vt_batch_stride + s], which the projection GEMM
// produces for free by swapping its operands (out^T = W . X^T).  That keeps every MFMA operand K-contiguous:
//
//   S^T[kv][q] = K[kv][:] . Q[q][:]^T      A = K tile (LDS), B = Q (registers)         v_mfma_f32_32x32x16_bf16
//   O^T[d][q]  = V^T[d][kv] . P^T[kv][q]   A = V^T tile (LDS), B = P^T (registers, straight from the S^T accumulators)
//
// With the "swapped" product each lane owns ONE query (column lane&31): the row max / row sum are in-lane plus one
// lane^32 exchange, the online-softmax rescale is a per-lane scalar, and P never leaves registers: the MFMA K-index
// permutation that the C/D layout imposes on P is simply applied to the V^T fragment reads as well.
//
// One block = 4 waves = 128 queries of one (batch, head); KV tiles of 64 in an NS-slot LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4: no staging registers).  NS - 1 tiles are in flight while one is consumed and the wait in
// front of each tile is a COUNTED s_waitcnt vmcnt((NS - 2) * LOADS) + raw s_barrier, so the K / V^T stream never
// drains: with the 2-slot ring of the first version every tile paid one full L2 / HBM round trip (~1-2 k cycles)
// against ~0.5 k cycles of MFMA work.  One rendezvous per tile orders both hazards: every wave has passed its wait
// for tile j (tile j has landed for all readers) and has retired its LDS reads of tile j-1 (lgkmcnt(0)), whose slot the
// DMA issued right after the barrier overwrites.
#include <type_traits>

#include "common.cuh"

int da_attn2_dispatch(const da_attention_params& p, hipStream_t s);   // attention2.hip

namespace {

__device__ uint4 g_zero_chunk[1];  // 16 B of zeros: the source of every K / V^T chunk past the end of the sequence

template <int D, int NW = 4>
struct AttnCfg {
  static constexpr int CPR = D / 8;              // 16-byte chunks per K row
  static constexpr int KBYTES = 64 * D * 2;      // K tile  [64][D]
  static constexpr int VBYTES = D * 128;         // V^T tile [D][64]
  static constexpr int STAGE = KBYTES + VBYTES;
  static constexpr int KCH = (64 * CPR) / (64 * NW);   // K chunks staged per thread (NW waves share the tile)
  static constexpr int VCH = (D * 8) / (64 * NW);      // V^T chunks staged per thread
};

// 16-byte-slot XOR applied to a K row so the 16 rows of a ds_read_b128 lane group spread over the LDS bank row.  The XOR
// must keep a chunk inside its row (CPR = D/8 chunks): D = 64 / 128 have power-of-two rows (conflict free); D = 96 (SD1.5
// head_dim 80, zero padded) can only permute the low two bits; D = 160 (SD1.5 head_dim 160, sequence <= 256: negligible
// share of a step) stays linear.
template <int D>
__device__ __forceinline__ int k_swz(int row) {
  if (D == 64) return (row >> 1) & 7;
  if (D == 128) return row & 15;
  if (D == 96) return (row >> 2) & 3;
  return 0;
}

// MASKED (D = 64 instantiations only): causal mask and / or an additive score bias -- the text encoders (CLIP: causal;
// T5 / UMT5: relative position bias + key padding, scale 1).  The unmasked instantiations carry none of this code.
// NW = waves per block = 32-query groups per block (4: 128 queries, the default; 2: 64 queries, da_attention_params.q_block).
// PIPE (unmasked, NS = 3): the P.V product of tile j - 1 is issued AFTER the Q.K^T product of tile j, so its MFMAs run
// under tile j's softmax arithmetic instead of behind it (one more P fragment set and one more ring slot -- tile j - 1's V^T
// stays resident while tile j + 1 lands).  Same operations on the same values in the same order per accumulator:
// bit-identical to the unpipelined kernel.
// PIPE == 2 (unmasked, NS = 4): additionally the Q.K^T product of tile j + 1 is issued inside tile j's softmax slices, next to
// tile j - 1's P.V: per slice one (D = 128: two) MFMA of each, so NEITHER matrix product has a phase of its own and the
// iteration is as long as its VALU work (one more score-tile register set, one more ring slot: tile j - 1's V^T, tile j's V^T
// and tile j + 1's K resident while tile j + 2 lands).  Same operations on the same values in the same order per accumulator.
template <int D, int NS, bool MASKED = false, int NW = 4, int PIPE = 0>
__global__ __launch_bounds__(64 * NW, (D <= 64 ? 2 : 1)) void attn_fwd_kernel(const da_attention_params p) {
  static_assert(PIPE != 1 || (NS == 3 && !MASKED), "the pipelined loop is the unmasked 3-slot variant");
  static_assert(PIPE != 2 || (NS == 4 && !MASKED && NW == 4), "the doubly pipelined loop is the unmasked 4-slot variant");
  using C = AttnCfg<D, NW>;
  constexpr int QT = 32 * NW;                // queries per block
  constexpr int PD = NS - 1;                 // prefetch distance (tiles in flight ahead of the one being consumed)
  constexpr int LOADS = C::KCH + C::VCH;     // LDS-DMA instructions per wave per tile
  static_assert(NS >= 2 && NS <= 4 && (PD - 1) * LOADS <= 63, "vmcnt is a 6-bit counter");
  static_assert(NS * C::STAGE <= 160 * 1024, "K / V^T ring exceeds the LDS of a CU");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  // XCD-aware mapping (speed only): block id -> (XCD = id % 8, slot = id / 8).  All query tiles of one (batch, head)
  // pair go to ONE XCD, so its K / V^T (1 MB at S = 4096, D = 64; a 17 MB stream for Wan) is pulled into a single L2
  // and shared by the blocks that walk it side by side, instead of being fetched by all eight XCDs.
  const int qtiles = (p.Sq + QT - 1) / QT;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pair = (slot / qtiles) * 8 + xcd;            // (batch, head) pairs are dealt round-robin to the XCDs
  if (pair >= p.B * p.H) return;                          // ragged last round: leaves before any barrier
  const int b = pair / p.H, h = pair - b * p.H;
  const int q0 = (slot % qtiles) * QT + wave * 32;

  const uint16_t* __restrict__ Q = (const uint16_t*)p.q + (size_t)b * p.q_batch_stride + (size_t)h * D;
  const uint16_t* __restrict__ K = (const uint16_t*)p.k + (size_t)b * p.k_batch_stride + (size_t)h * D;
  const uint16_t* __restrict__ VT = (const uint16_t*)p.vt + (size_t)h * D * p.vt_ld + (size_t)b * p.vt_batch_stride;
  uint16_t* __restrict__ O = (uint16_t*)p.out + (size_t)b * p.o_batch_stride + (size_t)h * D;

  // ---- Q fragments (MFMA B operand): lane (q = l31, hi) holds Q[q][16*ks + 8*hi + 0..7] ----
  bf16x8_t qf[D / 16];
  {
    const int q = q0 + l31;
    const bool ok = q < p.Sq;
    const uint16_t* qp = Q + (size_t)(ok ? q : 0) * p.q_row_stride + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      uint4 v = *(const uint4*)(qp + 16 * ks);
      if (!ok) v = make_uint4(0, 0, 0, 0);
      qf[ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  }

  // ---- staging: LDS-DMA, one 1 KiB piece (64 lanes x 16 B) per wave instruction ----
  // The LDS image is lane-linear, so the bank swizzles live on the SOURCE side: linear chunk p of the K tile is
  // (row = p / CPR, slot = p % CPR) and receives global chunk slot ^ k_swz(row); chunk p of the V^T tile is
  // (d = p / 8, slot = p % 8) and receives chunk slot ^ ((d >> 1) & 7).  Chunks past the end of the sequence are
  // redirected to a line of zeros BEFORE the load (a pointer select: nothing ever waits on a loaded value here).
  const uint16_t* zsrc = (const uint16_t*)g_zero_chunk;
  auto issue = [&](int tile, int buf) {
    const int kv0 = tile * 64;
    unsigned char* kb = smem + buf * C::STAGE;
    unsigned char* vb = kb + C::KBYTES;
#pragma unroll
    for (int i = 0; i < C::KCH; ++i) {
      const int pch = (i * NW + wave) * 64 + lane;
      const int row = pch / C::CPR, slot = pch % C::CPR;
      const int c = slot ^ k_swz<D>(row);
      const int kv = kv0 + row;
      const uint16_t* src = (kv < p.Skv_alloc) ? K + (size_t)kv * p.k_row_stride + c * 8 : zsrc;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(kb + (i * NW + wave) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < C::VCH; ++i) {
      const int pch = (i * NW + wave) * 64 + lane;
      const int d = pch >> 3, slot = pch & 7;
      const int c = slot ^ ((d >> 1) & 7);
      const int kv = kv0 + c * 8;
      const uint16_t* src = (kv < p.Skv_alloc) ? VT + (size_t)d * p.vt_ld + kv : zsrc;  // Skv_alloc % 8 == 0
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(vb + (i * NW + wave) * 1024), 16, 0, 0);
    }
  };
  // wait until at most G of this wave's later tiles are still in flight (so tile j has landed) and this wave's LDS
  // reads have retired, then rendezvous (raw barrier: __syncthreads would drain vmcnt to 0) + compiler fence
#define DA_ATTN_VMCNT_CASE(G_) \
  case G_: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((G_) * LOADS <= 63 ? (G_) * LOADS : 63) : "memory"); break;
  auto wait_tile = [&](int g) {
    switch (g) {
      DA_ATTN_VMCNT_CASE(1) DA_ATTN_VMCNT_CASE(2)
      default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
#undef DA_ATTN_VMCNT_CASE

  f32x16_t o[D / 32];
#pragma unroll
  for (int i = 0; i < D / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float sl2 = p.scale * 1.4426950408889634f;

  const int ntiles = (p.Skv + 63) >> 6;
  const int ksw = k_swz<D>(l31);           // K fragment swizzle of this lane's row
  const int vsw = (l31 >> 1) & 7;          // V^T 16-byte chunk swizzle of this lane's row ((32*dt + l31) >> 1) & 7

  if constexpr (PIPE == 2) {
    bf16x8_t pf[4];            // P^T fragments of the previous tile, consumed one iteration late
    auto vfrag = [&](const unsigned char* vb, int k) {           // V^T fragment of P.V MFMA k = u * DT + dt
      constexpr int DT = D / 32;
      const int u = k / DT, dt = k % DT;
      const unsigned char* vrow = vb + (32 * dt + l31) * 128;
      const uint2 a0 = *(const uint2*)(vrow + (((2 * u) ^ vsw) << 4) + 8 * hi);
      const uint2 a1 = *(const uint2*)(vrow + (((2 * u + 1) ^ vsw) << 4) + 8 * hi);
      return __builtin_bit_cast(bf16x8_t, make_uint4(a0.x, a0.y, a1.x, a1.y));
    };
    auto kfrag = [&](const unsigned char* kb, int k) {           // K fragment of Q.K^T MFMA k = 2 * ks + st
      const int ks = k >> 1, st = k & 1;
      return *(const bf16x8_t*)(kb + (32 * st + l31) * (2 * D) + (((2 * ks + hi) ^ ksw) << 4));
    };
    constexpr int DT = D / 32, NPV = 4 * DT, NQK = 2 * (D / 16), PERV = NPV / 8, PERK = NQK / 8;
    static_assert(NPV % 8 == 0 && NQK % 8 == 0, "eight softmax slices per tile");
    issue(0, 0);
    if (ntiles > 1) {
      issue(1, 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    f32x16_t s[2], sn[2];    // scores of the tile in its softmax / partial scores of the next tile
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[st][r] = 0.f;
#pragma unroll
    for (int k = 0; k < NQK; ++k) s[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(smem, k), qf[k >> 1], s[k & 1], 0, 0, 0);
    // One iteration = softmax of tile j over the score registers `s` (complete since the previous iteration), with tile j - 1's
    // P.V (HAS_PREV) and tile j + 1's Q.K^T into `sn` (HAS_NEXT) riding in its eight slices; RAGGED: tile j may reach past Skv.
    // The LAST Q.K^T MFMA of each score tile writes its result into `s` (D != C): tile st's registers are dead once slice 6 + st
    // has packed them, so the final product of tile 0 goes out at the head of slice 7 and that of tile 1 right behind slice 7 --
    // no copy and ONE loop body.
    auto iter = [&](int j, auto has_prev_c, auto has_next_c, auto ragged_c) __attribute__((always_inline)) {
      constexpr bool HAS_PREV = decltype(has_prev_c)::value, HAS_NEXT = decltype(has_next_c)::value, RAGGED = decltype(ragged_c)::value;
      // in flight: tile j + 1 only (sent one iteration ago); past the rendezvous every wave has finished iteration j - 1, i.e.
      // its reads of tile j - 2's V^T -- the slot tile j + 2 now goes to
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (j + 2 < ntiles) issue(j + 2, (j + 2) & 3);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* kbn = smem + ((j + 1) & 3) * C::STAGE;
      const unsigned char* vbp = smem + ((j + 3) & 3) * C::STAGE + C::KBYTES;
      const int kv0 = j * 64;
      if constexpr (RAGGED) {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kv = kv0 + 32 * st + (r & 3) + 8 * (r >> 2) + 4 * hi;
            s[st][r] = (kv >= p.Skv) ? -1e30f : s[st][r];
          }
      }
      float mx = -1e30f, m_new = 0.f, alpha = 1.f, psum = 0.f;
      // (P^T of tile j is packed straight into `pf`: fragment u of tile j - 1 was consumed by the P.V MFMAs of slices 2u, 2u + 1,
      // all issued before slice 6 / 7 overwrite it -- sixteen registers the 256-register budget of two waves per SIMD needs)
      auto slice = [&](int sl) {
        if (sl == 1) {
#pragma unroll
          for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[st][r]);
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          m_new = fmaxf(m_run, mx * sl2);
          alpha = __builtin_amdgcn_exp2f(m_run - m_new);
          m_run = m_new;
        } else if (sl >= 2 && sl <= 5) {
          const int st = (sl - 2) >> 1, r0 = 8 * ((sl - 2) & 1);
#pragma unroll
          for (int r = r0; r < r0 + 8; ++r) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[st][r], sl2, -m_new));
            s[st][r] = e;
            psum += e;
          }
        } else if (sl >= 6) {
          if (sl == 6) l_run = l_run * alpha + psum;
#pragma unroll
          for (int u = 2 * (sl - 6); u < 2 * (sl - 6) + 2; ++u) {
            const int b8 = 8 * (u & 1);
            const uint4 pk = make_uint4(pack_bf2(s[u >> 1][b8 + 0], s[u >> 1][b8 + 1]), pack_bf2(s[u >> 1][b8 + 2], s[u >> 1][b8 + 3]),
                                        pack_bf2(s[u >> 1][b8 + 4], s[u >> 1][b8 + 5]), pack_bf2(s[u >> 1][b8 + 6], s[u >> 1][b8 + 7]));
            pf[u] = __builtin_bit_cast(bf16x8_t, pk);
          }
        }
      };
      // fragments of slice sl + 1 are read during slice sl (statically indexed: no copies, short live ranges)
      bf16x8_t avs[8][PERV], kfs[8][PERK];
      if constexpr (HAS_PREV) {
#pragma unroll
        for (int q = 0; q < PERV; ++q) avs[0][q] = vfrag(vbp, q);
      }
      if constexpr (HAS_NEXT) {
#pragma unroll
        for (int q = 0; q < PERK; ++q) kfs[0][q] = kfrag(kbn, q);
      }
      const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) {
        if (sl < 7) {
          if constexpr (HAS_PREV) {
#pragma unroll
            for (int q = 0; q < PERV; ++q) avs[sl + 1][q] = vfrag(vbp, (sl + 1) * PERV + q);
          }
          if constexpr (HAS_NEXT) {
#pragma unroll
            for (int q = 0; q < PERK; ++q) kfs[sl + 1][q] = kfrag(kbn, (sl + 1) * PERK + q);
          }
        }
        if constexpr (HAS_PREV) {
#pragma unroll
          for (int q = 0; q < PERV; ++q) {
            const int k = sl * PERV + q;
            o[k % DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avs[sl][q], pf[k / DT], o[k % DT], 0, 0, 0);
          }
        }
        if constexpr (HAS_NEXT) {
#pragma unroll
          for (int q = 0; q < PERK; ++q) {
            const int k = sl * PERK + q;                          // k < 2: the first product of score tile k (C operand = 0)
            if (k < NQK - 2) sn[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfs[sl][q], qf[k >> 1], k < 2 ? zero16 : sn[k & 1], 0, 0, 0);
          }
          if (sl == 7) s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfs[(NQK - 2) / PERK][(NQK - 2) % PERK], qf[(NQK - 2) >> 1], sn[0], 0, 0, 0);
        }
        slice(sl);
        if constexpr (HAS_NEXT) {
          if (sl == 7) s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfs[(NQK - 1) / PERK][(NQK - 1) % PERK], qf[(NQK - 1) >> 1], sn[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(pf[u]));
      __builtin_amdgcn_sched_barrier(0);
      // rescale AFTER tile j - 1's product has been added: O_j-1 complete, then * alpha_j, then (next iteration) + P_j V_j
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
        for (int i = 0; i < D / 32; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (ntiles == 1) {
      iter(0, F_{}, F_{}, T_{});
    } else {
      iter(0, F_{}, T_{}, F_{});
      for (int j = 1; j < ntiles - 1; ++j) iter(j, T_{}, T_{}, F_{});
      iter(ntiles - 1, T_{}, F_{}, T_{});
    }
    {                                                             // the last tile's product
      const unsigned char* vb = smem + ((ntiles - 1) & 3) * C::STAGE + C::KBYTES;
#pragma unroll
      for (int k = 0; k < NPV; ++k) o[k % DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(vb, k), pf[k / DT], o[k % DT], 0, 0, 0);
    }
  } else if constexpr (PIPE == 1) {
    bf16x8_t pf[4];            // P^T fragments of the previous tile, consumed one iteration late
    auto pv = [&](const unsigned char* vb) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
          const unsigned char* vrow = vb + (32 * dt + l31) * 128;
          const uint2 a0 = *(const uint2*)(vrow + (((2 * u) ^ vsw) << 4) + 8 * hi);
          const uint2 a1 = *(const uint2*)(vrow + (((2 * u + 1) ^ vsw) << 4) + 8 * hi);
          const uint4 av = make_uint4(a0.x, a0.y, a1.x, a1.y);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), pf[u], o[dt], 0, 0, 0);
        }
      }
    };
    issue(0, 0);
    int cur = 0, prv = 2, nxt = 1;   // ring slots of tile j, tile j - 1, tile j + 1
    // One iteration.  HAS_PREV: tile j - 1's product is pending; RAGGED: tile j may reach past Skv.  Both are compile-time
    // so that [Q.K^T(j), P.V(j-1), softmax(j), pack P(j)] is ONE branch-free scheduling region: the directives at its end
    // put each P.V MFMA in front of its share of the softmax VALU work (the in-order issue would otherwise run all 16
    // MFMAs first and the softmax behind them).
    auto iter = [&](int j, auto has_prev_c, auto ragged_c) {
      constexpr bool HAS_PREV = decltype(has_prev_c)::value, RAGGED = decltype(ragged_c)::value;
      // only tile j is outstanding here; after the rendezvous every wave has finished iteration j - 1, i.e. its reads
      // of tile j - 2's V^T -- the slot tile j + 1 now goes to
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (j + 1 < ntiles) issue(j + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* kb = smem + cur * C::STAGE;

      f32x16_t s[2];
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[st][r] = 0.f;
      {
        const unsigned char* krow0 = kb + l31 * (2 * D);
        const unsigned char* krow1 = kb + (32 + l31) * (2 * D);
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) {
          const int off = ((2 * ks + hi) ^ ksw) << 4;
          const bf16x8_t kf0 = *(const bf16x8_t*)(krow0 + off);
          const bf16x8_t kf1 = *(const bf16x8_t*)(krow1 + off);
          s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0, qf[ks], s[0], 0, 0, 0);
          s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1, qf[ks], s[1], 0, 0, 0);
        }
      }
      // ragged last tile: select, not branch
      const int kv0 = j * 64;
      if constexpr (RAGGED) {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kv = kv0 + 32 * st + (r & 3) + 8 * (r >> 2) + 4 * hi;
            s[st][r] = (kv >= p.Skv) ? -1e30f : s[st][r];
          }
      }
      // The softmax of tile j in 8 slices, each fenced behind ONE (D = 128: two) MFMA of O^T += V^T(j-1) . P^T(j-1): the
      // wave issues in order, so a P.V MFMA only runs under softmax VALU work that FOLLOWS it in the instruction stream.
      // (sched_group_barrier could not express this: the scheduler kept all 16 MFMAs in front of the first v_max.)
      float mx = -1e30f, m_new = 0.f, alpha = 1.f, psum = 0.f;
      bf16x8_t pn[4];
      auto slice = [&](int sl) {
        if (sl == 1) {
#pragma unroll
          for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[st][r]);
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          m_new = fmaxf(m_run, mx * sl2);
          alpha = __builtin_amdgcn_exp2f(m_run - m_new);
          m_run = m_new;
        } else if (sl >= 2 && sl <= 5) {
          const int st = (sl - 2) >> 1, r0 = 8 * ((sl - 2) & 1);
#pragma unroll
          for (int r = r0; r < r0 + 8; ++r) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[st][r], sl2, -m_new));
            s[st][r] = e;
            psum += e;
          }
        } else if (sl >= 6) {
          if (sl == 6) l_run = l_run * alpha + psum;
#pragma unroll
          for (int u = 2 * (sl - 6); u < 2 * (sl - 6) + 2; ++u) {
            const int b8 = 8 * (u & 1);
            const uint4 pk = make_uint4(pack_bf2(s[u >> 1][b8 + 0], s[u >> 1][b8 + 1]), pack_bf2(s[u >> 1][b8 + 2], s[u >> 1][b8 + 3]),
                                        pack_bf2(s[u >> 1][b8 + 4], s[u >> 1][b8 + 5]), pack_bf2(s[u >> 1][b8 + 6], s[u >> 1][b8 + 7]));
            pn[u] = __builtin_bit_cast(bf16x8_t, pk);
          }
        }
      };
      if constexpr (HAS_PREV) {
        constexpr int DT = D / 32, NPV = 4 * DT, PER = NPV / 8;   // P.V MFMAs per slice
        const unsigned char* vb = smem + prv * C::STAGE + C::KBYTES;
        auto vload = [&](int k) {
          const int u = k / DT, dt = k % DT;
          const unsigned char* vrow = vb + (32 * dt + l31) * 128;
          const uint2 a0 = *(const uint2*)(vrow + (((2 * u) ^ vsw) << 4) + 8 * hi);
          const uint2 a1 = *(const uint2*)(vrow + (((2 * u + 1) ^ vsw) << 4) + 8 * hi);
          return __builtin_bit_cast(bf16x8_t, make_uint4(a0.x, a0.y, a1.x, a1.y));
        };
        bf16x8_t av[PER], an[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) av[q] = vload(q);
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
#pragma unroll
          for (int q = 0; q < PER; ++q)
            if (sl < 7) an[q] = vload((sl + 1) * PER + q);          // next slice's V^T fragments fly under this slice
#pragma unroll
          for (int q = 0; q < PER; ++q) {
            const int k = sl * PER + q;
            o[k % DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[q], pf[k / DT], o[k % DT], 0, 0, 0);
          }
          slice(sl);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < PER; ++q) av[q] = an[q];
        }
      } else {
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) slice(sl);
      }
      // the packed P must exist HERE: otherwise the compiler sinks the exponentials below the rescale branch and the
      // P.V MFMAs above are left with nothing to run over
#pragma unroll
      for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(pn[u]));
      __builtin_amdgcn_sched_barrier(0);
      // rescale AFTER tile j - 1's product has been added: O_j-1 complete, then * alpha_j, then (next iteration) + P_j V_j
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
        for (int i = 0; i < D / 32; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) pf[u] = pn[u];
      const int t3 = prv;
      prv = cur;
      cur = nxt;
      nxt = t3;
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (ntiles == 1) {
      iter(0, F_{}, T_{});
    } else {
      iter(0, F_{}, F_{});
      for (int j = 1; j < ntiles - 1; ++j) iter(j, T_{}, F_{});
      iter(ntiles - 1, T_{}, T_{});
    }
    pv(smem + prv * C::STAGE + C::KBYTES);   // the last tile's product (its slot was not refilled: no tile ntiles + 1)
  } else {
#pragma unroll
  for (int t0 = 0; t0 < PD; ++t0)
    if (t0 < ntiles) issue(t0, t0);

  int cur = 0, nxt = PD % NS;   // ring slot of tile j / of the tile issued in iteration j
  for (int j = 0; j < ntiles; ++j) {
    // issued so far: tiles 0 .. min(ntiles, j + PD) - 1; tile j must have landed -> the later ones may stay in flight
    wait_tile(min(ntiles, j + PD) - j - 1);
    if (j + PD < ntiles) issue(j + PD, nxt);   // slot of tile j - 1: every wave's reads of it retired before the barrier
    __builtin_amdgcn_sched_barrier(0);

    const unsigned char* kb = smem + cur * C::STAGE;
    const unsigned char* vb = kb + C::KBYTES;

    // ---- S^T = K . Q^T : two 32-kv tiles, interleaved so consecutive MFMAs hit independent accumulators ----
    f32x16_t s[2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[st][r] = 0.f;
    {
      const unsigned char* krow0 = kb + l31 * (2 * D);
      const unsigned char* krow1 = kb + (32 + l31) * (2 * D);
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        const int off = ((2 * ks + hi) ^ ksw) << 4;
        const bf16x8_t kf0 = *(const bf16x8_t*)(krow0 + off);
        const bf16x8_t kf1 = *(const bf16x8_t*)(krow1 + off);
        s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0, qf[ks], s[0], 0, 0, 0);
        s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1, qf[ks], s[1], 0, 0, 0);
      }
    }

    // ---- online softmax; lane owns query l31, kv index of s[st][r] = 32*st + (r&3) + 8*(r>>2) + 4*hi ----
    // The softmax scale is folded into the exponent: p = exp2(s * sl2 - m) is one fma + one exp per score, and the
    // running maximum is tracked on the scaled values (sl2 > 0).
    const int kv0 = j * 64;
    float mx = -1e30f;
    if (kv0 + 64 > p.Skv) {  // ragged last tile only (wave-uniform)
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + 32 * st + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (kv >= p.Skv) s[st][r] = -1e30f;
        }
    }
    if constexpr (MASKED) {
      const int qi = q0 + l31;                                   // this lane's query
      if (p.bias) {
        // s_eff = s + bias / scale, so that s_eff * scale = scale * q.k + bias; 4 consecutive keys per load
        const float inv_scale = 1.0f / p.scale;
        const size_t boff = (size_t)b * p.bias_batch_stride + (size_t)h * p.bias_head_stride +
                            (size_t)min(qi, p.Sq - 1) * p.bias_row_stride;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int kvb = kv0 + 32 * st + 8 * rg + 4 * hi;
            float bv[4];
            if (p.bias_f32) {
              const float4 t4 = *(const float4*)((const float*)p.bias + boff + kvb);
              bv[0] = t4.x; bv[1] = t4.y; bv[2] = t4.z; bv[3] = t4.w;
            } else {
              const uint2 t2 = *(const uint2*)((const uint16_t*)p.bias + boff + kvb);
              bv[0] = bf_lo(t2.x); bv[1] = bf_hi(t2.x); bv[2] = bf_lo(t2.y); bv[3] = bf_hi(t2.y);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
              s[st][4 * rg + e] = (bv[e] <= -1e29f || s[st][4 * rg + e] <= -1e29f)   // masked by the bias / past Skv
                                      ? -1e30f : __builtin_fmaf(bv[e], inv_scale, s[st][4 * rg + e]);
          }
      }
      if (p.causal) {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kv0 + 32 * st + (r & 3) + 8 * (r >> 2) + 4 * hi > qi) s[st][r] = -1e30f;
      }
    }
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[st][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * sl2);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[st][r], sl2, -m_new));
        if constexpr (MASKED) e = (s[st][r] <= -1e29f) ? 0.f : e;   // a fully masked tile must contribute nothing
        s[st][r] = e;
        psum += e;
      }
    l_run = l_run * alpha + psum;  // per-half partial; halves are combined after the loop
    // the O rescale is D/2 multiplies per lane per tile; once the running maxima have settled (after the first tiles)
    // alpha == 1 in every lane and the pass is skipped wave-uniformly (x * 1.0f is exact: same bits either way)
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
      for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }

    // ---- P^T fragments: MFMA u (K=16) takes registers 8*(u&1)..+7 of tile u>>1; packed two per v_cvt_pk_bf16_f32 ----
    bf16x8_t pf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = 8 * (u & 1);
      const uint4 pk = make_uint4(pack_bf2(s[u >> 1][b + 0], s[u >> 1][b + 1]), pack_bf2(s[u >> 1][b + 2], s[u >> 1][b + 3]),
                                  pack_bf2(s[u >> 1][b + 4], s[u >> 1][b + 5]), pack_bf2(s[u >> 1][b + 6], s[u >> 1][b + 7]));
      pf[u] = __builtin_bit_cast(bf16x8_t, pk);
    }

    // ---- O^T += V^T . P^T ; V^T fragment of lane (d = 32*dt + l31, hi): kv 16u+4hi+{0..3} and 16u+8+4hi+{0..3} ----
    // u outer / dt inner: the D/32 accumulators of one u are independent, so the MFMAs issue back to back
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt) {
        const unsigned char* vrow = vb + (32 * dt + l31) * 128;
        const uint2 a0 = *(const uint2*)(vrow + (((2 * u) ^ vsw) << 4) + 8 * hi);
        const uint2 a1 = *(const uint2*)(vrow + (((2 * u + 1) ^ vsw) << 4) + 8 * hi);
        const uint4 av = make_uint4(a0.x, a0.y, a1.x, a1.y);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), pf[u], o[dt], 0, 0, 0);
      }
    }

    cur = (cur + 1 == NS) ? 0 : cur + 1;
    nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
  }
  }  // !PIPE

  // ---- epilogue ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < p.Sq) {
    uint16_t* op = O + (size_t)q * p.o_row_stride;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack_bf2(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
        pk.y = pack_bf2(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        *(uint2*)(op + 32 * dt + 8 * g + 4 * hi) = pk;
      }
  }
}

template <int D, int NS, bool MASKED = false, int NW = 4, int PIPE = 0>
int launch_attn(const da_attention_params& p, hipStream_t s) {
  using C = AttnCfg<D, NW>;
  const size_t lds = (size_t)NS * C::STAGE;
  auto kern = attn_fwd_kernel<D, NS, MASKED, NW, PIPE>;
  if (lds > 48 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DA_ERR_LAUNCH;
      attr_set = true;
    }
  }
  const int qtiles = (p.Sq + 32 * NW - 1) / (32 * NW), rounds = (p.B * p.H + 7) / 8;
  dim3 grid(8 * rounds * qtiles);
  DA_LAUNCH(kern, grid, dim3(64 * NW), lds, s, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// ring depth: p.ring_slots (2..4) when the caller pins it (A/B measurements, tests), else 2 for D <= 64 and 3 above.
// Measured (profiles/r02b_kernel_experiments.md): the tile loop is bound by its VALU work (exp2 / max / packing), not by
// the K / V^T stream -- a deeper ring changes nothing at D = 128 (S = 4608: 586 / 563 / 566 us for 2 / 3 / 4 slots) and
// costs occupancy at D = 64 (S = 4096: 147 / 149 / 167 us)
// Default of da_attention_params.pv_delay, measured (profiles/r02g_attention_pv_delay.md): + 3-6 % at D = 64 on long key
// sequences, - 6 % at D = 128 (Flux), nothing on the 77-key cross attention -- the tile loop is bound by its VALU instruction
// count (exp2 at quarter rate, max / sum, packing), which both loops share, not by MFMA time waiting behind VALU time.
template <int D>
int launch_attn_ring(const da_attention_params& p, hipStream_t s) {
  using C = AttnCfg<D>;
  constexpr int def_ns = (D <= 64 || 3 * C::STAGE > 160 * 1024) ? 2 : 3;
  if (p.bias || p.causal) {
    if constexpr (D == 64) return launch_attn<64, 2, true>(p, s);
    else return DA_ERR_UNSUPPORTED;
  }
  int ns = p.ring_slots ? p.ring_slots : def_ns;
  if (p.Skv <= 64) ns = 2;   // a single tile: nothing to pipeline
  if constexpr (D == 64) {
    // 64-query workgroups only on request: measured (profiles/r02e_kernel_experiments.md) they do not help even where 128-query
    // workgroups leave CUs with one block (SDXL S = 1024: 34.9 vs 36.1 us) and cost 1.3-1.5x at S = 4096
    const bool q64 = p.q_block == 64;
    if (q64 && ns == 2) return launch_attn<64, 2, false, 2>(p, s);
  }
  // PV-delayed loop: on request, for the head sizes it is built for; by default where it measured faster (see above)
  if constexpr (D == 64 || D == 128) {
    const bool delay = p.pv_delay ? p.pv_delay == 1 : (D == 64 && p.Skv >= 512);
    if (p.pv_delay == 2 && p.Skv > 64 && (p.ring_slots == 0 || p.ring_slots == 4) && p.q_block != 64) {
      if constexpr (4 * C::STAGE <= 160 * 1024) return launch_attn<D, 4, false, 4, 2>(p, s);
    }
    if (delay && p.Skv > 64 && (p.ring_slots == 0 || p.ring_slots == 3) && p.q_block != 64)
      return launch_attn<D, 3, false, 4, 1>(p, s);
  }
  switch (ns) {
    case 2: return launch_attn<D, 2>(p, s);
    case 3: if constexpr (3 * C::STAGE <= 160 * 1024) return launch_attn<D, 3>(p, s); else return DA_ERR_UNSUPPORTED;
    case 4: if constexpr (4 * C::STAGE <= 160 * 1024) return launch_attn<D, 4>(p, s); else return DA_ERR_UNSUPPORTED;
  }
  return DA_ERR_INVALID;
}

}  // namespace

extern "C" int da_attention_bf16(const da_attention_params* pp, void* stream) {
  if (!pp) return DA_ERR_INVALID;
  const da_attention_params& p = *pp;
  if (!p.q || !p.k || !p.vt || !p.out) return DA_ERR_INVALID;
  if (p.B <= 0 || p.H <= 0 || p.Sq <= 0 || p.Skv <= 0) return DA_ERR_INVALID;
  if (p.Skv_alloc < p.Skv || (p.Skv_alloc & 7)) return DA_ERR_INVALID;
  if (p.ring_slots != 0 && (p.ring_slots < 2 || p.ring_slots > 4)) return DA_ERR_INVALID;
  if (p.q_block != 0 && p.q_block != 64 && p.q_block != 128 && p.q_block != 256) return DA_ERR_INVALID;
  if (p.algo < 0 || p.algo > 5) return DA_ERR_INVALID;
  if (p.pv_delay < -1 || p.pv_delay > 2) return DA_ERR_INVALID;
  if (p.bias && ((p.bias_row_stride != 0 && p.bias_row_stride < ((p.Skv + 63) & ~63)) || (p.bias_row_stride & 3) || (p.bias_batch_stride & 3) ||
                 (p.bias_head_stride & 3) || p.scale == 0.0f))
    return DA_ERR_INVALID;
  if ((p.q_row_stride & 7) || (p.k_row_stride & 7) || (p.vt_ld & 7) || (p.vt_batch_stride & 7) || (p.o_row_stride & 3))
    return DA_ERR_UNSUPPORTED;
  if ((p.q_batch_stride & 7) || (p.k_batch_stride & 7) || (p.o_batch_stride & 3)) return DA_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  // second generation (attention2.hip) unless the caller pinned the first one or one of its variants
  const bool pinned_v1 = p.algo == 1 || (p.algo == 0 && (p.ring_slots == 2 || p.q_block == 64 || p.pv_delay != 0));
  if (!pinned_v1 && (p.o_row_stride & 3) == 0) {
    const int rc = da_attn2_dispatch(p, s);
    if (rc != DA_ERR_UNSUPPORTED || p.algo >= 2) return rc;
  }
  if (p.q_block == 256) return DA_ERR_UNSUPPORTED;
  switch (p.D) {
    case 64: return launch_attn_ring<64>(p, s);
    case 96: return launch_attn_ring<96>(p, s);
    case 128: return launch_attn_ring<128>(p, s);
    case 160: return launch_attn_ring<160>(p, s);
  }
  return DA_ERR_UNSUPPORTED;
}
