// Shared kernel template of the bf16 MFMA GEMM / implicit-GEMM conv (instantiated by gemm.hip for nn.Linear and by
// gemm_conv.hip for nn.Conv2d so the two halves compile in parallel).  See gemm.hip for the design notes.
#pragma once
#include <type_traits>

#include "common.cuh"
#include "diffusers_amd.h"

namespace da_gemm {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant expression in the body
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

static __device__ uint4 g_zero_line[8];  // 128 B of zeros: source for out-of-bounds rows in the direct-to-LDS path

// Buffer descriptors must be PROVABLY wave-uniform or the compiler wraps every buffer op in a readfirstlane waterfall
// loop: pass the base pointer (as two halves) and the byte count through readfirstlane once.
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins exist in the device pass only
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, size_t bytes) {
  const uint64_t v = (uint64_t)base;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  const int n = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffffull ? 0x7fffffffull : bytes));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}
#endif

struct RowInfo {      // per staged activation row (implicit GEMM gather state)
  int base;           // linear: row index (or -1 if out of range); conv: b*Hin
  int oy, ox;         // conv: oy*stride - pad, ox*stride - pad
};

// WM x WN waves, each wave owns MT x NT MFMA tiles of 32x32  ->  block tile (32*MT*WM) x (32*NT*WN), K slices of 64.
// STAGES = LDS ring depth, prefetch distance PD = STAGES - 1: PD K slices are in flight while one is multiplied.  The
// operands of these GEMMs stream from HBM / the Infinity Cache (weights are read once per denoising step, activations
// were written by the previous kernel), i.e. at ~2 k cycles of latency, so the sustained fill rate of a CU is
// (bytes in flight) / latency: the ring is made as deep as the 160 KiB of LDS allow and the wait before each
// rendezvous is a COUNTED s_waitcnt vmcnt((PD-1)*LOADS), never 0, so PD-1 slices stay in flight across every barrier.
//
// SM = staging mode: 0 register staged (global_load -> ds_write), 1 LDS-DMA with per-lane 64-bit source pointers
// (global_load_lds; the only mode that can redirect a lane to the zero line, which conv padding needs), 2 LDS-DMA through
// a buffer descriptor (buffer_load ... lds): descriptor base = the block's first operand row, per-lane byte offset fixed
// for the whole K loop, K-slice advance in ONE scalar offset -- no vector address arithmetic inside the loop.  Mode 2
// For nn.Linear rows past M / N are clamped to the last valid row (their products land in output rows / columns the
// epilogue never stores).  For conv the per-lane offset is recomputed once per TAP (not per slice) and a padded tap sets
// bit 31 of it: beyond num_records, so the hardware range check writes zeros to LDS (the convention of
// ck::amd_direct_load_global_to_lds); descriptor bases sit at the tile's first input row, which keeps offsets 31-bit on
// tensors of any size.
//
// Two launch-level extensions (speed only for the first, a different fp32 summation order for the second):
//  * PAIR: one launch may carry TWO independent problems (prob_a on blocks [0, grid_a), prob_b on the rest) -- e.g. the fused
//    Q|K projection (160 tiles of 128x256 at SDXL's 1280-wide level) and the swapped V^T projection (80 tiles) of one
//    self-attention layer, which separately fill 62 % and 31 % of the 256 CUs and together 94 %.  Each block runs
//    exactly the code it would run in its own launch: results are bit-identical to two launches.
//  * SPLIT-K (p.split_k > 1): the K slices of a tile are dealt to split_k blocks of the same XCD.
//    Blocks 0 .. split_k-2 publish their fp32 accumulators to a workspace slot with write-through (sc1) stores, drain
//    them and raise a flag; the LAST block (highest block id: dispatched after its producers) polls the flags, adds the
//    slots in index order with sc1 loads and runs the epilogue.  That is the agent-scope hand-off of
//    cdna_hip_programming.md Guideline 16 (R1: sc1 payload both sides, drained before a relaxed agent-scope flag, ONE
//    polling lane, bounded spin); the consumer re-arms the flag for the next launch on the stream.  The sum order is
//    fixed (own slices, then slot 0, 1, ...), so a given split_k is deterministic; different split_k differ in the
//    last fp32 bit of the sum.  The host admits split-K only when every block of the launch is co-resident.
template <int WM, int WN, int MT, int NT, int STAGES, bool CONV, int SM, bool SPLITK = false, bool KSKIP = false,
          bool LNFOLD = false>
__global__ __launch_bounds__(64 * WM * WN) void igemm_bf16_kernel(const da_gemm_params prob_a, const int xcd_gx_a,
                                                                  const da_gemm_params prob_b, const int xcd_gx_b,
                                                                  const int grid_a) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (and cannot parse the buffer builtins)
  // wave-uniform: a scalar select between the two kernarg blocks (nn.Linear only: conv launches are never paired, and the
  // select would cost the conv instantiations 24 bytes of scratch per lane)
  const bool second = !CONV && (int)blockIdx.x >= grid_a;
  const da_gemm_params& p = second ? prob_b : prob_a;
  const int xcd_gx = second ? xcd_gx_b : xcd_gx_a;
  const int bid = second ? (int)blockIdx.x - grid_a : (int)blockIdx.x;
  constexpr bool GLDS = (SM != 0), BLDS = (SM == 2);
  constexpr int NW = WM * WN, NTHR = 64 * NW, RP = NTHR / 8;  // RP = tile rows staged per pass (one 1 KiB piece per wave)
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  // LayerNorm fold (da_gemm_params.stats_out / ln_*): its own instantiations (LNFOLD) -- carried by every kernel it cost the
  // register-heaviest 4-wave tile 5-9 % (128x128, 2 slots: SDXL GEGLU projection 88 -> 80 us without it), and the fold
  // itself measured slower than the LayerNorm kernel it replaces (profiles/r02d_layernorm_fold.md).
  constexpr bool LNF = LNFOLD && !CONV && (MT * NT < 8);
  constexpr int XR = BM / RP, WR = BN / RP;  // staged rows per thread
  constexpr int XBYTES = BM * 128, WBYTES = BN * 128, STAGE = XBYTES + WBYTES;
  constexpr int PD = STAGES - 1;             // prefetch distance
  constexpr int LOADS = XR + WR;             // LDS-DMA instructions per wave per K slice
  static_assert(XR >= 1 && XR <= 4 && WR >= 1 && WR <= 4, "tile / thread-count combination not stageable");
  static_assert(GLDS || (STAGES == 2 && NW == 4), "register staging exists for the 4-wave 2-stage tiles only");
  static_assert(STAGES >= 2 && STAGES <= 8 && (PD - 1) * LOADS <= 63, "vmcnt is a 6-bit counter");
  static_assert(STAGES * STAGE <= 160 * 1024, "LDS ring exceeds the 160 KiB of a CU");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int t = threadIdx.x;
  // the wave index is wave-uniform but only readfirstlane proves it: LDS-DMA destinations (M0) and the fragment bases
  // below then stay in scalar registers
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int l31 = lane & 31, hi = lane >> 5;

  // ---- XCD-aware tile mapping (speed only: nothing depends on where a block really runs) ----
  // Block b is observed to run on XCD b % 8, each XCD has its own 4 MiB L2 and a CU's fill rate is set by the average
  // latency of its ~64 outstanding L1 misses, i.e. by the L2 hit rate.  Each XCD therefore owns one RECTANGLE of the
  // tile grid, (8 / xcd_gx) row groups x xcd_gx column groups, chosen on the host so that the operand panels one XCD
  // touches, (rows + columns) * K bytes, are smallest: a weight matrix that does not fit L2 is then pulled from HBM by
  // ONE XCD instead of all eight, and the blocks running side by side in an XCD share both operand panels.
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int gyn = 8 / xcd_gx;
  const int tm_per = (tiles_m + gyn - 1) / gyn, tn_per = (tiles_n + xcd_gx - 1) / xcd_gx;
  const int xcd = bid & 7, kblk = bid >> 3;       // grid_a is a multiple of 8: both problems see XCD = block id % 8
  const int gy = xcd / xcd_gx, gx = xcd - gy * xcd_gx;
  // split-K: the split index is the slowest part of the per-XCD block index, so the split_k blocks of a tile share the
  // tile's XCD (same L2 for the partial hand-off) and the reducer (last index) has the highest block id of the tile
  // (SPLITK is a template parameter: the hand-off code costs ~50 VGPRs -- the accumulators are touched by VALU adds and
  // buffer stores, not only by MFMAs -- which the unsplit instantiations must not pay)
  const int split = (SPLITK && p.split_k > 1) ? p.split_k : 1;
  const int rect = tm_per * tn_per;
  const int sidx = (split > 1) ? kblk / rect : 0;
  const int kb2 = kblk - sidx * rect;
  const int lm = kb2 / tn_per, ln = kb2 - lm * tn_per;
  const int tm = gy * tm_per + lm, tn = gx * tn_per + ln;
  if (tm >= tiles_m || tn >= tiles_n) return;  // ragged rectangle: the whole block leaves before any barrier
  const int m0 = tm * BM, n0 = tn * BN;

  const uint16_t* __restrict__ A = (const uint16_t*)p.A;
  const uint16_t* __restrict__ A2 = (const uint16_t*)p.A2;
  const uint16_t* __restrict__ Wt = (const uint16_t*)p.W;

  // ---- staging assignment: thread t stages LDS slot (row = (t>>3)+RP*i, pos = t&7) from source chunk sc ----
  const int srow = t >> 3;
  const int spos = t & 7;
  const int sc = spos ^ ((t >> 4) & 7);  // (row>>1)&7 == (t>>4)&7 for every i (RP is a multiple of 16)

  RowInfo xr[XR];
  const int Hv = CONV ? (p.Hin << p.up) : 0, Wv = CONV ? (p.Win << p.up) : 0;
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    const int m = m0 + srow + RP * i;
    if (CONV) {
      if (m < p.M) {
        const int hw = p.Hout * p.Wout;
        const int b = m / hw;
        const int rem = m - b * hw;
        const int oy = rem / p.Wout;
        const int ox = rem - oy * p.Wout;
        xr[i].base = b * p.Hin;
        xr[i].oy = oy * p.stride - p.pad;
        xr[i].ox = ox * p.stride - p.pad;
      } else {
        xr[i].base = 0;
        xr[i].oy = -100000;  // never in range
        xr[i].ox = -100000;
      }
    } else {
      xr[i].base = (m < p.M) ? m : -1;
      xr[i].oy = 0;
      xr[i].ox = 0;
    }
  }
  int wrow[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    const int n = n0 + srow + RP * i;
    wrow[i] = (n < p.N) ? n : -1;
  }

  // K slices of this block: all of them, or (split-K) the sidx-th of split nearly equal runs
  const int nk_all = p.K >> 6;
  const int nk_base = nk_all / split, nk_rem = nk_all - nk_base * split;
  const int k_begin = sidx * nk_base + min(sidx, nk_rem);
  const int nk = nk_base + (sidx < nk_rem ? 1 : 0);
  const int Ctot = CONV ? (p.C1 + p.C2) : 0;

  // buffer-addressed staging: descriptors over the block's operand panels, per-lane byte offsets in vo_*
  // (descriptors are built unconditionally -- the type has no default initialiser -- and are dead unless BLDS)
  size_t xbase = 0, xbytes = 0x7fffffff, x2bytes = 0x7fffffff;
  int pb = 0;  // conv: first input pixel the tile can touch (descriptor origin)
  if constexpr (BLDS) {
    if constexpr (CONV) {
      const int hw = p.Hout * p.Wout;
      const int b0 = m0 / hw;
      const int oy0 = (m0 - b0 * hw) / p.Wout;
      pb = __builtin_amdgcn_readfirstlane((b0 * p.Hin + (max(0, oy0 * p.stride - p.pad) >> p.up)) * p.Win);
      const size_t pix_left = (size_t)(p.M / hw) * p.Hin * p.Win - (size_t)pb;
      xbytes = min(pix_left * p.C1 * 2, (size_t)0x7fffffff);
      x2bytes = min(pix_left * p.C2 * 2, (size_t)0x7fffffff);
    } else {
      xbase = (size_t)m0 * p.lda;
    }
  }
  __amdgpu_buffer_rsrc_t rs_x = uniform_rsrc(A + (CONV ? (size_t)pb * p.C1 : xbase), xbytes);
  __amdgpu_buffer_rsrc_t rs_x2 = uniform_rsrc((CONV && A2) ? A2 + (size_t)pb * p.C2 : A, x2bytes);
  __amdgpu_buffer_rsrc_t rs_w = uniform_rsrc(Wt + (BLDS ? (size_t)n0 * p.ldw : 0), 0x7fffffff);
  int vo_x[XR], vo_x2[XR], vo_w[WR];
  int bk_off = k_begin * 128;  // byte offset of the cursor K slice inside a weight / activation row (scalar)
  // K-slice cursor of the NEXT slice to issue.  Slices are issued strictly in order, so the (tap, channel) position
  // of the implicit GEMM advances incrementally (no integer division in the loop): K index = tap * Ctot + c.  A
  // split-K block starts at slice k_begin: one division here (a slice never straddles taps: Ctot % 64 == 0).
  int is_kh = 0, is_kw = 0, is_c0 = 0;  // conv: kernel row / column of the tap, first channel of the slice
  if constexpr (CONV) {
    if (k_begin > 0) {
      const int tap = (k_begin * 64) / Ctot;
      is_c0 = k_begin * 64 - tap * Ctot;
      is_kh = tap / p.conv;
      is_kw = tap - is_kh * p.conv;
    }
  }
  // conv: offsets of the activation rows for tap (kh, kw) -- called once per tap, not per K slice
  auto tap_offsets = [&](int kh, int kw) {
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int iy = xr[i].oy + kh, ix = xr[i].ox + kw;
      const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
      const int rel = (xr[i].base + (iy >> p.up)) * p.Win + (ix >> p.up) - pb;
      vo_x[i] = ok ? (rel * p.C1 + sc * 8) * 2 : (int)0x80000000;
      vo_x2[i] = ok ? (rel * p.C2 + sc * 8) * 2 : (int)0x80000000;
    }
  };
  if constexpr (BLDS) {
    if constexpr (CONV) {
      tap_offsets(is_kh, is_kw);
    } else {
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        vo_x[i] = (min(srow + RP * i, p.M - 1 - m0) * p.lda + sc * 8) * 2;
        vo_x2[i] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) vo_w[i] = (min(srow + RP * i, p.N - 1 - n0) * p.ldw + sc * 8) * 2;
  }

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 xg0, xg1, xg2, xg3, wg0, wg1, wg2, wg3;  // named (not arrays) so they never land in scratch
  xg0 = xg1 = xg2 = xg3 = wg0 = wg1 = wg2 = wg3 = make_uint4(0, 0, 0, 0);

  size_t is_k = (size_t)k_begin * 64;   // linear / weights: element offset of the slice inside a row

  // source pointer of activation row i for the cursor slice (a 128-byte line of zeros when the slot must be zero,
  // so every staging load is unconditional and the compiler keeps them all in flight)
  const uint16_t* zline = (const uint16_t*)g_zero_line;
  auto x_src = [&](int i) -> const uint16_t* {
    if (CONV) {
      const int iy = xr[i].oy + is_kh, ix = xr[i].ox + is_kw;
      if ((unsigned)iy >= (unsigned)Hv || (unsigned)ix >= (unsigned)Wv) return zline;
      const int sy = iy >> p.up, sx = ix >> p.up;
      const size_t pix = (size_t)(xr[i].base + sy) * p.Win + sx;
      if (is_c0 < p.C1) return A + pix * p.C1 + is_c0 + sc * 8;
      return A2 + pix * p.C2 + (is_c0 - p.C1) + sc * 8;
    } else {
      if (xr[i].base < 0) return zline;
      return A + (size_t)xr[i].base * p.lda + is_k + sc * 8;
    }
  };
  auto w_src = [&](int i) -> const uint16_t* {
    if (wrow[i] < 0) return zline;
    return Wt + (size_t)wrow[i] * p.ldw + is_k + sc * 8;
  };

  // Staging is written as macros (not lambdas) so the staged registers stay in VGPRs.
#define DA_STAGE_ISSUE(BUF)                                                                                            \
  do {                                                                                                                 \
    if constexpr (BLDS) {                                                                                              \
      unsigned char* xb_ = smem + (BUF) * STAGE;                                                                       \
      unsigned char* wb_ = xb_ + XBYTES;                                                                               \
      if (CONV && is_c0 >= p.C1) { /* second concat source (wave-uniform choice) */                                    \
        _Pragma("unroll") for (int i = 0; i < XR; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(                       \
            rs_x2, (__attribute__((address_space(3))) void*)(xb_ + (i * NW + wave) * 1024), 16, vo_x2[i],              \
            (is_c0 - p.C1) * 2, 0, 0);                                                                                 \
      } else {                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < XR; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(                       \
            rs_x, (__attribute__((address_space(3))) void*)(xb_ + (i * NW + wave) * 1024), 16, vo_x[i],                \
            CONV ? is_c0 * 2 : bk_off, 0, 0);                                                                          \
      }                                                                                                                \
      _Pragma("unroll") for (int i = 0; i < WR; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(                         \
          rs_w, (__attribute__((address_space(3))) void*)(wb_ + (i * NW + wave) * 1024), 16, vo_w[i], bk_off, 0, 0);   \
      bk_off += 128;                                                                                                   \
      if constexpr (CONV) {                                                                                            \
        is_c0 += 64;                                                                                                   \
        if (is_c0 >= Ctot) {                                                                                           \
          is_c0 = 0;                                                                                                   \
          if (++is_kw >= p.conv) {                                                                                     \
            is_kw = 0;                                                                                                 \
            ++is_kh;                                                                                                   \
          }                                                                                                            \
          tap_offsets(is_kh, is_kw);                                                                                   \
        }                                                                                                              \
      }                                                                                                                \
    } else if (GLDS) {                                                                                                 \
      unsigned char* xb_ = smem + (BUF) * STAGE;                                                                       \
      unsigned char* wb_ = xb_ + XBYTES;                                                                               \
      _Pragma("unroll") for (int i = 0; i < XR; ++i) {                                                                 \
        const uint16_t* s_ = x_src(i);                                                                                 \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s_,                            \
                                         (__attribute__((address_space(3))) void*)(xb_ + (i * NW + wave) * 1024), 16,  \
                                         0, 0);                                                                        \
      }                                                                                                                \
      _Pragma("unroll") for (int i = 0; i < WR; ++i) {                                                                 \
        const uint16_t* s_ = w_src(i);                                                                                 \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s_,                            \
                                         (__attribute__((address_space(3))) void*)(wb_ + (i * NW + wave) * 1024), 16,  \
                                         0, 0);                                                                        \
      }                                                                                                                \
    } else {                                                                                                           \
      xg0 = *(const uint4*)x_src(0);                                                                                   \
      xg1 = *(const uint4*)x_src(XR > 1 ? 1 : 0);                                                                      \
      if constexpr (XR > 2) {                                                                                          \
        xg2 = *(const uint4*)x_src(2);                                                                                 \
        xg3 = *(const uint4*)x_src(3);                                                                                 \
      }                                                                                                                \
      wg0 = *(const uint4*)w_src(0);                                                                                   \
      wg1 = *(const uint4*)w_src(WR > 1 ? 1 : 0);                                                                      \
      if constexpr (WR > 2) {                                                                                          \
        wg2 = *(const uint4*)w_src(2);                                                                                 \
        wg3 = *(const uint4*)w_src(3);                                                                                 \
      }                                                                                                                \
    }                                                                                                                  \
    /* advance the cursor to the next K slice (the buffer mode advanced its own above) */                              \
    if constexpr (!BLDS) {                                                                                             \
      is_k += 64;                                                                                                      \
      if (CONV) {                                                                                                      \
        is_c0 += 64;                                                                                                   \
        if (is_c0 >= Ctot) {                                                                                           \
          is_c0 = 0;                                                                                                   \
          if (++is_kw >= p.conv) {                                                                                     \
            is_kw = 0;                                                                                                 \
            ++is_kh;                                                                                                   \
          }                                                                                                            \
        }                                                                                                              \
      }                                                                                                                \
    }                                                                                                                  \
  } while (0)
#define DA_STAGE_COMMIT(BUF)                                                                                           \
  do {                                                                                                                 \
    if (!GLDS) {                                                                                                       \
      unsigned char* xb_ = smem + (BUF) * STAGE;                                                                       \
      unsigned char* wb_ = xb_ + XBYTES;                                                                               \
      unsigned char* xs_ = xb_ + srow * 128 + spos * 16;                                                               \
      unsigned char* ws_ = wb_ + srow * 128 + spos * 16;                                                               \
      *(uint4*)(xs_) = xg0;                                                                                            \
      if constexpr (XR > 1) *(uint4*)(xs_ + RP * 128) = xg1;                                                           \
      if constexpr (XR > 2) {                                                                                          \
        *(uint4*)(xs_ + 2 * RP * 128) = xg2;                                                                           \
        *(uint4*)(xs_ + 3 * RP * 128) = xg3;                                                                           \
      }                                                                                                                \
      *(uint4*)(ws_) = wg0;                                                                                            \
      if constexpr (WR > 1) *(uint4*)(ws_ + RP * 128) = wg1;                                                           \
      if constexpr (WR > 2) {                                                                                          \
        *(uint4*)(ws_ + 2 * RP * 128) = wg2;                                                                           \
        *(uint4*)(ws_ + 3 * RP * 128) = wg3;                                                                           \
      }                                                                                                                \
    }                                                                                                                  \
  } while (0)
  // Wait until at most G later slices of THIS wave's LDS-DMA are still in flight, then rendezvous.  The raw s_barrier
  // (not __syncthreads, whose fence would drain vmcnt to 0) lets those slices stay in flight across the barrier; the
  // asm "memory" clobbers keep the compiler from moving LDS accesses across the rendezvous.  G is wave-uniform.
#define DA_VMCNT_CASE(G_) \
  case G_: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((G_) * LOADS <= 63 ? (G_) * LOADS : 63) : "memory"); break;
#define DA_STAGE_WAIT(G)                                                                                               \
  do {                                                                                                                 \
    if (GLDS) {                                                                                                        \
      switch (G) {                                                                                                     \
        DA_VMCNT_CASE(1) DA_VMCNT_CASE(2) DA_VMCNT_CASE(3) DA_VMCNT_CASE(4) DA_VMCNT_CASE(5) DA_VMCNT_CASE(6)          \
        default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;                                    \
      }                                                                                                                \
      __builtin_amdgcn_s_barrier();                                                                                    \
      asm volatile("" ::: "memory");                                                                                   \
    } else {                                                                                                           \
      __syncthreads();                                                                                                 \
    }                                                                                                                  \
  } while (0)

  // fragment read offsets (bytes) inside a tile: row (l31) * 128 + ((2*ks+hi) ^ ((l31>>1)&7)) * 16
  const int fsw = (l31 >> 1) & 7;
  const int frow = l31 * 128;

  // `full` (KSKIP builds only): false = the slice's upper 32 K elements are channel padding (da_gemm_params.k_valid): its
  // k-steps 2 and 3 are not run.  Without KSKIP the body is one straight-line region of 4 k-steps.
  auto compute = [&](int buf, bool full) {
    const unsigned char* xb = smem + buf * STAGE + (wm * MT * 32) * 128 + frow;
    const unsigned char* wb = smem + buf * STAGE + XBYTES + (wn * NT * 32) * 128 + frow;
    // two fragment register sets: the ds_reads of k-step ks+1 are issued before the MFMAs of k-step ks, so the LDS
    // latency of one step hides under the matrix work of the previous one instead of in front of it
    bf16x8_t wf[2][NT], xf[2][MT];
    auto frag = [&](int set, int ks) {
      const int off = ((2 * ks + hi) ^ fsw) << 4;
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[set][j] = *(const bf16x8_t*)(wb + j * 32 * 128 + off);
#pragma unroll
      for (int i = 0; i < MT; ++i) xf[set][i] = *(const bf16x8_t*)(xb + i * 32 * 128 + off);
    };
    auto mfmas = [&](int ks) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][j], xf[ks & 1][i], acc[i][j], 0, 0, 0);
    };
    // The program order below is only a wish: without the directives the scheduler folds both fragment sets back into one
    // (read, wait, multiply).  Pin it: first set, then each step's MFMAs interleaved with the NEXT step's ds_reads.
#define DA_SG_DS(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define DA_SG_MF(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
    auto pin_step = [&]() {   // one k-step's MFMAs over the next step's MT + NT fragment reads
      if constexpr (MT * NT == 4) {         // 4 reads under 4 MFMAs
        DA_SG_DS(1); DA_SG_MF(1); DA_SG_DS(1); DA_SG_MF(1); DA_SG_DS(1); DA_SG_MF(1); DA_SG_DS(1); DA_SG_MF(1);
      } else if constexpr (MT * NT == 2) {  // 3 reads, 2 MFMAs
        DA_SG_DS(1); DA_SG_MF(1); DA_SG_DS(1); DA_SG_MF(1); DA_SG_DS(1);
      } else if constexpr (MT * NT == 1) {  // 2 reads, 1 MFMA
        DA_SG_DS(1); DA_SG_MF(1); DA_SG_DS(1);
      } else {                              // 4 x 2: 6 reads under 8 MFMAs
        static_assert(MT * NT == 8, "add an interleave pattern for this wave tile");
        DA_SG_DS(1); DA_SG_MF(2); DA_SG_DS(1); DA_SG_MF(1); DA_SG_DS(1); DA_SG_MF(1);
        DA_SG_DS(1); DA_SG_MF(2); DA_SG_DS(1); DA_SG_MF(1); DA_SG_DS(1); DA_SG_MF(1);
      }
    };
    if constexpr (!KSKIP) {
      frag(0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < 3) frag((ks + 1) & 1, ks + 1);
        mfmas(ks);
      }
      DA_SG_DS(MT + NT);
      pin_step(); pin_step(); pin_step();
      DA_SG_MF(MT * NT);
    } else {
      // k-steps 0 and 1, with step 2's fragments fetched under step 1's MFMAs whether or not they will be used (a half
      // slice's upper bytes are staged like any others), then ONE wave-uniform branch around steps 2 and 3
      frag(0, 0);
      frag(1, 1);
      mfmas(0);
      frag(0, 2);
      mfmas(1);
      DA_SG_DS(MT + NT);
      pin_step(); pin_step();
      __builtin_amdgcn_sched_barrier(0);
      if (full) {
        frag(1, 3);
        mfmas(2);
        mfmas(3);
        pin_step();
        DA_SG_MF(MT * NT);
      }
    }
#undef DA_SG_DS
#undef DA_SG_MF
  };

  // LayerNorm fold, consumer side, part 1: ONE batch of unconditional 16-byte loads of this lane's rows' partials (every
  // row holds DA_LN_MAX_PARTS slots; slots past ln_parts are masked in part 2).  DA_LN_PAIR_LOADS pairs per lane half
  // cover 4 * DA_LN_PAIR_LOADS partials per row -- the host refuses more.
  float4 ln_v[MT][DA_LN_PAIR_LOADS];
  if constexpr (LNF) {
    if (p.ln_stats) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = min(m0 + (wm * MT + i) * 32 + l31, p.M - 1);
        const float* sp = p.ln_stats + (size_t)m * p.ln_stats_ld;
#pragma unroll
        for (int u = 0; u < DA_LN_PAIR_LOADS; ++u) ln_v[i][u] = *(const float4*)(sp + 4 * (hi + 2 * u));
      }
    }
  }

  // Residual prefetch: ONE batch of unconditional 8-byte loads of every residual fragment this lane will add in the
  // epilogue, issued before the staging prologue so the round trip hides under the K loop.  Left in the epilogue's
  // per-fragment loop the loads sit behind the activation branches and retire one L2 round trip at a time: + 3-5 us on
  // a launch whose K loop is 8 us (profiles/r02e_kernel_experiments.md).  Rows / columns past the edge are clamped to
  // valid addresses (their values are never stored).  In-place accumulation (residual == C, the temporal taps of a
  // causal Conv3d) is unaffected: each element is read and later written by the same lane.
  // (Not for the 4 x 2 wave tile, whose accumulators already fill the register file, nor for split-K builds.)
  constexpr bool RES_PF = (MT * NT < 8) && !SPLITK;
  uint2 res_v[RES_PF ? MT : 1][RES_PF ? NT : 1][4];
  if constexpr (RES_PF) {
    const uint16_t* __restrict__ resid_pf = (const uint16_t*)p.residual;
    if (resid_pf) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = min(m0 + (wm * MT + i) * 32 + l31, p.M - 1);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = min(n0 + (wn * NT + j) * 32 + 8 * g + 4 * hi, p.N - 4);
            res_v[i][j][g] = *(const uint2*)(resid_pf + (size_t)m * p.ldr + n);
          }
      }
    }
  }
  // Same for the bias (one 8-byte load per 4 output channels, shared by the MT row tiles) and the per-batch channel
  // vector (ResnetBlock2D's time embedding): in the epilogue loop each sat in front of its fragment's arithmetic.
  constexpr bool RV_PF = RES_PF && !KSKIP;   // the k_valid builds sit at the 256-register line of 2 waves per SIMD
  uint2 bias_v[RES_PF ? NT : 1][4], rowvec_v[RV_PF ? MT : 1][RV_PF ? NT : 1][4];
  {
    const uint16_t* __restrict__ bias_pf = (const uint16_t*)p.bias;
    if constexpr (RES_PF) if (bias_pf) {   // GEGLU: sub-tile 2jp holds the value rows, 2jp + 1 the gate rows (= value + 32): same formula
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          bias_v[j][g] = *(const uint2*)(bias_pf + min(n0 + (wn * NT + j) * 32 + 8 * g + 4 * hi, p.N - 4));
    }
    if constexpr (RV_PF) {
      const uint16_t* __restrict__ rowvec_pf = (const uint16_t*)p.rowvec;
      if (rowvec_pf) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int m = min(m0 + (wm * MT + i) * 32 + l31, p.M - 1);
          const size_t ro = (size_t)(m / p.rows_per_batch) * p.ld_rowvec;
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              rowvec_v[i][j][g] = *(const uint2*)(rowvec_pf + ro + min(n0 + (wn * NT + j) * 32 + 8 * g + 4 * hi, p.N - 4));
        }
      }
    }
  }

  // ---- main loop: LDS ring of STAGES slices, one rendezvous per K slice ----
  // prologue: slices 0 .. min(PD, nk) - 1 go to ring slots 0 .. ; then slice 0 must have landed
#pragma unroll
  for (int s = 0; s < PD; ++s) {
    if (s < nk) {
      DA_STAGE_ISSUE(s);
      if (s == 0) DA_STAGE_COMMIT(0);
    }
  }
  // LayerNorm fold, consumer side, part 2: mean / rstd of this lane's MT output rows (the loads were issued BEFORE the
  // staging prologue, so their round trip overlaps the first K slices' flight; the compiler's vmcnt for them leaves the
  // LDS-DMA issued since in flight).  Lane halves sum alternate 16-byte pairs of partials, then combine: fixed order.
  float ln_mu[MT], ln_rs[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) ln_mu[i] = 0.f, ln_rs[i] = 1.f;
  if constexpr (LNF) {
    if (p.ln_stats) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < DA_LN_PAIR_LOADS; ++u) {
          const int q = 2 * (hi + 2 * u);
          s1 += (q < p.ln_parts ? ln_v[i][u].x : 0.f) + (q + 1 < p.ln_parts ? ln_v[i][u].z : 0.f);
          s2 += (q < p.ln_parts ? ln_v[i][u].y : 0.f) + (q + 1 < p.ln_parts ? ln_v[i][u].w : 0.f);
        }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const float inv_c = 1.0f / (float)p.K;          // the normalised dimension is this GEMM's K
        const float mean = s1 * inv_c;
        ln_mu[i] = mean;
        ln_rs[i] = rsqrtf(fmaxf(s2 * inv_c - mean * mean, 0.f) + p.ln_eps);
      }
    }
  }
  {
    const int g0 = min(nk, PD) - 1;  // slices still allowed in flight once slice 0 is needed
    DA_STAGE_WAIT(g0);
  }
  int cur = 0;          // ring slot of slice kt
  int nxt = PD;         // ring slot the next issued slice goes to (STAGES == PD + 1)
  // KSKIP: position of the slice being COMPUTED inside its K period (a kernel tap's channels / the whole K of a Linear)
  const int kper = CONV ? Ctot : p.K;
  int cpos = (KSKIP && p.k_valid > 0 && k_begin > 0) ? (k_begin * 64) % kper : 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + PD < nk);
    if (more) DA_STAGE_ISSUE(nxt);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch in flight under this slice's MFMAs
    if constexpr (KSKIP) {
      int live = 64;                    // valid K elements of this slice (wave-uniform)
      if (p.k_valid > 0) {
        live = p.k_valid - cpos;
        cpos += 64;
        if (cpos >= kper) cpos = 0;
      }
      // a slice whose upper half is padding runs k-steps 0 and 1 only, one that is all padding none (the padded operand
      // values are zeros by contract, so any other split just multiplies some of them)
      if (live > 0) compute(cur, live > 32);
    } else {
      compute(cur, true);
    }
    if (kt + 1 < nk) {
      if (more) DA_STAGE_COMMIT(nxt);
      // issued so far: min(nk, kt + PD + 1) slices; slices 0 .. kt+1 must have landed before the next iteration
      const int g = min(nk - kt - 2, PD - 1);
      DA_STAGE_WAIT(g);
    }
    cur = (cur + 1 == STAGES) ? 0 : cur + 1;
    nxt = (nxt + 1 == STAGES) ? 0 : nxt + 1;
  }
#undef DA_STAGE_ISSUE
#undef DA_STAGE_COMMIT
#undef DA_STAGE_WAIT
#undef DA_VMCNT_CASE

  // ---- split-K hand-off (see the kernel header): producers publish and leave, the reducer gathers ----
  if constexpr (SPLITK) {
    if (split > 1) {
      constexpr int TILE_FLOATS = BM * BN;
      float* ws = (float*)p.workspace;
      int* flags = (int*)p.sync_flags;
      const int slot0 = (tm * tiles_n + tn) * (split - 1);
      if (sidx < split - 1) {
        const int slot = slot0 + sidx;
        __amdgpu_buffer_rsrc_t rs = uniform_rsrc(ws + (size_t)slot * TILE_FLOATS, (size_t)TILE_FLOATS * 4);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              u32x4_t v;
              v.x = __float_as_uint(acc[i][j][4 * q + 0]); v.y = __float_as_uint(acc[i][j][4 * q + 1]);
              v.z = __float_as_uint(acc[i][j][4 * q + 2]); v.w = __float_as_uint(acc[i][j][4 * q + 3]);
              // lane-linear image: 16 B per lane, 64 lanes contiguous; sc1 = write-through past the XCD's L2
              __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((((i * NT + j) * 4 + q) * NTHR) + t) * 16, 0, 16);
            }
            __builtin_amdgcn_sched_barrier(0);   // one sub-tile's accumulator reads at a time (VGPR budget, see the reducer)
          }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains before the flag
        __syncthreads();
        if (t == 0) __hip_atomic_store(flags + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      // Round 6: the reducer first waits for ALL producers (they finish within a microsecond of each other: equal K shares), then adds
      // the partial tiles SB slots at a time -- the loads of SB slots' copy of one 32 x 32 sub-tile are in flight together, the adds
      // stay in slot order (own slices, then slot 0, 1, ...: the same sums, bit for bit).  One slot and one sub-tile at a time, as
      // before, the tail of a split-6 launch on a two-sub-tile tile was ten serialised round trips to the memory-side cache (the
      // partials leave the XCD's L2 with their write-through stores): ~12 us of the 35 us the small-M deep-K convs of the SD1.5 /
      // DDPM U-Nets take.
      if (t == 0) {   // ONE lane polls ONE word at a time, relaxed, with a bounded spin (a lost producer must not hang the GPU)
        for (int sp = 0; sp < split - 1; ++sp) {
          const int slot = slot0 + sp;
          int spins = 0;
          while (__hip_atomic_load(flags + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1 << 20)) {
              __hip_atomic_store(flags + DA_SPLITK_ERR_SLOT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
          }
          __hip_atomic_store(flags + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
        }
      }
      __syncthreads();
      constexpr int SB = (MT * NT <= 2) ? 4 : 2;           // slots in flight per sub-tile (16 VGPRs each)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          for (int sp0 = 0; sp0 < split - 1; sp0 += SB) {
            // one 32x32 sub-tile of up to SB slots (4 x 16 B per lane and slot in flight): without the fences the scheduler hoists
            // every load of every slot above the first add and the kernel's VGPR budget grows by the whole partial tiles
            u32x4_t v[SB][4];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
              // (a slot past the last one re-reads the last: loaded, never added)
              const int slot = slot0 + min(sp0 + u, split - 2);
              __amdgpu_buffer_rsrc_t rs = uniform_rsrc(ws + (size_t)slot * TILE_FLOATS, (size_t)TILE_FLOATS * 4);
#pragma unroll
              for (int q = 0; q < 4; ++q)
                v[u][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((((i * NT + j) * 4 + q) * NTHR) + t) * 16, 0, 16);
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
              if (sp0 + u < split - 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  acc[i][j][4 * q + 0] += __uint_as_float(v[u][q].x); acc[i][j][4 * q + 1] += __uint_as_float(v[u][q].y);
                  acc[i][j][4 * q + 2] += __uint_as_float(v[u][q].z); acc[i][j][4 * q + 3] += __uint_as_float(v[u][q].w);
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
  }

  // ---- optional: pull a later launch's weight towards the memory-side cache (da_gemm_params.prefetch; see gemm2_kernel.cuh).
  // Buffer-addressed builds only (they own the scratch KiB behind the ring); every epilogue operand of those builds was fetched
  // before the K loop, so nothing queues behind these reads ----
  if constexpr (SM == 2 && !SPLITK && !LNFOLD) {
    if (p.prefetch) {
      const int nchunk = (int)min((long long)0x7fffffff >> 10, p.prefetch_bytes >> 10);
      __amdgpu_buffer_rsrc_t rs_pf = uniform_rsrc(p.prefetch, (size_t)nchunk << 10);
      const int stride = (int)gridDim.x * (WM * WN);
      int c = (int)blockIdx.x * (WM * WN) + wave;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (c < nchunk)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_pf, (__attribute__((address_space(3))) void*)(smem + (BM + BN) * 128 * STAGES), 16,
                                                   lane * 16, c << 10, 0, 0);
        c += stride;
      }
    }
  }

  // ---- epilogue: lane holds, for output row m (= lane&31 within the 32-tile), channels 8*(r>>2)+4*hi+(r&3) ----
  const uint16_t* __restrict__ bias = (const uint16_t*)p.bias;
  const uint16_t* __restrict__ rowvec = (const uint16_t*)p.rowvec;
  const uint16_t* __restrict__ resid = (const uint16_t*)p.residual;
  const uint16_t* __restrict__ bias_rows = (const uint16_t*)p.bias_rows;
  const uint16_t* __restrict__ gate = (const uint16_t*)p.gate;
  const bool geglu = (p.act == DA_ACT_GEGLU || p.act == DA_ACT_GEGLU_TANH);
  float st_sum[MT], st_sq[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) st_sum[i] = 0.f, st_sq[i] = 0.f;
  // (a lambda over a compile-time row-tile index, not a loop: the 4 x 2 wave tile's loop was not unrolled any more once the
  // body grew, and its dynamically indexed accumulators went through 512 B of scratch per lane)
  auto epilogue_rows = [&](auto i_c) {
    constexpr int i = decltype(i_c)::value;
    const int m = m0 + (wm * MT + i) * 32 + l31;
    if (m >= p.M) return;
    const int bidx = (rowvec != nullptr || gate != nullptr) ? (m / p.rows_per_batch) : 0;
    const float brow = bias_rows ? bf2f(bias_rows[m]) : 0.f;
    if (geglu) {
      // packed weight rows: per 64 rows = [32 value rows | 32 gate rows]; tile pair (2jp, 2jp+1) = (value, gate)
      if constexpr ((NT & 1) == 0) {
#pragma unroll
        for (int jp = 0; jp < NT / 2; ++jp) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int cin = 8 * g + 4 * hi;                                   // channel inside the 32-wide half
            const int nv = n0 + (wn * NT + 2 * jp) * 32 + cin;                // packed row of value
            const int no = (n0 >> 1) + (wn * (NT / 2) + jp) * 32 + cin;       // output column
            if (nv >= p.N) continue;
            float o[4];
            float lsv[4] = {0.f, 0.f, 0.f, 0.f}, lcv[4] = {0.f, 0.f, 0.f, 0.f}, lsg[4] = {0.f, 0.f, 0.f, 0.f}, lcg[4] = {0.f, 0.f, 0.f, 0.f};
            const bool fold = LNF && p.ln_stats != nullptr;
            if (fold) {   // LayerNorm fold (see da_gemm_params): s / c of the value and the gate rows, 16 bytes each
              const float4 a = *(const float4*)(p.ln_s + nv), b = *(const float4*)(p.ln_c + nv);
              const float4 c = *(const float4*)(p.ln_s + nv + 32), d = *(const float4*)(p.ln_c + nv + 32);
              lsv[0] = a.x; lsv[1] = a.y; lsv[2] = a.z; lsv[3] = a.w; lcv[0] = b.x; lcv[1] = b.y; lcv[2] = b.z; lcv[3] = b.w;
              lsg[0] = c.x; lsg[1] = c.y; lsg[2] = c.z; lsg[3] = c.w; lcg[0] = d.x; lcg[1] = d.y; lcg[2] = d.z; lcg[3] = d.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float hv = acc[i][2 * jp][4 * g + e] * p.alpha;
              float gv = acc[i][2 * jp + 1][4 * g + e] * p.alpha;
              if (fold) {
                hv = ln_rs[i] * (hv - ln_mu[i] * lsv[e]) + lcv[e];
                gv = ln_rs[i] * (gv - ln_mu[i] * lsg[e]) + lcg[e];
              }
              if (bias) {
                uint2 bh, bg;
                if constexpr (RES_PF) {
                  bh = bias_v[2 * jp][g], bg = bias_v[2 * jp + 1][g];
                } else {
                  bh = *(const uint2*)(bias + nv), bg = *(const uint2*)(bias + nv + 32);
                }
                hv += (e == 0) ? bf_lo(bh.x) : (e == 1) ? bf_hi(bh.x) : (e == 2) ? bf_lo(bh.y) : bf_hi(bh.y);
                gv += (e == 0) ? bf_lo(bg.x) : (e == 1) ? bf_hi(bg.x) : (e == 2) ? bf_lo(bg.y) : bf_hi(bg.y);
              }
              // reference rounds the projection to bf16 before chunk/gelu/mul (activations.py:113-124)
              hv = bf2f(f2bf(hv));
              gv = bf2f(f2bf(gv));
              o[e] = hv * bf2f(f2bf(p.act == DA_ACT_GEGLU ? gelu_erf_f(gv) : gelu_tanh_f(gv)));
            }
            uint2 pk;
            pk.x = pack_bf2(o[0], o[1]);
            pk.y = pack_bf2(o[2], o[3]);
            *(uint2*)((uint16_t*)p.C + (size_t)m * p.ldc + no) = pk;
          }
        }
      }
      return;
    }
    float st1 = 0.f, st2 = 0.f;   // LayerNorm fold, producer side: this lane's share of the row's (sum, sum of squares)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + (wn * NT + j) * 32 + 8 * g + 4 * hi;
        if (n >= p.N) continue;  // N is a multiple of 4 (checked on the host)
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[i][j][4 * g + e] * p.alpha;
        if (LNF && p.ln_stats) {
          const float4 sv = *(const float4*)(p.ln_s + n), cv = *(const float4*)(p.ln_c + n);
          o[0] = ln_rs[i] * (o[0] - ln_mu[i] * sv.x) + cv.x;
          o[1] = ln_rs[i] * (o[1] - ln_mu[i] * sv.y) + cv.y;
          o[2] = ln_rs[i] * (o[2] - ln_mu[i] * sv.z) + cv.z;
          o[3] = ln_rs[i] * (o[3] - ln_mu[i] * sv.w) + cv.w;
        }
        if (bias) {
          uint2 bv;
          if constexpr (RES_PF) bv = bias_v[j][g];
          else bv = *(const uint2*)(bias + n);
          o[0] += bf_lo(bv.x); o[1] += bf_hi(bv.x); o[2] += bf_lo(bv.y); o[3] += bf_hi(bv.y);
        }
        if (bias_rows) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += brow;
        }
        if (rowvec) {
          uint2 rv;
          if constexpr (RV_PF) rv = rowvec_v[i][j][g];
          else rv = *(const uint2*)(rowvec + (size_t)bidx * p.ld_rowvec + n);
          o[0] += bf_lo(rv.x); o[1] += bf_hi(rv.x); o[2] += bf_lo(rv.y); o[3] += bf_hi(rv.y);
        }
        if (p.act == DA_ACT_GELU_TANH) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = gelu_tanh_f(bf2f(f2bf(o[e])));
        } else if (p.act == DA_ACT_SILU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = silu_f(bf2f(f2bf(o[e])));
        } else if (p.act == DA_ACT_GELU_ERF) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = gelu_erf_f(bf2f(f2bf(o[e])));
        } else if (p.act == DA_ACT_QUICK_GELU) {
          // x * sigmoid(1.702 * x), each torch op rounded to bf16 as the reference's elementwise chain rounds it
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xv = bf2f(f2bf(o[e]));
            const float tv = bf2f(f2bf(1.702f * xv));
            o[e] = xv * bf2f(f2bf(1.0f / (1.0f + __expf(-tv))));
          }
        }
        if (gate && p.gate_f32) {
          // WanTransformerBlock: hidden.float() + Linear(x) (bf16) * gate (fp32), rounded once at the store
          const float4 gv = *(const float4*)((const float*)p.gate + (size_t)bidx * p.ld_gate + n);
          o[0] = bf2f(f2bf(o[0])) * gv.x; o[1] = bf2f(f2bf(o[1])) * gv.y;
          o[2] = bf2f(f2bf(o[2])) * gv.z; o[3] = bf2f(f2bf(o[3])) * gv.w;
        } else if (gate) {
          // reference: y = Linear(x) (bf16) ; g = gate * y (bf16) ; out = residual + g
          const uint2 gv = *(const uint2*)(gate + (size_t)bidx * p.ld_gate + n);
          o[0] = bf2f(f2bf(bf2f(f2bf(o[0])) * bf_lo(gv.x)));
          o[1] = bf2f(f2bf(bf2f(f2bf(o[1])) * bf_hi(gv.x)));
          o[2] = bf2f(f2bf(bf2f(f2bf(o[2])) * bf_lo(gv.y)));
          o[3] = bf2f(f2bf(bf2f(f2bf(o[3])) * bf_hi(gv.y)));
        }
        if (resid) {
          uint2 rv;
          if constexpr (RES_PF) rv = res_v[i][j][g];
          else rv = *(const uint2*)(resid + (size_t)m * p.ldr + n);
          o[0] += bf_lo(rv.x); o[1] += bf_hi(rv.x); o[2] += bf_lo(rv.y); o[3] += bf_hi(rv.y);
        }
        if (p.out_scale != 1.0f) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] *= p.out_scale;
        }
        if (p.out_f32) {
          *(float4*)((float*)p.C + (size_t)m * p.ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
          uint2 pk;
          pk.x = pack_bf2(o[0], o[1]);
          pk.y = pack_bf2(o[2], o[3]);
          *(uint2*)((uint16_t*)p.C + (size_t)m * p.ldc + n) = pk;
          if (LNF && p.stats_out) {   // statistics of the STORED (bf16-rounded) values, as a LayerNorm kernel would see them
            const float r0 = bf_lo(pk.x), r1 = bf_hi(pk.x), r2 = bf_lo(pk.y), r3 = bf_hi(pk.y);
            st1 += (r0 + r1) + (r2 + r3);
            st2 += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
          }
        }
      }
    }
    if (LNF && p.stats_out) {
      st_sum[i] = st1;
      st_sq[i] = st2;
    }
  };
  static_for<MT>(epilogue_rows);
  // LayerNorm fold, producer side: ONE partial per (row, column tile).  The WN waves of a block row hold disjoint column
  // ranges of the same rows: combine them through LDS (the ring is free once every wave has left the main loop) in wave
  // order -- fixed summation order, no atomics -- and let thread r write row r's pair.
  if constexpr (LNF) {
    if (p.stats_out) {
      float2* red = (float2*)smem;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float a = st_sum[i] + __shfl_xor(st_sum[i], 32, 64);
        const float b = st_sq[i] + __shfl_xor(st_sq[i], 32, 64);
        if (hi == 0) red[((wm * MT + i) * 32 + l31) * WN + wn] = make_float2(a, b);
      }
      __syncthreads();
      if (t < BM && m0 + t < p.M) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) {
          const float2 v = red[t * WN + w];
          a += v.x;
          b += v.y;
        }
        *(float2*)(p.stats_out + (size_t)(m0 + t) * p.stats_ld + 2 * tn) = make_float2(a, b);
      }
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}


// Column groups (1, 2, 4 or 8) of the per-XCD rectangles: minimise (rows + columns) of operand panels per XCD, weighted
// by the grid inflation a ragged split causes.
inline int choose_xcd_gx(int tiles_m, int tiles_n, int BM, int BN) {
  int best = 1;
  double best_cost = 1e300;
  for (int gx = 1; gx <= 8; gx *= 2) {
    const int gy = 8 / gx;
    const int tm_per = (tiles_m + gy - 1) / gy, tn_per = (tiles_n + gx - 1) / gx;
    const double inflation = (double)(8 * tm_per * tn_per) / ((double)tiles_m * tiles_n);
    const double cost = ((double)tm_per * BM + (double)tn_per * BN) * inflation * inflation;
    if (cost < best_cost) {
      best_cost = cost;
      best = gx;
    }
  }
  return best;
}

// CUs of the current device (256 on an MI355X), queried once
inline int compute_units() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      return 256;
    return cus;
  }();
  return n;
}

// blocks of one problem: 8 XCD rectangles x split_k
template <int BM, int BN>
inline int problem_grid(const da_gemm_params& p, int* gx_out) {
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int gx = choose_xcd_gx(tiles_m, tiles_n, BM, BN), gy = 8 / gx;
  *gx_out = gx;
  return 8 * ((tiles_m + gy - 1) / gy) * ((tiles_n + gx - 1) / gx) * (p.split_k > 1 ? p.split_k : 1);
}

// pb == nullptr: one problem.  Otherwise both problems run in ONE launch (see the kernel header).
template <int WM, int WN, int MT, int NT, int STAGES, bool CONV, int SM, bool SPLITK = false, bool KSKIP = false,
          bool LNFOLD = false>
int launch(const da_gemm_params& p, const da_gemm_params* pb, hipStream_t s) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  int gx_a = 1, gx_b = 1;
  const int grid_a = problem_grid<BM, BN>(p, &gx_a);
  const int grid_b = pb ? problem_grid<BM, BN>(*pb, &gx_b) : 0;
  const size_t lds = (size_t)(BM + BN) * 128 * STAGES + (SM == 2 ? 1024 : 0);   // ring (+ the prefetch scratch KiB)
  if (!SPLITK && p.split_k > 1) return DA_ERR_UNSUPPORTED;
  if (p.split_k > 1) {
    // the reducer block of a tile spins on its producers: every block of the launch must be co-resident (LDS and the
    // 1-2 blocks of 256 / 512 threads a CU takes at this kernel's register count bound it), and the workspace must
    // hold (split_k - 1) fp32 tiles per output tile
    if (pb) return DA_ERR_UNSUPPORTED;
    const int per_cu = (int)((160 * 1024) / lds) >= 2 && WM * WN == 4 ? 2 : 1;
    if (grid_a > compute_units() * per_cu) return DA_ERR_UNSUPPORTED;
    const size_t tiles = (size_t)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    if (!p.workspace || !p.sync_flags || (p.K >> 6) < p.split_k) return DA_ERR_INVALID;
    if (tiles * (p.split_k - 1) * BM * BN * 4 > (size_t)p.workspace_bytes) return DA_ERR_INVALID;
    if (tiles * (p.split_k - 1) > DA_SPLITK_ERR_SLOT) return DA_ERR_INVALID;
  }
  auto kern = igemm_bf16_kernel<WM, WN, MT, NT, STAGES, CONV, SM, SPLITK, KSKIP, LNFOLD>;
  if (lds > 48 * 1024) {
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DA_ERR_LAUNCH;
      attr_set = true;
    }
  }
  DA_LAUNCH(kern, dim3(grid_a + grid_b), dim3(64 * WM * WN), lds, s, p, gx_a, pb ? *pb : p, gx_b, grid_a);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// A/B switch for measurements: DA_GEMM_FLAT_STAGING=1 in the environment keeps nn.Linear on per-lane pointers (mode 1).
inline bool flat_staging_forced() {
  static const bool forced = [] {
    const char* e = getenv("DA_GEMM_FLAT_STAGING");
    return e && e[0] == '1';
  }();
  return forced;
}

// 31-bit offset budget of the buffer-addressed mode.  Linear: 256 rows of a panel + the K offset.  Conv: the input
// pixels one 256-row tile can reach -- its own rows scaled by stride^2 (or / 4 when upsampling), one image row above and
// three below, an image seam -- times the channel count.
inline bool buffer_staging_fits(const da_gemm_params& p) {
  const size_t lim = 0x3fffffffull;
  if ((size_t)p.ldw * 512 >= lim || (size_t)p.K * 2 >= lim) return false;
  if (!p.conv) return (size_t)p.lda * 512 < lim;
  const size_t cmax = (size_t)(p.C1 > p.C2 ? p.C1 : p.C2);
  const size_t span = (size_t)256 * p.stride * p.stride + (size_t)(6 + 2 * p.stride) * p.Win + 64;
  return span * cmax * 2 < lim && (size_t)p.M / ((size_t)p.Hout * p.Wout) * p.Hin * p.Win < 0x7fffffffull;
}

// (tile, staging) -> kernel instantiation.  staging: 0 register staged (2 slots); 1..5 LDS-DMA with 2/3/4/6/8 ring slots.
template <bool CONV>
int dispatch(const da_gemm_params& p, int tile, int staging, hipStream_t s, const da_gemm_params* pb = nullptr) {
  // LDS-DMA variants use the buffer-addressed mode whenever a tile's operand panels fit 31-bit byte offsets (always,
  // for the shapes of this engine); per-lane pointers (mode 1) are the fallback
  const bool buf = !flat_staging_forced() && buffer_staging_fits(p) && (!pb || buffer_staging_fits(*pb));
  if (p.split_k > 1) {
    // split-K instantiations: buffer-addressed staging only, the tiles that can be short of blocks (>= 128 wide)
    if (!buf || pb) return DA_ERR_UNSUPPORTED;
#define DA_SK(T_, ST_, WM_, WN_, MT_, NT_, NS_) \
  if (tile == (T_) && staging == (ST_)) return launch<WM_, WN_, MT_, NT_, NS_, CONV, 2, true>(p, nullptr, s)
    DA_SK(DA_TILE_128x128, DA_STAGE_LDS_DIRECT, 2, 2, 2, 2, 2);
    DA_SK(DA_TILE_128x128, DA_STAGE_LDS_DIRECT3, 2, 2, 2, 2, 3);
    DA_SK(DA_TILE_256x128, DA_STAGE_LDS_DIRECT3, 4, 2, 2, 2, 3);
    DA_SK(DA_TILE_128x256, DA_STAGE_LDS_DIRECT3, 2, 4, 2, 2, 3);
    DA_SK(DA_TILE_128x64, DA_STAGE_LDS_DIRECT3, 2, 2, 2, 1, 3);
    DA_SK(DA_TILE_64x128, DA_STAGE_LDS_DIRECT3, 2, 2, 1, 2, 3);
    if constexpr (!CONV) {
      DA_SK(DA_TILE_128x128, DA_STAGE_LDS_DIRECT4, 2, 2, 2, 2, 4);
      DA_SK(DA_TILE_256x128, DA_STAGE_LDS_DIRECT, 4, 2, 2, 2, 2);
      DA_SK(DA_TILE_128x256, DA_STAGE_LDS_DIRECT, 2, 4, 2, 2, 2);
    }
#undef DA_SK
    return DA_ERR_UNSUPPORTED;
  }
  if (p.stats_out || p.ln_stats || (pb && (pb->stats_out || pb->ln_stats))) {
    // LayerNorm-fold instantiations (nn.Linear, buffer-addressed staging): every 4- and 8-wave tile with 2 ring slots, the
    // three 3-slot variants the SDXL shapes use
    if constexpr (CONV) {
      return DA_ERR_UNSUPPORTED;
    } else {
      if (!buf || pb || p.split_k > 1) return DA_ERR_UNSUPPORTED;
#define DA_LN(T_, ST_, WM_, WN_, MT_, NT_, NS_) \
  if (tile == (T_) && staging == (ST_)) return launch<WM_, WN_, MT_, NT_, NS_, false, 2, false, false, true>(p, nullptr, s)
      DA_LN(DA_TILE_128x128, DA_STAGE_LDS_DIRECT, 2, 2, 2, 2, 2);
      DA_LN(DA_TILE_64x128, DA_STAGE_LDS_DIRECT, 2, 2, 1, 2, 2);
      DA_LN(DA_TILE_128x64, DA_STAGE_LDS_DIRECT, 2, 2, 2, 1, 2);
      DA_LN(DA_TILE_64x64, DA_STAGE_LDS_DIRECT, 2, 2, 1, 1, 2);
      DA_LN(DA_TILE_256x128, DA_STAGE_LDS_DIRECT, 4, 2, 2, 2, 2);
      DA_LN(DA_TILE_128x256, DA_STAGE_LDS_DIRECT, 2, 4, 2, 2, 2);
      DA_LN(DA_TILE_128x128_W8, DA_STAGE_LDS_DIRECT, 2, 4, 2, 1, 2);
      DA_LN(DA_TILE_128x64, DA_STAGE_LDS_DIRECT3, 2, 2, 2, 1, 3);
      DA_LN(DA_TILE_256x128, DA_STAGE_LDS_DIRECT3, 4, 2, 2, 2, 3);
      DA_LN(DA_TILE_128x256, DA_STAGE_LDS_DIRECT3, 2, 4, 2, 2, 3);
#undef DA_LN
      return DA_ERR_UNSUPPORTED;
    }
  }
  if (p.k_valid > 0 && buf && !pb) {
    // instantiations that skip the MFMA steps over channel padding (the tiles the large video convs use); any other
    // variant runs the problem too -- it multiplies the zeros
#define DA_KS(T_, ST_, WM_, WN_, MT_, NT_, NS_) \
  if (tile == (T_) && staging == (ST_)) return launch<WM_, WN_, MT_, NT_, NS_, CONV, 2, false, true>(p, nullptr, s)
    DA_KS(DA_TILE_128x128, DA_STAGE_LDS_DIRECT, 2, 2, 2, 2, 2);
    DA_KS(DA_TILE_128x64, DA_STAGE_LDS_DIRECT, 2, 2, 2, 1, 2);
    DA_KS(DA_TILE_64x128, DA_STAGE_LDS_DIRECT, 2, 2, 1, 2, 2);
    DA_KS(DA_TILE_256x128, DA_STAGE_LDS_DIRECT3, 4, 2, 2, 2, 3);
    DA_KS(DA_TILE_128x256, DA_STAGE_LDS_DIRECT3, 2, 4, 2, 2, 3);
    DA_KS(DA_TILE_256x256, DA_STAGE_LDS_DIRECT, 2, 4, 4, 2, 2);
    DA_KS(DA_TILE_128x128_W8, DA_STAGE_LDS_DIRECT, 2, 4, 2, 1, 2);
    DA_KS(DA_TILE_128x128_W8, DA_STAGE_LDS_DIRECT3, 2, 4, 2, 1, 3);
#undef DA_KS
  }
#define DA_V(WM_, WN_, MT_, NT_, ST_, G_)                                                         \
  do {                                                                                            \
    if constexpr (G_) {                                                                           \
      if (buf) return launch<WM_, WN_, MT_, NT_, ST_, CONV, 2>(p, pb, s);                         \
    }                                                                                             \
    return launch<WM_, WN_, MT_, NT_, ST_, CONV, (G_) ? 1 : 0>(p, pb, s);                         \
  } while (0)
  switch (staging) {
    case DA_STAGE_REGISTER:
      switch (tile) {
        case DA_TILE_128x128: DA_V(2, 2, 2, 2, 2, false);
        case DA_TILE_64x128: DA_V(2, 2, 1, 2, 2, false);
        case DA_TILE_128x64: DA_V(2, 2, 2, 1, 2, false);
        case DA_TILE_64x64: DA_V(2, 2, 1, 1, 2, false);
      }
      return DA_ERR_UNSUPPORTED;
    case DA_STAGE_LDS_DIRECT:
      switch (tile) {
        case DA_TILE_128x128: DA_V(2, 2, 2, 2, 2, true);
        case DA_TILE_64x128: DA_V(2, 2, 1, 2, 2, true);
        case DA_TILE_128x64: DA_V(2, 2, 2, 1, 2, true);
        case DA_TILE_64x64: DA_V(2, 2, 1, 1, 2, true);
        case DA_TILE_256x128: DA_V(4, 2, 2, 2, 2, true);
        case DA_TILE_128x256: DA_V(2, 4, 2, 2, 2, true);
        case DA_TILE_256x256: DA_V(2, 4, 4, 2, 2, true);
        case DA_TILE_128x128_W8: DA_V(2, 4, 2, 1, 2, true);
      }
      return DA_ERR_UNSUPPORTED;
    case DA_STAGE_LDS_DIRECT3:
      switch (tile) {
        case DA_TILE_128x128: DA_V(2, 2, 2, 2, 3, true);
        case DA_TILE_64x128: DA_V(2, 2, 1, 2, 3, true);
        case DA_TILE_128x64: DA_V(2, 2, 2, 1, 3, true);
        case DA_TILE_64x64: DA_V(2, 2, 1, 1, 3, true);
        case DA_TILE_256x128: DA_V(4, 2, 2, 2, 3, true);
        case DA_TILE_128x256: DA_V(2, 4, 2, 2, 3, true);
        case DA_TILE_128x128_W8: DA_V(2, 4, 2, 1, 3, true);
      }
      return DA_ERR_UNSUPPORTED;  // 256x256 x 3 slots would need 192 KiB of LDS
    case DA_STAGE_LDS_DIRECT4:
      switch (tile) {
        case DA_TILE_128x128: DA_V(2, 2, 2, 2, 4, true);
        case DA_TILE_64x128: DA_V(2, 2, 1, 2, 4, true);
        case DA_TILE_128x64: DA_V(2, 2, 2, 1, 4, true);
        case DA_TILE_64x64: DA_V(2, 2, 1, 1, 4, true);
        case DA_TILE_128x128_W8: DA_V(2, 4, 2, 1, 4, true);
      }
      return DA_ERR_UNSUPPORTED;
    case DA_STAGE_LDS_DIRECT6:
      switch (tile) {
        case DA_TILE_64x128: DA_V(2, 2, 1, 2, 6, true);
        case DA_TILE_128x64: DA_V(2, 2, 2, 1, 6, true);
        case DA_TILE_64x64: DA_V(2, 2, 1, 1, 6, true);
      }
      return DA_ERR_UNSUPPORTED;
    case DA_STAGE_LDS_DIRECT8:
      switch (tile) {
        case DA_TILE_64x64: DA_V(2, 2, 1, 1, 8, true);
      }
      return DA_ERR_UNSUPPORTED;
  }
#undef DA_V
  // the ping-pong stagings are codes of the K2 family (gemm2_kernel.cuh): known, not built here
  return (staging == DA_STAGE_PINGPONG || staging == DA_STAGE_PINGPONG3) ? DA_ERR_UNSUPPORTED : DA_ERR_INVALID;
}

}  // namespace da_gemm
