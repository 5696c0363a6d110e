// Second GEMM / implicit-GEMM family ("K2"): 8 waves per workgroup = 2 K-GROUPS x (WM x WN) waves, v_mfma_f32_16x16x32_bf16.
//
// Why it exists (profiles/r03a_ceiling_table.md, r02e_kernel_experiments.md).  The first family (gemm_kernel.cuh) is a
// 4-wave tile with one rendezvous per 64-wide K slice: a workgroup ALONE on its CU spends ~900-1500 cycles per slice for
// 128-512 cycles of MFMA work (the slice's ds_read latency, the barrier and the wait for the LDS-DMA sit in front of the
// MFMAs of every slice), so it only approaches the matrix pipe's rate when 2-3 workgroups share a CU -- and the SDXL
// shapes (M = 2048 tokens) do not have the tiles for that: 2048 x 1280 is 320 tiles of 128 x 64 on 256 CUs.  This family
// is built for ONE workgroup per CU:
//   * tile shapes in 16-column steps (128 x 80: 2048 x 1280 = exactly 256 tiles, one per CU; 128 x 160 for the 640-wide
//     level; 128 x 128 for the GEGLU projection: 5 per CU);
//   * the two K-groups take alternate K slices of the SAME output tile (even / odd), each with its own accumulators, so a
//     SIMD always holds two waves with independent MFMA streams; their partial sums meet once, through LDS, after the loop,
//     and each group finishes half of the tile's rows (both halves of the epilogue run in parallel);
//   * K slices are staged in PAIRS (one per group) and there is ONE rendezvous per pair, placed in the MIDDLE of the
//     pair's MFMAs: k-step 0 | wait + s_barrier + LDS-DMA of a later pair + ds_reads of the next pair's k-step 0 | k-step 1.
//     At the rendezvous every fragment of the current pair is already in registers, so (a) its ring slot is free at once
//     and (b) the next pair's first fragments arrive under this pair's second-half MFMAs: the matrix stream of a wave is
//     continuous across slices, there is no per-slice bubble.
// Staging, LDS image, swizzle and the implicit-GEMM addressing are those of gemm_kernel.cuh (buffer-addressed LDS-DMA,
// 1 KiB pieces of 8 rows x 128 B, 16-byte slots XOR-swizzled with (row >> 1) & 7 on the SOURCE side): the same image is
// conflict-free for the 16 x 16 x 32 fragment reads (lane = row & 15, 16-byte chunk (lane >> 4) + 4 * k-step).
//
// Numerics: fp32 accumulation, K summed as (even slices) + (odd slices); every K2 tile shape gives bit-identical results
// to every other K2 shape; against the first family the fp32 summation order differs (last-bit differences before the
// bf16 rounding of the output).
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.cuh"
#include "diffusers_amd.h"
#include "gemm2_shared.cuh"

namespace da_gemm2 {


// Stage timestamps (tools/trace_gemm2.hip builds this header with -DDA_GEMM2_TRACE; the library never does): s_memtime at the
// marks below, kept in SGPRs and written by lane 0 of every wave after its last output store.
#if defined(DA_GEMM2_TRACE)
__device__ unsigned long long* g_da2_trace;
#define DA2_TRACE(i) tr_[i] = __builtin_readcyclecounter()
#else
#define DA2_TRACE(i) ((void)0)
#endif



// KG K-groups of WM x WN waves (KG * WM * WN == 8), each wave owns MT x NT tiles of 16 x 16  ->  block tile (16*MT*WM) x (16*NT*WN).
// KG == 2: the design described above (two groups of four waves on alternate K slices; the ring unit is a slice PAIR).
// KG == 1: the same loop with all eight waves on every slice (ring unit = one slice, no exchange after the loop): the large
//          tiles (128 x 320, 256 x 128) of problems with several tiles per CU, where what binds a 128 x 128 tile is the
//          L2 -> LDS traffic per flop, (1 / BM + 1 / BN) bytes, not the latency of a lone workgroup.
// NSLOT = ring depth in units (2 or 3).
// PP (KG == 2 only) = the two K-groups run half an iteration APART ("ping-pong"): each group stages only its own slice parity
//          (four waves x (PX + PW) / 4 pieces), every iteration has two block barriers, and group 1 entered the loop one barrier
//          late -- so the LDS-DMA issue of one group (what binds the in-phase loop: all eight waves queue their loads on the CU's
//          one texture-address path right behind the rendezvous, profiles/r03d_pmc_sq_counters.md) always runs under the
//          other group's MFMA-only half.
// STREAMW (KG == 1, 2-slice ring) = "streaming W": a wave tile too large to hold both k-steps' fragments next to its accumulators
//          (64 x 160: 160 accumulator registers) keeps the slice's X fragments and streams the W fragments through a short
//          register queue, each feeding MT MFMAs; one rendezvous per slice, the next-but-one slice's LDS-DMA spread under
//          the slice's MFMAs.  The 256 x 320 tile this enables covers M 2048 x N 10240 (SDXL's GEGLU projection) in exactly
//          one round of 256 workgroups at 1 / 256 + 1 / 320 staged bytes per flop.
// GIL (KG == 1, two wave columns) = GEGLU-interleaved column ownership: of each 64-column [32 value | 32 gate] group of the packed weight a
//          wave owns the 16-column tiles {wn, wn + 2} (one value tile and ITS gate tile), so a wave tile whose width (160)
//          is not a multiple of 64 can still finish GEGLU on its own accumulators.
// LNF = LayerNorm fold (da_gemm_params.stats_out / ln_*), nn.Linear only, in its OWN instantiations: carried by every kernel the
//          fold's registers cost the register-heaviest tiles 5-9 % (profiles/r02d_layernorm_fold.md).
//          Producer (stats_out): the row-contiguous store path also sums the bf16-ROUNDED values it stores (what a LayerNorm
//          kernel would read): every lane's 8 channels -> LDS -> 16 lanes add the PPR pieces of their row in piece order
//          (deterministic) and write ONE (sum, sum of squares) pair per (row, 16 NH-column band).
//          Consumer (ln_stats): mean / rstd of the rows a lane finishes, formed from the producer's partials (loaded with the
//          other epilogue operands, behind the first LDS-DMA), and rstd * (acc - mean * s[n]) + c[n] applied to every
//          accumulator group in front of the bias.
// XA > 0 (round 5; one instantiation: the 128 x 128 two-K-group tile with the LayerNorm fold) = cross-attention in the epilogue
//          of attn2.to_q (attention_processor.py:2743-2777 with encoder_hidden_states of <= 16 * XA text tokens): after the exchange a
//          wave holds the finished q of 32 queries x ONE 64-wide head in the MFMA C layout -- which IS the B-operand layout of
//          S^T = K . Q^T for a permuted order of the head's channels -- so the 77-key attention (scores, softmax, P . V) runs on the
//          wave's own registers against K / V^T of the tile's two heads (LDS-DMA into the free ring), and the launch writes
//          softmax(q k^T) v instead of q: one launch where there were two (to_q, flash attention) and no q round trip.
template <int KG, int WM, int WN, int MT, int NT, int NSLOT, bool CONV, bool PP = false, bool STREAMW = false, bool GIL = false,
          bool LNF = false, int XA = 0>
__global__ __launch_bounds__(512) void igemm2_bf16_kernel(const da_gemm_params p, const int xcd_gx_chunk) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int xcd_gx = xcd_gx_chunk & 255;                  // XCD columns; bits 8.. : conv channel-chunk size in 64-wide slices (0: whole)
  static_assert((KG == 1 || KG == 2) && KG * WM * WN == 8, "eight waves: one or two K-groups");
  constexpr int BM = 16 * MT * WM, BN = 16 * NT * WN;
  constexpr int PX = BM / 8, PW = BN / 8;                 // 1 KiB pieces (8 rows x 128 B) per slice
  constexpr int XBYTES = BM * 128, SLICE = (BM + BN) * 128, PAIR = KG * SLICE;   // PAIR = the ring unit (KG slices)
  static_assert(!PP || KG == 2, "ping-pong is a property of the two K-groups");
  static_assert(!STREAMW || (KG == 1 && NSLOT == 2 && !CONV), "streaming-W loop: one K-group, two-slice ring, nn.Linear");
  static_assert(!GIL || (KG == 1 && !CONV && WN == 2 && (NT % 2) == 0), "GEGLU-interleaved ownership: two wave columns, value / gate tile pairs");
  static_assert(!LNF || (!CONV && !STREAMW), "LayerNorm fold: nn.Linear instantiations of the ring loops");
  static_assert(XA == 0 || (LNF && KG == 2 && WM == 2 && WN == 2 && MT == 4 && NT == 4 && NSLOT == 2 && XA <= 8),
                "cross-attention epilogue: the 128 x 128 two-K-group tile (a wave finishes 32 rows x one 64-wide head)");
  constexpr int SW = PP ? 4 : 8;                          // waves that share the staging of one unit (PP: one slice, own group)
  constexpr int UX = PP ? PX : KG * PX, UW = PP ? PW : KG * PW;   // pieces of that unit
  constexpr int XI = (UX + SW - 1) / SW, WI = (UW + SW - 1) / SW;   // LDS-DMA instructions per wave per unit (upper bound)
  constexpr int XRAG = UX % SW ? 1 : 0, WRAG = UW % SW ? 1 : 0;   // a last row of pieces only some waves own
  static_assert(NSLOT == 2 || NSLOT == 3, "ring of 2 or 3 slice pairs");
  constexpr int DUMP = 1024;                               // where the out-of-range pieces of ragged tiles and the prefetch land
  constexpr int LNSC = LNF ? 2 * BN * 4 : 0;               // LNF consumer: the block's s[n] / c[n] (fp32), behind the dump KiB
  // XA: K / V^T of the tile's two heads in the upper half of the ring once the K loop is over (the lower half carries the partial-sum
  // exchange, then the output staging)
  constexpr int XA_K_OFF = 4 * MT * NT * 1024, XA_KB = XA * 16 * 128, XA_V_OFF = XA_K_OFF + 32768;
  static_assert(XA == 0 || (2 * XA_KB <= 32768 && XA_V_OFF + 32768 <= NSLOT * PAIR), "cross-attention operands do not fit the ring");
  static_assert(NSLOT * PAIR + DUMP + LNSC <= 160 * 1024, "LDS ring exceeds the 160 KiB of a CU");
  static_assert(BM % 8 == 0 && BN % 8 == 0, "tile rows are staged in 8-row pieces");
  static_assert(!CONV || (UX % SW) == 0, "conv: the slice of an activation piece must be a compile-time constant");
  static_assert((NSLOT - 1) * (XI + WI) <= 63, "vmcnt is a 6-bit counter");
  static_assert(KG == 1 || MT * NT * 1024 * 4 <= NSLOT * PAIR, "partial-sum exchange does not fit the ring");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

#if defined(DA_GEMM2_TRACE)
  unsigned long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  DA2_TRACE(0);                                           // kernel entry
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = KG == 2 ? wave >> 2 : 0, wq = KG == 2 ? wave & 3 : wave;   // K-group, position inside the group
  const int wm = wq / WN, wn = wq - wm * WN;
  const int r16 = lane & 15, kq = lane >> 4;

  // ---- XCD-aware tile mapping (as gemm_kernel.cuh: block b runs on XCD b % 8, each XCD owns a rectangle of tiles) ----
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int gyn = 8 / xcd_gx;
  const int tm_per = (tiles_m + gyn - 1) / gyn, tn_per = (tiles_n + xcd_gx - 1) / xcd_gx;
  const int bid = (int)blockIdx.x;
  const int xcd = bid & 7, kblk = bid >> 3;
  const int gy = xcd / xcd_gx, gx = xcd - gy * xcd_gx;
  const int lm = kblk / tn_per, ln = kblk - lm * tn_per;
  const int tm = gy * tm_per + lm, tn = gx * tn_per + ln;
  if (tm >= tiles_m || tn >= tiles_n) return;             // ragged rectangle: the whole block leaves before any barrier
  const int m0 = tm * BM, n0 = tn * BN;

  const uint16_t* __restrict__ A = (const uint16_t*)p.A;
  const uint16_t* __restrict__ A2 = (const uint16_t*)p.A2;
  const uint16_t* __restrict__ Wt = (const uint16_t*)p.W;
  const int nk = p.K >> 6;                                // K slices
  const int nk2 = (nk + KG - 1) / KG;                     // ring units (slice pairs for KG == 2)
  const int Ctot = CONV ? (p.C1 + p.C2) : 0;
  const int Hv = CONV ? (p.Hin << p.up) : 0, Wv = CONV ? (p.Win << p.up) : 0;

  // ---- staging assignment: wave w stages pieces q = i * 8 + w of the pair's 2 * PX activation and 2 * PW weight pieces ----
  // piece q -> slice parity h = q / P, piece r = q % P inside the slice; lane -> row 8 r + (lane >> 3), 16-byte slot lane & 7
  int x_h[XI], x_r[XI], w_h[WI], w_r[WI];
  bool x_ok[XI], w_ok[WI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int q = i * SW + (PP ? wq : wave);
    // compile-time constants wherever all staging waves of a row i agree (so the unrolled issue code has no branch / select)
    x_ok[i] = (i * SW + SW - 1 < UX) ? true : (q < UX);
    x_h[i] = (KG == 1 || PP) ? 0 : (i * 8 >= PX) ? 1 : (i * 8 + 7 < PX) ? 0 : (q >= PX ? 1 : 0);
    x_r[i] = q - x_h[i] * PX;
    if (!x_ok[i]) x_h[i] = 0, x_r[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int q = i * SW + (PP ? wq : wave);
    w_ok[i] = (i * SW + SW - 1 < UW) ? true : (q < UW);
    w_h[i] = (KG == 1 || PP) ? 0 : (i * 8 >= PW) ? 1 : (i * 8 + 7 < PW) ? 0 : (q >= PW ? 1 : 0);
    w_r[i] = q - w_h[i] * PW;
    if (!w_ok[i]) w_h[i] = 0, w_r[i] = 0;
  }
  // PP: this wave's staging state [0] describes its OWN group's slice parity; the LDS half it fills is its group's
  const int own_half = PP ? g * SLICE : 0;
  constexpr int my_loads = XI + WI;                       // LDS-DMA instructions per wave per pair (ragged pieces included)

  // conv: per staged activation row, the output pixel it belongs to
  int xr_base[XI], xr_oy[XI], xr_ox[XI];
  if constexpr (CONV) {
    const int hw = p.Hout * p.Wout;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int m = m0 + x_r[i] * 8 + (lane >> 3);
      if (m < p.M) {
        const int b = m / hw;
        const int rem = m - b * hw;
        const int oy = rem / p.Wout;
        xr_base[i] = b * p.Hin;
        xr_oy[i] = oy * p.stride - p.pad;
        xr_ox[i] = (rem - oy * p.Wout) * p.stride - p.pad;
      } else {
        xr_base[i] = 0;
        xr_oy[i] = -100000;
        xr_ox[i] = -100000;
      }
    }
  }

  // buffer descriptors over the block's operand panels (see gemm_kernel.cuh, staging mode 2)
  size_t xbase = 0, xbytes = 0x7fffffff, x2bytes = 0x7fffffff;
  int pb = 0;
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
  __amdgpu_buffer_rsrc_t rs_x = uniform_rsrc(A + (CONV ? (size_t)pb * p.C1 : xbase), xbytes);
  __amdgpu_buffer_rsrc_t rs_x2 = uniform_rsrc((CONV && A2) ? A2 + (size_t)pb * p.C2 : A, x2bytes);
  __amdgpu_buffer_rsrc_t rs_w = uniform_rsrc(Wt + (size_t)n0 * p.ldw, 0x7fffffff);

  int vo_x[XI], vo_x2[XI], vo_w[WI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int row = x_r[i] * 8 + (lane >> 3);
    const int sc = (lane & 7) ^ ((row >> 1) & 7);
    if constexpr (!CONV) vo_x[i] = (min(row, p.M - 1 - m0) * p.lda + sc * 8) * 2;
    else vo_x[i] = 0;
    vo_x2[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int row = w_r[i] * 8 + (lane >> 3);
    const int sc = (lane & 7) ^ ((row >> 1) & 7);
    vo_w[i] = (min(row, p.N - 1 - n0) * p.ldw + sc * 8) * 2;
  }
  // conv: one (kernel row, kernel column, first channel) cursor per slice parity; the per-lane offsets of the pieces of a
  // parity are recomputed only when its tap changes.  Pairs are staged strictly in order, so the state below always
  // describes the NEXT pair to stage (`st_pr`); it is advanced right after a pair's loads were issued, i.e. the address
  // arithmetic of pair n + 1 runs under the MFMAs that follow the issue of pair n and the issue itself is straight-line.
  // Round 6 -- K order of a k x k conv, optionally CHUNKED (per launch, where a tap sweep outgrows the L2: gemm2_shared.cuh conv_chunk_slices has the rule and the measurements).
  // The weight rows are [tap][channel] and K is walked in that order: all channels of tap 0, then all channels of tap 1 ...  The nine
  // taps read the SAME activation pixels shifted by one, but a tap's sweep over C channels is (CUs of an XCD) x (tile rows) x C x 2 B
  // -- 5 MB per XCD at C = 640 on 128-row tiles -- more than the XCD's 4 MB L2, so every tap re-fetches the activations through the
  // fabric: 404 MB fetched for a conv with 67 MB of operands (profiles/r06_sdxl_traffic.md, 6.4x).  Chunked: for each chunk of
  // `cchunk` channels, all taps, then the next chunk -- the taps of a chunk re-read 1-2 MB per XCD, which stays in L2.  Same slices,
  // another order of summation; cchunk >= C (the default) is the tap-major order.
  const int cchunk = CONV ? (((xcd_gx_chunk >> 8) > 0 && p.conv > 1) ? min((xcd_gx_chunk >> 8) * 64, Ctot) : Ctot) : 0;
  int c_kh[2] = {0, 0}, c_kw[2] = {0, 0}, c_c0[2] = {0, 0};
  int c_beg[2] = {0, 0}, c_end[2] = {cchunk, cchunk};     // channel range of the chunk each parity's cursor is in
  int st_pr = 0;                                          // next pair to stage
  int st_s[2] = {PP ? min(g, nk - 1) : 0, nk > 1 ? 1 : 0};   // slice (clamped to nk - 1) each parity stages next
  // a slice of the next pair that lies past K is staged as zeros (bit 31 of the per-lane offset: beyond num_records)
  int st_zero[2] = {(!PP || g < nk) ? 0 : (int)0x80000000, (nk > 1) ? 0 : (int)0x80000000};
  auto tap_offsets = [&](int h) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      if ((KG == 2 && !PP && i >= XI / 2 ? 1 : 0) != h) continue;   // CONV, in phase: pieces i < XI / 2 are parity 0
      const int row = x_r[i] * 8 + (lane >> 3);
      const int sc = (lane & 7) ^ ((row >> 1) & 7);
      const int iy = xr_oy[i] + c_kh[h], ix = xr_ox[i] + c_kw[h];
      const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
      const int rel = (xr_base[i] + (iy >> p.up)) * p.Win + (ix >> p.up) - pb;
      vo_x[i] = ok ? (rel * p.C1 + sc * 8) * 2 : (int)0x80000000;
      vo_x2[i] = ok ? (rel * p.C2 + sc * 8) * 2 : (int)0x80000000;
    }
  };
  auto cursor_step = [&](int h) {                         // advance parity h's cursor by one slice (wave-uniform)
    c_c0[h] += 64;
    if (c_c0[h] >= c_end[h]) {                            // this tap has seen the chunk's channels: next tap, same chunk
      if (++c_kw[h] >= p.conv) {
        c_kw[h] = 0;
        if (++c_kh[h] >= p.conv && c_end[h] < Ctot) {     // all taps of the chunk done: first tap of the next chunk
          c_kh[h] = 0;
          c_beg[h] = c_end[h];
          c_end[h] = min(c_end[h] + cchunk, Ctot);
        }
      }
      c_c0[h] = c_beg[h];
    }
  };
  // byte offset of the cursor's K slice inside a weight row ([tap][channel]: in the old order this was slice index * 128)
  auto w_off = [&](int h) { return ((c_kh[h] * p.conv + c_kw[h]) * Ctot + c_c0[h]) * 2; };
  if constexpr (CONV) {
    if (PP) {
      if (g == 1 && nk > 1) cursor_step(0);               // group 1 starts on slice 1
    } else if (KG == 2 && nk > 1) {
      cursor_step(1);                                     // parity 1 starts on slice 1
    }
    tap_offsets(0);
    if (KG == 2 && !PP) tap_offsets(1);
  }
  int st_woff[2] = {CONV ? w_off(0) : 0, CONV ? w_off(1) : 0};   // conv: weight-row byte offset of the slice each parity stages next

#define DA2_LDS(ptr) ((__attribute__((address_space(3))) void*)(ptr))
  // Issue the LDS-DMA of the next slice pair into ring slot `slot` (straight-line code).  A slice past the end of K (odd
  // slice count; the pairs "staged" by the last NSLOT rendezvous) is staged as ZEROS (bit 31 of the per-lane offset: beyond
  // num_records, the range check writes zeros to LDS without touching memory), so an odd slice count makes group 1 multiply
  // zeros once, every rendezvous issues the same number of loads (one vmcnt immediate) and the loop body has no branch.
  // Pieces only some waves own (ragged tiles) go first: their wave-uniform branch then sits in front of the straight-line part.
  auto stage_issue_base = [&](unsigned char* base, auto which_c) {
    constexpr int WHICH = decltype(which_c)::value;      // -1: every piece of the unit; k >= 0: the k-th piece only (streaming-W loop)
    auto issue_x = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int h = x_h[i];
      // (h may be a run-time value on ragged tiles: select, never index.)  A piece this wave does not own (ragged last row)
      // is issued all the same, out of range (zeros, no memory traffic) into the dump KiB behind the ring: every wave issues
      // the same loads, the loop body stays ONE basic block and the compiler's lgkmcnt / vmcnt bookkeeping stays exact.
      const int z = x_ok[i] ? (h ? st_zero[1] : st_zero[0]) : (int)0x80000000;
      unsigned char* dst = x_ok[i] ? base + own_half + h * SLICE + x_r[i] * 1024 : smem + NSLOT * PAIR;
      if constexpr (CONV) {
        const int c0 = h ? c_c0[1] : c_c0[0];
        if (p.C2 > 0 && c0 >= p.C1)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x2, DA2_LDS(dst), 16, vo_x2[i] | z, (c0 - p.C1) * 2, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, DA2_LDS(dst), 16, vo_x[i] | z, c0 * 2, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, DA2_LDS(dst), 16, vo_x[i] | z, (h ? st_s[1] : st_s[0]) * 128, 0, 0);
      }
    };
    auto issue_w = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int h = w_h[i];
      const int z = w_ok[i] ? (h ? st_zero[1] : st_zero[0]) : (int)0x80000000;
      unsigned char* dst = w_ok[i] ? base + own_half + h * SLICE + XBYTES + w_r[i] * 1024 : smem + NSLOT * PAIR;
      if constexpr (CONV)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, DA2_LDS(dst), 16, vo_w[i] | z, h ? st_woff[1] : st_woff[0], 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, DA2_LDS(dst), 16, vo_w[i] | z, (h ? st_s[1] : st_s[0]) * 128, 0, 0);
    };
    if constexpr (WHICH >= 0) {                           // weights (the HBM-cold operand) first
      if constexpr (WHICH < WI) issue_w(std::integral_constant<int, WHICH < WI ? WHICH : 0>{});
      else issue_x(std::integral_constant<int, WHICH >= WI ? WHICH - WI : 0>{});
    } else {
      static_for<XI>(issue_x);
      static_for<WI>(issue_w);
    }
  };
  auto stage_issue = [&](int slot) { stage_issue_base(smem + slot * PAIR, std::integral_constant<int, -1>{}); };
  auto stage_issue_one = [&](unsigned char* base, auto k_c) { stage_issue_base(base, k_c); };
  // Move the staging state to the following pair (scalar arithmetic; conv: a tap change recomputes that parity's offsets).
  auto stage_advance = [&]() {
    ++st_pr;
#pragma unroll
    for (int h = 0; h < (PP ? 1 : KG); ++h) {
      const int want = min(KG * st_pr + (PP ? g : h), nk - 1);
      if constexpr (CONV) {
        const int kh0 = c_kh[h], kw0 = c_kw[h];
        while (st_s[h] < want) {
          cursor_step(h);
          ++st_s[h];
        }
        if (c_kh[h] != kh0 || c_kw[h] != kw0) tap_offsets(h);
        st_woff[h] = w_off(h);
      } else {
        st_s[h] = want;
      }
    }
    st_zero[0] = (KG * st_pr + (PP ? g : 0) < nk) ? 0 : (int)0x80000000;
    st_zero[1] = (KG * st_pr + 1 < nk) ? 0 : (int)0x80000000;
  };
  // wait until at most PAIRS later pairs of THIS wave's LDS-DMA are in flight (a compile-time immediate)
#define DA2_WAIT_PAIRS(PAIRS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PAIRS) * my_loads) : "memory")

  // ---- epilogue operands of the half tile this wave will finish: fetched BEFORE the K loop (their round trip hides there) ----
  // After the loop group g keeps the row tiles [g * MH, (g + 1) * MH) (column tiles when MT is odd) and sends the rest.
  constexpr bool SPLIT_M = KG == 1 || (MT % 2) == 0;
  static_assert(SPLIT_M || (NT % 2) == 0, "one of the wave tile's dimensions must split between the two K-groups");
  constexpr int MH = (KG == 2 && SPLIT_M) ? MT / 2 : MT, NH = (KG == 1 || SPLIT_M) ? NT : NT / 2;
  const bool geglu = (p.act == DA_ACT_GEGLU || p.act == DA_ACT_GEGLU_TANH);
  auto row_of = [&](int ih) { return m0 + (wm * MT + (SPLIT_M ? g * MH : 0) + ih) * 16 + r16; };
  // 16-column tile `jh` of this wave inside the block tile (GIL: tiles {wn, wn + 2} of every group of four)
  auto col_tile = [&](int jh) { return GIL ? 4 * (jh >> 1) + 2 * (jh & 1) + wn : wn * NT + (SPLIT_M ? 0 : g * NH) + jh; };
  auto col_of = [&](int jh) { return n0 + col_tile(jh) * 16 + 4 * kq; };
  // (Only while the registers last: a wave that finishes more than 12 tiles -- the KG == 1 shapes -- fetches the residual and
  // the channel vector in the epilogue instead; its launches are long enough for that round trip not to matter.)
  constexpr bool PF = MH * NH <= (CONV ? 8 : 12);         // (conv carries its pixel cursors: fewer registers to spare)
  // Row-contiguous output (STG): a wave's 16 x (16 NH) band of finished values goes through LDS once so that every lane stores
  // (and, for the residual, loads) 16 contiguous bytes of ONE row -- NH * 32 contiguous bytes per row instead of NH separate
  // 32-byte segments per row.  The store path is bound by line requests, not bytes (profiles/r03e_gemm_stage_trace.md: the
  // 8-byte-per-lane epilogue cost 650-820 cycles per 16 x 16 tile, 13-27 % of a launch).  Piece q = lane + 64 t of a band:
  // row q / PPR, 16-byte (8-channel) slot q % PPR.  Needs 16-byte aligned rows; anything else keeps the 8-byte path.
  constexpr bool STG = NH >= 2;
  constexpr int PPR = 2 * NH, NPIECE = 16 * PPR, TT = (NPIECE + 63) / 64, ROWB = NH * 64 + 32;   // (ROWB / 2: the packed rows)
  static_assert(!STG || (KG == 2 ? 4 * MT * NT * 1024 : 0) + 8 * 16 * ROWB <= NSLOT * PAIR, "output staging does not fit the ring");
  constexpr int STATS_OFF = (KG == 2 ? 4 * MT * NT * 1024 : 0) + 8 * 16 * ROWB;   // LNF producer: 8 waves x TT * 64 float2 slots
  static_assert(!(LNF && STG) || STATS_OFF + 8 * TT * 64 * 8 <= NSLOT * PAIR, "statistics scratch does not fit the ring");
  const bool wide = STG && !GIL && !geglu && !p.out_f32 && !(p.ldc & 7) && !((size_t)p.C & 15) && !(p.N & 7) &&
                    (!p.residual || (!(p.ldr & 7) && !((size_t)p.residual & 15)));
  const int band_col0 = n0 + (wn * NT + (SPLIT_M ? 0 : g * NH)) * 16;
  auto band_row0 = [&](int ih) { return m0 + (wm * MT + (SPLIT_M ? g * MH : 0) + ih) * 16; };
  constexpr bool PFN = PF && !STG;                        // the 8-byte layout prefetches its residual only where it is the only path
  uint2 res_v[PFN ? MH : 1][PFN ? NH : 1], bias_v[NH], rowvec_v[PF ? MH : 1][PF ? NH : 1];
  uint4 resw[(PF && STG) ? MH : 1][(PF && STG) ? TT : 1];
#pragma unroll
  for (int ih = 0; ih < (PF ? MH : 1); ++ih)
#pragma unroll
    for (int jh = 0; jh < (PF ? NH : 1); ++jh) rowvec_v[ih][jh] = make_uint2(0, 0);
#pragma unroll
  for (int ih = 0; ih < (PFN ? MH : 1); ++ih)
#pragma unroll
    for (int jh = 0; jh < (PFN ? NH : 1); ++jh) res_v[ih][jh] = make_uint2(0, 0);
#pragma unroll
  for (int ih = 0; ih < ((PF && STG) ? MH : 1); ++ih)
#pragma unroll
    for (int t = 0; t < ((PF && STG) ? TT : 1); ++t) resw[ih][t] = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int jh = 0; jh < NH; ++jh) bias_v[jh] = make_uint2(0, 0);
  // LayerNorm fold, consumer side: the 24 (sum, sum of squares) slots of a row this lane finishes are read by its four kq lanes,
  // six slots (three 16-byte loads) each; rows hold DA_LN_MAX_PARTS slots, the ones past ln_parts are masked when summed.
  // The slots are requested with the other epilogue operands (behind the first LDS-DMA) and REDUCED to (mean, rstd) right behind the
  // first rendezvous -- they have landed with pair 0 -- so that only two floats per row ride through the K loop.
  constexpr bool LN_PF = LNF;
  float4 lnq[LN_PF ? MH : 1][3];
#pragma unroll
  for (int ih = 0; ih < (LN_PF ? MH : 1); ++ih)
#pragma unroll
    for (int u = 0; u < 3; ++u) lnq[ih][u] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto ln_load = [&](int ih, float4* dst) {
    const int m = min(row_of(ih), p.M - 1);
    const float* sp = p.ln_stats + (size_t)m * p.ln_stats_ld;
    // Unconditional loads (a predicated load whose other arm writes zeros makes the compiler wait for the load before the zero
    // write: vmcnt(0) in the prologue); a pair past the valid slots re-reads the row's first pair instead (masked when summed), so
    // that 16 partials touch only the first 128-byte line of the row.
#pragma unroll
    for (int u = 0; u < 3; ++u) dst[u] = *(const float4*)(sp + ((6 * kq + 2 * u < p.ln_parts) ? 12 * kq + 4 * u : 0));
  };
  // Issued right behind the FIRST pair's LDS-DMA (the matrix stream starts first; these loads are older than pairs 1 .. and
  // therefore covered by every counted wait that covers pair 0).
  float4 lnsc_v = make_float4(0.f, 0.f, 0.f, 0.f);
  auto prefetch_epilogue_operands = [&]() {
    if constexpr (LNF) {
      if (p.ln_stats) {
        // the block's BN columns of s and c go to LDS once (every lane of the epilogue reads 2 x 16 bytes per 4-column group:
        // from global memory that was a chain of L2 round trips, + 10 us on the GEGLU projection); visible after the first
        // rendezvous of the K loop, read after the last
        // (requested here, written to LDS behind the first rendezvous -- by then it has landed with pair 0; writing it at once
        // parked every workgroup for the round trip in front of its K loop: + 2.7 k cycles per workgroup,
        // profiles/r04i_gemm_stage_trace_lnfold.md)
        if (t < BN / 2) {
          const int j = t < BN / 4 ? t : t - BN / 4;        // 16-byte group of s (first BN / 4 threads) or c
          const int n = min(n0 + 4 * j, p.N - 4);
          lnsc_v = *(const float4*)((t < BN / 4 ? p.ln_s : p.ln_c) + n);
        }
      }
    }
    if constexpr (LN_PF) {
      if (p.ln_stats) {
#pragma unroll
        for (int ih = 0; ih < MH; ++ih) ln_load(ih, lnq[ih]);
      }
    }
    const uint16_t* __restrict__ resid = (const uint16_t*)p.residual;
    const uint16_t* __restrict__ bias = (const uint16_t*)p.bias;
    const uint16_t* __restrict__ rowvec = (const uint16_t*)p.rowvec;
    if constexpr (PFN) {
      if (resid) {
#pragma unroll
        for (int ih = 0; ih < MH; ++ih) {
          const int m = min(row_of(ih), p.M - 1);
#pragma unroll
          for (int jh = 0; jh < NH; ++jh) res_v[ih][jh] = *(const uint2*)(resid + (size_t)m * p.ldr + min(col_of(jh), p.N - 4));
        }
      }
    }
    if constexpr (PF && STG) {
      if (resid && wide) {
#pragma unroll
        for (int ih = 0; ih < MH; ++ih)
#pragma unroll
          for (int t = 0; t < TT; ++t) {
            // every lane loads (a piece index past the band is clamped to the last piece; its value is never used): a load under a
            // lane predicate made the compiler park the wave on vmcnt(0) in the prologue -- behind pair 0's LDS-DMA -- before it
            // could reuse the address registers (profiles/r04i_gemm_stage_trace_lnfold.md, "ring issue" + 1 k cycles)
            const int q = (NPIECE % 64 == 0) ? lane + 64 * t : min(lane + 64 * t, NPIECE - 1);
            const int m = min(band_row0(ih) + q / PPR, p.M - 1), n = min(band_col0 + (q % PPR) * 8, p.N - 8);
            resw[ih][t] = *(const uint4*)(resid + (size_t)m * p.ldr + n);
          }
      }
    }
    if (!STREAMW && bias) {
#pragma unroll
      for (int jh = 0; jh < NH; ++jh) bias_v[jh] = *(const uint2*)(bias + min(col_of(jh), p.N - 4));
    }
    if (PF && rowvec) {
#pragma unroll
      for (int ih = 0; ih < (PF ? MH : 0); ++ih) {
        const int m = min(row_of(ih), p.M - 1);
        const size_t ro = (size_t)(m / p.rows_per_batch) * p.ld_rowvec;
#pragma unroll
        for (int jh = 0; jh < NH; ++jh) rowvec_v[ih][jh] = *(const uint2*)(rowvec + ro + min(col_of(jh), p.N - 4));
      }
    }
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: row r16 of a 16-row tile, 16-byte chunk (kq + 4 ks) ^ ((r16 >> 1) & 7); k-step 1 = chunk ^ 4
  const int fsw = (r16 >> 1) & 7;
  const int foff0 = r16 * 128 + ((kq ^ fsw) << 4), foff1 = r16 * 128 + (((kq ^ fsw) ^ 4) << 4);
  const int xfrag = g * SLICE + (wm * MT) * 2048, wfrag = g * SLICE + XBYTES + (GIL ? wn : wn * NT) * 2048;
  auto wtile_off = [](int j) { return (GIL ? 4 * (j >> 1) + 2 * (j & 1) : j) * 2048; };   // LDS offset of the wave's j-th W tile
  bf16x8_t xf0[MT], wf0[NT], xf1[MT], wf1[NT];
#define DA2_FRAG(XF, WF, SLOT, OFF)                                                                         \
  do {                                                                                                      \
    const unsigned char* xb_ = smem + (SLOT) * PAIR + xfrag + (OFF);                                        \
    const unsigned char* wb_ = smem + (SLOT) * PAIR + wfrag + (OFF);                                        \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) WF[j] = *(const bf16x8_t*)(wb_ + wtile_off(j));          \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) XF[i] = *(const bf16x8_t*)(xb_ + i * 2048);              \
  } while (0)
#define DA2_MFMA(XF, WF)                                                                                    \
  do {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                          \
    _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                          \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[j], XF[i], acc[i][j], 0, 0, 0);                \
  } while (0)
#define DA2_SG_DS(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define DA2_SG_MF(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define DA2_SG_VM(n) __builtin_amdgcn_sched_group_barrier(0x020, n, 0)

  // ---- prologue: pairs 0 .. NSLOT-1 in flight (zeros past the end of K), pair 0 landed, its k-step-0 fragments on their way ----
  DA2_TRACE(1);                                           // set-up done, first LDS-DMA about to issue
  // (streaming-W loop: slice 0 only -- every iteration of that loop, the first included, sends the following slice)
#pragma unroll
  for (int s = 0; s < (STREAMW ? 1 : NSLOT); ++s) {
    stage_issue(s);
    stage_advance();
    if (s == 0) prefetch_epilogue_operands();
  }
  DA2_TRACE(2);                                           // ring issued
  DA2_WAIT_PAIRS(STREAMW ? 0 : NSLOT - 1);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  DA2_TRACE(3);                                           // first pair landed for everybody
  // LayerNorm fold, consumer side: mean / rstd of the MH rows this lane finishes.  Each kq lane adds its six slots in slot order,
  // the four lanes of a row combine as (a + b) + (c + d): the same bits in all four.
  float ln_mu[MH], ln_rs[MH];
#pragma unroll
  for (int ih = 0; ih < MH; ++ih) ln_mu[ih] = 0.f, ln_rs[ih] = 1.f;
  const bool ln_on = LNF && p.ln_stats != nullptr;
  if constexpr (LNF) {
    if (ln_on) {
      if (t < BN / 2) {
        float* lnsc = (float*)(smem + NSLOT * PAIR + DUMP);
        const int j = t < BN / 4 ? t : t - BN / 4;
        *(float4*)(lnsc + (t < BN / 4 ? 0 : BN) + 4 * j) = lnsc_v;
      }
      const float inv_c = 1.0f / (float)p.K;              // the normalised dimension is this GEMM's K
#pragma unroll
      for (int ih = 0; ih < MH; ++ih) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int q = 6 * kq + 2 * u;
          s1 += (q < p.ln_parts ? lnq[ih][u].x : 0.f) + (q + 1 < p.ln_parts ? lnq[ih][u].z : 0.f);
          s2 += (q < p.ln_parts ? lnq[ih][u].y : 0.f) + (q + 1 < p.ln_parts ? lnq[ih][u].w : 0.f);
        }
        s1 += __shfl_xor(s1, 16, 64);
        s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const float mean = s1 * inv_c;
        ln_mu[ih] = mean;
        ln_rs[ih] = rsqrtf(fmaxf(s2 * inv_c - mean * mean, 0.f) + p.ln_eps);
      }
    }
  }
  if constexpr (STREAMW) {
    // ---- streaming-W loop: per slice 2 k-steps x NT W fragments, each read once and used by MT MFMAs ----
    constexpr int QD = 2;                                 // W fragments in flight ahead of the MFMAs that use them
    constexpr int NG = 2 * NT;                            // (k-step, W tile) groups per slice
    static_assert(XI + WI <= NG, "one LDS-DMA per MFMA group at most");
    const unsigned char* xbase_ = smem + xfrag, * wbase_ = smem + wfrag;
    auto wfrag_at = [&](const unsigned char* wb, int gi) {   // group gi = ks * NT + j
      return *(const bf16x8_t*)(wb + wtile_off(gi % NT) + (gi < NT ? foff0 : foff1));
    };
    // The two waves that share a SIMD (w and w + 4) run this body with their LDS-DMA issue in DIFFERENT thirds of the slice
    // (DOFF): an LDS-DMA instruction holds its wave's issue slot for ~100 cycles, and with both waves of a SIMD stalled at the
    // same time the matrix pipe idled for a third of every slice (4.06 k cycles per slice against 2.56 k of MFMA work).
    auto slice_body = [&](auto doff_c, int slot) __attribute__((always_inline)) {
      constexpr int DOFF = decltype(doff_c)::value;
      static_assert(DOFF + XI + WI <= NG, "LDS-DMA window past the end of the slice");
      const unsigned char* xb = xbase_ + slot * PAIR;
      const unsigned char* wb = wbase_ + slot * PAIR;
      bf16x8_t xs[2][MT], wq[QD + 1];
#pragma unroll
      for (int i = 0; i < MT; ++i) xs[0][i] = *(const bf16x8_t*)(xb + i * 2048 + foff0);
#pragma unroll
      for (int q = 0; q < QD; ++q) wq[q] = wfrag_at(wb, q);
#pragma unroll
      for (int i = 0; i < MT; ++i) xs[1][i] = *(const bf16x8_t*)(xb + i * 2048 + foff1);
      // Source order IS the schedule: a sched_barrier after every group.  (Group pins only fix the instruction KINDS: the
      // scheduler then filled each "one ds_read" slot with the read needed soonest and the queue depth collapsed to zero.)
      __builtin_amdgcn_sched_barrier(0);
      unsigned char* nbase = smem + (slot ^ 1) * PAIR;    // slice sl + 1 (zeros past the end of K)
      static_for<NG>([&](auto gc) {
        constexpr int gi = decltype(gc)::value, ks = gi / NT, j = gi % NT;
        if constexpr (gi + QD < NG) wq[(gi + QD) % (QD + 1)] = wfrag_at(wb, gi + QD);
        if constexpr (gi >= DOFF && gi < DOFF + XI + WI) stage_issue_one(nbase, std::integral_constant<int, gi - DOFF>{});
#pragma unroll
        for (int i = 0; i < MT; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[gi % (QD + 1)], xs[ks][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    // One complete loop per wave group (the group is fixed for the life of the wave: one branch, not one per slice).
    auto slice_loop = [&](auto doff_c) __attribute__((always_inline)) {
      int slot = 0;
      for (int sl = 0; sl < nk2; ++sl) {
        // slice sl has landed and is visible (prologue / the rendezvous at the end of the previous iteration); the other ring
        // slot is free: every wave has retired its reads of slice sl - 1.  Slice sl + 1 goes out now, one LDS-DMA per MFMA
        // group inside this wave's window, weights (the HBM-cold operand) first.
        slice_body(doff_c, slot);
        __builtin_amdgcn_sched_barrier(0);
        stage_advance();
        if (sl + 1 < nk2) {
          // rendezvous: my reads of this slice retired, my share of slice sl + 1 landed (the only LDS-DMA in flight): past the
          // barrier slice sl + 1 is visible and this slice's slot may be overwritten
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          DA2_WAIT_PAIRS(0);
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
        slot ^= 1;
      }
    };
    if (wave < 4) slice_loop(std::integral_constant<int, 0>{});
    else slice_loop(std::integral_constant<int, NG - (XI + WI) - 3>{});
  } else {
  DA2_FRAG(xf0, wf0, 0, foff0);

  auto first_half_pins = [&]() {
    static_for<MT + NT>([](auto rc) {                     // k-step 1's MT + NT reads spread under k-step 0's MT * NT MFMAs
      constexpr int r = decltype(rc)::value, R = MT + NT, Q = MT * NT;
      DA2_SG_DS(1);
      DA2_SG_MF((Q * (r + 1)) / R - (Q * r) / R);
    });
  };
  // PP: group 1 enters the loop one barrier late (it pairs with group 0's first rendezvous) and stays half an iteration behind:
  // group 0's rendezvous is group 1's separator and vice versa; group 0 pays the barrier back after its last pair.
  if constexpr (PP) {
    if (g == 1) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
  int slot = 0;                                           // ring slot of pair pr
  for (int pr = 0; pr + 1 < nk2; ++pr) {
    // first half: k-step 0 under the ds_reads of k-step 1
    DA2_FRAG(xf1, wf1, slot, foff1);
    DA2_MFMA(xf0, wf0);
    first_half_pins();
    __builtin_amdgcn_sched_barrier(0);
    // rendezvous: every fragment of this pair is in registers (lgkmcnt(0)) and the next pair has landed for this wave;
    // past the barrier this pair's ring slot is free (nobody reads it any more) and the next pair is visible to everybody
    const int nslot = (slot + 1 == NSLOT) ? 0 : slot + 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DA2_WAIT_PAIRS(NSLOT - 2);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // second half: k-step 1 under the LDS-DMA issue of pair pr + NSLOT and the next pair's k-step-0 reads
    stage_issue(slot);
    DA2_FRAG(xf0, wf0, nslot, foff0);
    DA2_MFMA(xf1, wf1);
    static_for<XI + WI + MT + NT>([](auto rc) {
      // one memory operation per share of the MT * NT MFMAs: the LDS-DMA issues, then the fragment reads
      constexpr int r = decltype(rc)::value, V = XI + WI, R = V + MT + NT, Q = MT * NT;
      if constexpr (r < V) DA2_SG_VM(1);
      else DA2_SG_DS(1);
      DA2_SG_MF((Q * (r + 1)) / R - (Q * r) / R);
    });
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PP) {                                   // separator: the other group's rendezvous
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    stage_advance();
    slot = nslot;
  }
  // last pair: nothing left to wait for or to fetch
  DA2_FRAG(xf1, wf1, slot, foff1);
  DA2_MFMA(xf0, wf0);
  first_half_pins();
  __builtin_amdgcn_sched_barrier(0);
  DA2_MFMA(xf1, wf1);
  if constexpr (PP) {
    if (g == 0) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                       // pairs with group 1's last separator
      asm volatile("" ::: "memory");
    }
  }
  }
#undef DA2_FRAG
#undef DA2_MFMA

  DA2_TRACE(4);                                           // K loop done
  // ---- the two K-groups meet: each sends the half it does not finish, through the (now free) ring ----
  f32x4_t keep[MH][NH];
  if constexpr (KG == 1) {
#pragma unroll
    for (int ih = 0; ih < MH; ++ih)
#pragma unroll
      for (int jh = 0; jh < NH; ++jh) keep[ih][jh] = acc[ih][jh];
  } else {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (XA > 0) {
      // The ring is free of readers: K ([16 XA keys][64 d], 128-byte rows) and V^T ([64 d][128 keys], 256-byte rows) of the tile's
      // two heads go to its upper half by LDS-DMA while the partial sums cross the lower half.  16-byte slots XOR-swizzled on the
      // SOURCE side (the LDS image of a DMA is lane-linear): K slot ^ ((row >> 1) & 7), V^T slot ^ (row & 15) -- the 8-byte
      // fragment reads of the attention part are then conflict-free.  Keys past xa_skv_alloc arrive as hardware zeros (bit 31).
      if (p.xa_k) {
        constexpr int KP = 2 * XA, NPK = 2 * KP, NPV = 32, NP = NPK + NPV;
        const int hb = n0 >> 6, bq = m0 / p.rows_per_batch;
        __amdgpu_buffer_rsrc_t rs_k = uniform_rsrc(p.xa_k, 0x7fffffff), rs_v = uniform_rsrc(p.xa_vt, 0x7fffffff);
#pragma unroll
        for (int i = 0; i < (NP + 7) / 8; ++i) {
          const int q = i * 8 + wave;                     // wave-uniform piece index
          if (q < NPK) {
            const int hh = q >= KP ? 1 : 0, r8 = q - hh * KP;
            const int row = r8 * 8 + (lane >> 3);
            const int sl = (lane & 7) ^ ((row >> 1) & 7);
            const int off = ((bq * p.xa_skv_alloc + row) * p.xa_k_ld + (hb + hh) * 64 + sl * 8) * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, DA2_LDS(smem + XA_K_OFF + hh * XA_KB + r8 * 1024), 16,
                                                     off | (row < p.xa_skv_alloc ? 0 : (int)0x80000000), 0, 0, 0);
          } else if (q < NP) {
            const int qv = q - NPK, hh = qv >> 4, r4 = qv & 15;
            const int row = r4 * 4 + (lane >> 4);
            const int sl = (lane & 15) ^ (row & 15);
            const int off = (((hb + hh) * 64 + row) * (int)p.xa_vt_ld + bq * p.xa_skv_alloc + sl * 8) * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, DA2_LDS(smem + XA_V_OFF + hh * 16384 + r4 * 1024), 16,
                                                     off | (sl * 8 < p.xa_skv_alloc ? 0 : (int)0x80000000), 0, 0, 0);
          }
        }
      }
    }
    f32x4_t* xch = (f32x4_t*)smem + (size_t)wq * (MT * NT) * 64 + lane;
#pragma unroll
    for (int ih = 0; ih < MH; ++ih)
#pragma unroll
      for (int jh = 0; jh < NH; ++jh) {
        // lower / upper half of the wave tile: acc[ih][jh] and acc[ih + MH][jh] (column halves when MT is odd)
        const f32x4_t lo = acc[ih][jh];
        f32x4_t hi;
        if constexpr (SPLIT_M) hi = acc[ih + MT / 2][jh];
        else hi = acc[ih][jh + NT / 2];
        f32x4_t snd, kp;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          snd[e] = g ? lo[e] : hi[e];
          kp[e] = g ? hi[e] : lo[e];
        }
        keep[ih][jh] = kp;
        xch[(size_t)((ih * NH + jh) + (g ? 0 : MH * NH)) * 64] = snd;   // group 0 sends the upper half, group 1 the lower
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int ih = 0; ih < MH; ++ih)
#pragma unroll
      for (int jh = 0; jh < NH; ++jh) {
        const f32x4_t o = xch[(size_t)((ih * NH + jh) + (g ? MH * NH : 0)) * 64];   // what the partner sent for MY half
#pragma unroll
        for (int e = 0; e < 4; ++e) keep[ih][jh][e] += o[e];                        // (even slices) + (odd slices), commutative
      }
  }

  if constexpr (STREAMW) {                                // (no registers to carry the bias through this loop: fetched here)
    if (p.bias) {
#pragma unroll
      for (int jh = 0; jh < NH; ++jh) bias_v[jh] = *(const uint2*)((const uint16_t*)p.bias + min(col_of(jh), p.N - 4));
    }
  }
  // ---- optional: pull a later launch's weight towards the memory-side cache (da_gemm_params.prefetch).  Each wave sends up to
  // eight 1 KiB LDS-DMA reads of this workgroup's share into the scratch KiB; nothing waits for them (a kernel that loads its
  // residual in the epilogue would queue behind them: it skips the prefetch) ----
  auto weight_prefetch = [&]() {
  if (p.prefetch && (PF || !p.residual)) {
    const int nchunk = (int)min((long long)0x7fffffff >> 10, p.prefetch_bytes >> 10);
    __amdgpu_buffer_rsrc_t rs_pf = uniform_rsrc(p.prefetch, (size_t)nchunk << 10);
    const int stride = (int)gridDim.x * 8;
    int c = bid * 8 + wave;
    // up to 16 KiB per wave (round 4: 8 covered 16.8 MB per 256-workgroup launch -- 64 % of the 26 MB GEGLU projection weight)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (c < nchunk) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_pf, DA2_LDS(smem + NSLOT * PAIR), 16, lane * 16, c << 10, 0, 0);
      c += stride;
    }
  }
  };
  const bool xa_on = XA > 0 && p.xa_k != nullptr;
  if (!xa_on) weight_prefetch();                          // (XA: behind the attention part -- its K / V^T wait is a vmcnt(0))
  DA2_TRACE(5);                                           // partial sums exchanged
  // o[0..3] = alpha * acc of row tile ih, columns n .. n + 3  ->  LayerNorm-folded value (identity without ln_stats)
  auto ln_apply4 = [&](float* o, int ih, int n) __attribute__((always_inline)) {
    if constexpr (LNF) {
      if (ln_on) {
        const float* lnsc = (const float*)(smem + NSLOT * PAIR + DUMP);
        const int j = min(n - n0, BN - 4);                 // (n is clamped to N - 4 by the callers: stay inside the block's slice)
        const float4 sv = *(const float4*)(lnsc + j), cv = *(const float4*)(lnsc + BN + j);
        o[0] = ln_rs[ih] * (o[0] - ln_mu[ih] * sv.x) + cv.x;
        o[1] = ln_rs[ih] * (o[1] - ln_mu[ih] * sv.y) + cv.y;
        o[2] = ln_rs[ih] * (o[2] - ln_mu[ih] * sv.z) + cv.z;
        o[3] = ln_rs[ih] * (o[3] - ln_mu[ih] * sv.w) + cv.w;
      }
    }
  };
  // ---- XA: cross-attention on the wave's own q (32 queries x one head) ----
  // keep[ih][jh][e] = q^T[d = 16 jh + 4 kq + e][query 16 ih + r16].  With the MFMA k index u (0 .. 7) of k-block kq mapped to
  // d = 32 b + 16 (u / 4) + 4 kq + u % 4, the lane's eight values of keep[ih][2 b .. 2 b + 1] ARE the B fragment of S^T = K . Q^T for
  // d-block b, and K's A fragment is two 8-byte reads of a key row; the C layout of S^T (keys 16 kt + 4 kq + e of query r16) is the
  // B fragment of O^T = V^T . P^T under the same mapping applied to keys, and O^T comes out in the layout of `keep`: the ordinary
  // store path finishes the launch.  Softmax over the <= 16 XA keys is exact (no running maximum): exp2 of scale * log2(e) * s.
  if constexpr (XA > 0) {
    if (xa_on) {
      bf16x8_t qf[MH][2];
#pragma unroll
      for (int ih = 0; ih < MH; ++ih)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float o8[8];
#pragma unroll
          for (int hlf = 0; hlf < 2; ++hlf) {
            const int jh = 2 * b + hlf;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = keep[ih][jh][e] * p.alpha;
            ln_apply4(o, ih, min(col_of(jh), p.N - 4));
            const uint2 bv = bias_v[jh];
            o8[4 * hlf + 0] = o[0] + bf_lo(bv.x); o8[4 * hlf + 1] = o[1] + bf_hi(bv.x);
            o8[4 * hlf + 2] = o[2] + bf_lo(bv.y); o8[4 * hlf + 3] = o[3] + bf_hi(bv.y);
          }
          const uint4 pk = pack8(o8);                      // q rounded to bf16, as the reference's to_q output is
          qf[ih][b] = __builtin_bit_cast(bf16x8_t, pk);
        }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // my share of K / V^T landed; my exchange reads retired
      __builtin_amdgcn_s_barrier();                       // ... everybody's: K / V^T visible, the lower half of the ring is free
      asm volatile("" ::: "memory");
      const unsigned char* kb = smem + XA_K_OFF + wn * XA_KB;
      const unsigned char* vb = smem + XA_V_OFF + wn * 16384;
      f32x4_t sacc[XA][MH];
#pragma unroll
      for (int kt = 0; kt < XA; ++kt) {
#pragma unroll
        for (int ih = 0; ih < MH; ++ih) sacc[kt][ih] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int R = 16 * kt + r16, sw = (R >> 1) & 7;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const uint2 lo = *(const uint2*)(kb + R * 128 + (((4 * b + (kq >> 1)) ^ sw) << 4) + (kq & 1) * 8);
          const uint2 hi = *(const uint2*)(kb + R * 128 + (((4 * b + 2 + (kq >> 1)) ^ sw) << 4) + (kq & 1) * 8);
          const bf16x8_t kf = __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
          for (int ih = 0; ih < MH; ++ih) sacc[kt][ih] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ih][b], sacc[kt][ih], 0, 0, 0);
        }
      }
      constexpr int KBL = (XA + 1) / 2;                   // 32-key blocks of the P . V product
      const float sl2 = p.xa_scale * 1.4426950408889634f;
      bf16x8_t pf[MH][KBL];
      float linv[MH];
#pragma unroll
      for (int ih = 0; ih < MH; ++ih) {
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < XA; ++kt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = (16 * kt + 4 * kq + e < p.xa_skv) ? sacc[kt][ih][e] * sl2 : -3.0e38f;
            sacc[kt][ih][e] = v;
            mx = fmaxf(mx, v);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < KBL; ++c) {
          float p8[8];
#pragma unroll
          for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float pv = 0.f;
              if (2 * c + hlf < XA) pv = __builtin_amdgcn_exp2f(sacc[2 * c + hlf < XA ? 2 * c + hlf : 0][ih][e] - mx);   // masked keys: exp2(-huge) = 0
              p8[4 * hlf + e] = pv;
              sum += pv;
            }
          pf[ih][c] = __builtin_bit_cast(bf16x8_t, pack8(p8));
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        linv[ih] = 1.0f / sum;
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        f32x4_t oacc[MH];
#pragma unroll
        for (int ih = 0; ih < MH; ++ih) oacc[ih] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int Rd = 16 * dt + r16, sw = Rd & 15;
#pragma unroll
        for (int c = 0; c < KBL; ++c) {
          const uint2 lo = *(const uint2*)(vb + Rd * 256 + (((4 * c + (kq >> 1)) ^ sw) << 4) + (kq & 1) * 8);
          const uint2 hi = *(const uint2*)(vb + Rd * 256 + (((4 * c + 2 + (kq >> 1)) ^ sw) << 4) + (kq & 1) * 8);
          const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
          for (int ih = 0; ih < MH; ++ih) oacc[ih] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[ih][c], oacc[ih], 0, 0, 0);
        }
#pragma unroll
        for (int ih = 0; ih < MH; ++ih)
#pragma unroll
          for (int e = 0; e < 4; ++e) keep[ih][dt][e] = oacc[ih][e] * linv[ih];
      }
      weight_prefetch();
    }
  }
  // ---- epilogue: lane holds, for output row (r16 of a 16-row tile), channels 4 kq .. 4 kq + 3 of a 16-column tile ----
  const uint16_t* __restrict__ bias_rows = (const uint16_t*)p.bias_rows;
  const bool has_rowvec = p.rowvec != nullptr, has_res = p.residual != nullptr;
#if defined(DA_GEMM2_TRACE)
#define DA2_TRACE_END()                                                                                     \
  do {                                                                                                      \
    DA2_TRACE(6);                                         /* output stores issued */                        \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
    DA2_TRACE(7);                                         /* ... and acknowledged */                        \
    if (lane == 0) {                                                                                        \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) g_da2_trace[((size_t)blockIdx.x * 8 + wave) * 8 + i] = tr_[i]; \
    }                                                                                                       \
  } while (0)
#else
#define DA2_TRACE_END() ((void)0)
#endif
  if (geglu) {
    if constexpr (GIL) {
      // wave tiles (2u, 2u + 1) = block tiles (4u + wn, 4u + wn + 2) = a value tile and its gate tile.  The finished bf16 values
      // of the whole block tile (BM x BN / 2) meet in LDS and leave as whole rows, 16 bytes per lane.
      constexpr int OROW = BN + 16;                       // bytes per staged output row (BN / 2 bf16 + pad)
      static_assert(BM * OROW <= NSLOT * PAIR, "output tile does not fit the ring");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // every wave is out of the ring
      asm volatile("" ::: "memory");
      auto body = [&](auto tanh_c) __attribute__((always_inline)) {
        constexpr bool TANH = decltype(tanh_c)::value;
#pragma unroll
        for (int ih = 0; ih < MH; ++ih) {
          unsigned char* orow = smem + ((wm * MT + ih) * 16 + r16) * OROW + (wn * 16 + 4 * kq) * 2;
#pragma unroll
          for (int u = 0; u < NT / 2; ++u) {
            const uint2 bh = bias_v[2 * u], bg = bias_v[2 * u + 1];      // zeros without a bias
            float o[4], hq[4], gq[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) hq[e] = keep[ih][2 * u][e] * p.alpha, gq[e] = keep[ih][2 * u + 1][e] * p.alpha;
            ln_apply4(hq, ih, min(col_of(2 * u), p.N - 4));
            ln_apply4(gq, ih, min(col_of(2 * u + 1), p.N - 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float hv = hq[e], gv = gq[e];
              hv += (e == 0) ? bf_lo(bh.x) : (e == 1) ? bf_hi(bh.x) : (e == 2) ? bf_lo(bh.y) : bf_hi(bh.y);
              gv += (e == 0) ? bf_lo(bg.x) : (e == 1) ? bf_hi(bg.x) : (e == 2) ? bf_lo(bg.y) : bf_hi(bg.y);
              hv = bf2f(f2bf(hv));   // the reference rounds the projection to bf16 before chunk / gelu / mul
              gv = bf2f(f2bf(gv));
              o[e] = hv * bf2f(f2bf(TANH ? gelu_tanh_f(gv) : gelu_erf_f(gv)));
            }
            uint2 pk;
            pk.x = pack_bf2(o[0], o[1]);
            pk.y = pack_bf2(o[2], o[3]);
            *(uint2*)(orow + u * 64) = pk;
          }
        }
      };
      if (p.act == DA_ACT_GEGLU) body(std::false_type{});
      else body(std::true_type{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      constexpr int OPR = BN / 16, NOP = BM * OPR;        // 16-byte pieces per output row / per tile
      const int no0 = n0 >> 1, Nout = p.N >> 1;
#pragma unroll
      for (int k = 0; k < (NOP + 511) / 512; ++k) {
        const int q = t + 512 * k;
        if (NOP % 512 != 0 && q >= NOP) continue;
        const int row = q / OPR, c = q % OPR;
        if (m0 + row >= p.M || no0 + c * 8 >= Nout) continue;
        *(uint4*)((uint16_t*)p.C + (size_t)(m0 + row) * p.ldc + no0 + c * 8) = *(const uint4*)(smem + row * OROW + c * 16);
      }
      DA2_TRACE_END();
      return;
    }
    // packed weight rows: per 64 = [32 value | 32 gate]  ->  16-column tiles (4u, 4u+1) = value, (4u+2, 4u+3) = gate
    if constexpr (!GIL && SPLIT_M && (NT % 4) == 0) {
      auto body = [&](auto tanh_c) __attribute__((always_inline)) {
        constexpr bool TANH = decltype(tanh_c)::value;
#pragma unroll
        for (int ih = 0; ih < MH; ++ih) {
          const int m = row_of(ih);
          if (m >= p.M) continue;
#pragma unroll
          for (int u = 0; u < NT / 4; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
              const int jv = 4 * u + v, jg = jv + 2;
              const int nv = col_of(jv);
              if (nv >= p.N) continue;
              const int no = (n0 >> 1) + (wn * (NT / 4) + u) * 32 + v * 16 + 4 * kq;
              const uint2 bh = bias_v[jv], bg = bias_v[jg];          // zeros without a bias
              float o[4], hq[4], gq[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) hq[e] = keep[ih][jv][e] * p.alpha, gq[e] = keep[ih][jg][e] * p.alpha;
              ln_apply4(hq, ih, nv);
              ln_apply4(gq, ih, col_of(jg));
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float hv = hq[e], gv = gq[e];
                hv += (e == 0) ? bf_lo(bh.x) : (e == 1) ? bf_hi(bh.x) : (e == 2) ? bf_lo(bh.y) : bf_hi(bh.y);
                gv += (e == 0) ? bf_lo(bg.x) : (e == 1) ? bf_hi(bg.x) : (e == 2) ? bf_lo(bg.y) : bf_hi(bg.y);
                hv = bf2f(f2bf(hv));   // the reference rounds the projection to bf16 before chunk / gelu / mul
                gv = bf2f(f2bf(gv));
                o[e] = hv * bf2f(f2bf(TANH ? gelu_tanh_f(gv) : gelu_erf_f(gv)));
              }
              uint2 pk;
              pk.x = pack_bf2(o[0], o[1]);
              pk.y = pack_bf2(o[2], o[3]);
              *(uint2*)((uint16_t*)p.C + (size_t)m * p.ldc + no) = pk;
            }
        }
      };
      if (p.act == DA_ACT_GEGLU) body(std::false_type{});
      else body(std::true_type{});
    }
    return;
  }
  // Everything else: one straight-line instantiation per (activation, gate kind), chosen ONCE per launch.
  // PACKED = nothing happens behind the activation (no residual, out_scale == 1): the band goes through LDS already rounded to
  // bf16 and the way back is one 16-byte read + one 16-byte store per lane.
  auto wide_body = [&](auto act_c, auto gate_c, auto packed_c) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_c)::value, GATE = decltype(gate_c)::value;
    constexpr bool PACKED = decltype(packed_c)::value;
    if constexpr (STG) {
      if constexpr (KG == 1) {                          // (KG == 2: the exchange barriers already emptied the ring of readers)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      // (XA: the bands go to the LOWER half of the ring -- K / V^T sit where they normally go; the barrier in front of the attention
      // part has retired every wave's exchange reads)
      unsigned char* stg = smem + ((KG == 2 && !xa_on) ? 4 * MT * NT * 1024 : 0) + wave * (16 * ROWB);
      unsigned char* stg_w = stg + r16 * ROWB + kq * 16;      // this lane's slot in the MFMA layout
      const uint16_t* __restrict__ resid = (const uint16_t*)p.residual;
#pragma unroll
      for (int ih = 0; ih < MH; ++ih) {
        // finished-but-for-the-residual values of the band, fp32, in the MFMA layout -> LDS
        const int mc = min(row_of(ih), p.M - 1);
        const int bidx = (GATE != 0 || (!PF && has_rowvec)) ? (mc / p.rows_per_batch) : 0;
        const float brow = bias_rows ? bf2f(bias_rows[mc]) : 0.f;
#pragma unroll
        for (int jh = 0; jh < NH; ++jh) {
          const int n = min(col_of(jh), p.N - 4);
          float o[4];
          if (XA > 0 && xa_on) {                          // keep = the attention output: alpha / fold / bias went into q
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = keep[ih][jh][e];
          } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = keep[ih][jh][e] * p.alpha;
          ln_apply4(o, ih, n);
          uint2 rvv = make_uint2(0, 0);
          if constexpr (PF) rvv = rowvec_v[ih][jh];
          else if (has_rowvec) rvv = *(const uint2*)((const uint16_t*)p.rowvec + (size_t)bidx * p.ld_rowvec + n);
          epilogue4<ACT, GATE, false>(p, o, n, bidx, brow, bias_v[jh], rvv, make_uint2(0, 0));
          }
          if constexpr (PACKED) {
            uint2 pk;
            pk.x = pack_bf2(o[0], o[1]);
            pk.y = pack_bf2(o[2], o[3]);
            *(uint2*)(stg + r16 * (ROWB / 2) + kq * 8 + jh * 32) = pk;
          } else {
            *(float4*)(stg_w + jh * 64) = make_float4(o[0], o[1], o[2], o[3]);
          }
        }
        // ... and back, 8 consecutive channels of one row per lane: + residual, * out_scale, one 16-byte store
        // (LNF producer: the lane's share of its row's statistics, from the ROUNDED values, goes to the wave's scratch slots)
        float2* part = (float2*)(smem + STATS_OFF) + wave * (TT * 64);
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          const int q = lane + 64 * t;
          if (NPIECE % 64 != 0 && q >= NPIECE) continue;
          const int row = q / PPR, c8 = q % PPR;
          const int mo = band_row0(ih) + row, no = band_col0 + c8 * 8;
          const bool inside = mo < p.M && no < p.N;
          uint4 pk;
          if constexpr (PACKED) {
            pk = *(const uint4*)(stg + row * (ROWB / 2) + c8 * 16);
          } else {
            const float4 lo = *(const float4*)(stg + row * ROWB + c8 * 32), hi = *(const float4*)(stg + row * ROWB + c8 * 32 + 16);
            float o[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            uint4 rw = make_uint4(0, 0, 0, 0);
            if constexpr (PF) rw = resw[ih][t];
            else if (has_res && inside) rw = *(const uint4*)(resid + (size_t)mo * p.ldr + no);
            o[0] += bf_lo(rw.x); o[1] += bf_hi(rw.x); o[2] += bf_lo(rw.y); o[3] += bf_hi(rw.y);
            o[4] += bf_lo(rw.z); o[5] += bf_hi(rw.z); o[6] += bf_lo(rw.w); o[7] += bf_hi(rw.w);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] *= p.out_scale;
            pk.x = pack_bf2(o[0], o[1]); pk.y = pack_bf2(o[2], o[3]); pk.z = pack_bf2(o[4], o[5]); pk.w = pack_bf2(o[6], o[7]);
          }
          if (inside) *(uint4*)((uint16_t*)p.C + (size_t)mo * p.ldc + no) = pk;
          if constexpr (LNF) {
            if (p.stats_out) {
              float r[8];
              unpack8(pk, r);
              const float s1 = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
              const float s2 = ((r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3])) +
                               ((r[4] * r[4] + r[5] * r[5]) + (r[6] * r[6] + r[7] * r[7]));
              part[q] = inside ? make_float2(s1, s2) : make_float2(0.f, 0.f);
            }
          }
        }
        if constexpr (LNF) {
          if (p.stats_out) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // wave-private slots: no barrier
            if (lane < 16) {
              float a = 0.f, b = 0.f;
#pragma unroll
              for (int c = 0; c < PPR; ++c) {                       // piece order: a fixed summation order
                const float2 v = part[lane * PPR + c];
                a += v.x;
                b += v.y;
              }
              const int mo = band_row0(ih) + lane;
              if (mo < p.M && band_col0 < p.N)
                *(float2*)(p.stats_out + (size_t)mo * p.stats_ld + 2 * (band_col0 / (16 * NH))) = make_float2(a, b);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the slots are rewritten by the next row tile
          }
        }
      }
    }
  };
  auto narrow_body = [&](auto act_c, auto gate_c) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_c)::value, GATE = decltype(gate_c)::value;
#pragma unroll
    for (int ih = 0; ih < MH; ++ih) {
      const int m = row_of(ih);
      if (m >= p.M) continue;
      const int bidx = (GATE != 0 || (!PF && has_rowvec)) ? (m / p.rows_per_batch) : 0;
      const float brow = bias_rows ? bf2f(bias_rows[m]) : 0.f;
#pragma unroll
      for (int jh = 0; jh < NH; ++jh) {
        const int n = col_of(jh);
        if (n >= p.N) continue;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = keep[ih][jh][e] * p.alpha;
        ln_apply4(o, ih, n);
        uint2 rvv = make_uint2(0, 0), rsv = make_uint2(0, 0);
        if constexpr (PF) rvv = rowvec_v[ih][jh];
        else if (has_rowvec) rvv = *(const uint2*)((const uint16_t*)p.rowvec + (size_t)bidx * p.ld_rowvec + n);
        if constexpr (PFN) rsv = res_v[ih][jh];
        else if (has_res) rsv = *(const uint2*)((const uint16_t*)p.residual + (size_t)m * p.ldr + n);
        epilogue4<ACT, GATE, true>(p, o, n, bidx, brow, bias_v[jh], rvv, rsv);
        if (p.out_f32) {
          *(float4*)((float*)p.C + (size_t)m * p.ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
          uint2 pk;
          pk.x = pack_bf2(o[0], o[1]);
          pk.y = pack_bf2(o[2], o[3]);
          *(uint2*)((uint16_t*)p.C + (size_t)m * p.ldc + n) = pk;
        }
      }
    }
  };
  // LNF instantiations: a tile that lies in the transposed column block (da_gemm_params.vt: the V columns of a fused Q | K | V
  // projection) writes vt[(n - vt_col0)][m] straight from the MFMA layout -- the 16 lanes of a kq group hold 16 consecutive tokens of
  // one channel: 32 contiguous bytes per (channel, kq) -- with the alpha / LayerNorm-fold / bias part of the epilogue only.
  if constexpr (LNF && !GIL) {
    if (p.vt && n0 >= p.vt_col0) {
      uint16_t* __restrict__ vt = (uint16_t*)p.vt;
#pragma unroll
      for (int ih = 0; ih < MH; ++ih) {
        const int m = row_of(ih);
#pragma unroll
        for (int jh = 0; jh < NH; ++jh) {
          const int n = col_of(jh);
          if (n >= p.N) continue;
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = keep[ih][jh][e] * p.alpha;
          ln_apply4(o, ih, n);
          const uint2 bv = bias_v[jh];
          o[0] += bf_lo(bv.x); o[1] += bf_hi(bv.x); o[2] += bf_lo(bv.y); o[3] += bf_hi(bv.y);
          if (m < p.M) {
            uint16_t* dst = vt + (size_t)(n - p.vt_col0) * p.ld_vt + m;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(size_t)e * p.ld_vt] = f2bf(o[e]);
          }
        }
      }
      DA2_TRACE_END();
      return;
    }
  }
  {
    using std::integral_constant;
    auto by_act = [&](auto&& f, auto... tail) __attribute__((always_inline)) {
      if (p.act == DA_ACT_NONE) f(integral_constant<int, DA_ACT_NONE>{}, tail...);
      else if (p.act == DA_ACT_GELU_TANH) f(integral_constant<int, DA_ACT_GELU_TANH>{}, tail...);
      else if (p.act == DA_ACT_SILU) f(integral_constant<int, DA_ACT_SILU>{}, tail...);
      else if (p.act == DA_ACT_GELU_ERF) f(integral_constant<int, DA_ACT_GELU_ERF>{}, tail...);
      else f(integral_constant<int, DA_ACT_QUICK_GELU>{}, tail...);
    };
    // gated launches (adaLN gates of the DiT blocks: activation NONE in practice) keep the run-time activation switch; where
    // the row-contiguous path exists the 8-byte path is only the unaligned-rows fallback and gets ONE run-time-switched copy
    constexpr integral_constant<int, -1> RT{};
    constexpr integral_constant<int, 0> G0{};
    if (STG && wide) {
      const bool packed = !has_res && p.out_scale == 1.0f;
      if (p.gate && p.gate_f32) wide_body(RT, integral_constant<int, 2>{}, std::false_type{});
      else if (p.gate) wide_body(RT, integral_constant<int, 1>{}, std::false_type{});
      else if (packed) by_act(wide_body, G0, std::true_type{});
      else by_act(wide_body, G0, std::false_type{});
    } else {
      if (p.gate && p.gate_f32) narrow_body(RT, integral_constant<int, 2>{});
      else if (p.gate) narrow_body(RT, integral_constant<int, 1>{});
      else if constexpr (STG) narrow_body(RT, G0);
      else by_act(narrow_body, G0);
    }
  }
  DA2_TRACE_END();
#undef DA2_TRACE_END
#undef DA2_LDS
#undef DA2_WAIT_PAIRS
#undef DA2_SG_DS
#undef DA2_SG_MF
#undef DA2_SG_VM
#endif  // __HIP_DEVICE_COMPILE__
}

// ---- host side ----


template <int KG, int WM, int WN, int MT, int NT, int NSLOT, bool CONV, bool PP = false, bool STREAMW = false, bool GIL = false,
          bool LNF = false, int XA = 0>
int launch(const da_gemm_params& p, hipStream_t s) {
  constexpr int BM = 16 * MT * WM, BN = 16 * NT * WN;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int gx = choose_xcd_gx2(tiles_m, tiles_n, BM, BN, CONV ? xcd_conv_weighting(p) : 1, CONV ? (long long)(p.C1 + p.C2) * 2 : 0), gy = 8 / gx;
  const int tm_per = (tiles_m + gy - 1) / gy, tn_per = (tiles_n + gx - 1) / gx;
  const int grid = 8 * tm_per * tn_per;
  const long long rows_xcd = (long long)BM * (tm_per < (32 + tn_per - 1) / tn_per ? tm_per : (32 + tn_per - 1) / tn_per);
  constexpr size_t lds = (size_t)NSLOT * KG * (BM + BN) * 128 + 1024 + (LNF ? 2 * BN * 4 : 0);   // ring + the scratch KiB (ragged pieces, prefetch) + s / c
  auto kern = igemm2_bf16_kernel<KG, WM, WN, MT, NT, NSLOT, CONV, PP, STREAMW, GIL, LNF, XA>;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return DA_ERR_LAUNCH;
    attr_set = true;
  }
  DA_LAUNCH(kern, dim3(grid), dim3(512), lds, s, p, gx | (CONV ? conv_chunk_slices(p, rows_xcd) << 8 : 0));
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// K2 tile codes -> instantiations.  staging: DA_STAGE_LDS_DIRECT = ring of 2 slice pairs, DA_STAGE_LDS_DIRECT3 = 3 (where it fits);
// DA_STAGE_PINGPONG / DA_STAGE_PINGPONG3 = the same rings with the two K-groups half an iteration apart (KG == 2 tiles only).
// Column band of one statistics partial of a K2 tile that can PRODUCE them (0: it cannot): 16 * NH columns.
inline int stats_band_cols(int tile) { return (tile == DA_TILE_K2_128x80 || tile == DA_TILE_K2_128x160) ? 80 : 0; }

// LayerNorm-fold instantiations (nn.Linear): the tiles the SDXL transformer blocks use -- 128 x 80 / 128 x 160 as producer
// (to_out, + residual) and consumer (to_q), the interleaved-ownership GEGLU tile (128 x 320) as consumer.
inline int dispatch_lnf(const da_gemm_params& p, int tile, int staging, hipStream_t s) {
  const bool geglu = (p.act == DA_ACT_GEGLU || p.act == DA_ACT_GEGLU_TANH);
  const int ns = (staging == DA_STAGE_LDS_DIRECT || staging == DA_STAGE_PINGPONG) ? 2
                 : (staging == DA_STAGE_LDS_DIRECT3 || staging == DA_STAGE_PINGPONG3) ? 3 : 0;
  const bool pp = staging == DA_STAGE_PINGPONG || staging == DA_STAGE_PINGPONG3;
  if (ns == 0) return DA_ERR_UNSUPPORTED;
  if (p.xa_k) {
    // cross-attention epilogue (XA): the 128 x 128 tile, heads of 64 channels, a row tile inside one batch, <= 80 text tokens,
    // whole 16-byte output rows, nothing behind q but the attention
    if (tile != DA_TILE_K2_128x128 || ns != 2 || geglu || !p.xa_vt || p.vt || p.stats_out || p.residual || p.gate || p.rowvec ||
        p.bias_rows || p.out_f32 || p.act != DA_ACT_NONE || p.out_scale != 1.0f || (p.N & 127) || (p.ldc & 7) ||
        ((size_t)p.C & 15) || p.rows_per_batch <= 0 || (p.rows_per_batch & 127) || (p.M % p.rows_per_batch) || p.xa_skv <= 0 ||
        p.xa_skv > p.xa_skv_alloc || p.xa_skv_alloc > 80 || (p.xa_skv_alloc & 7) || (p.xa_k_ld & 7) || (p.xa_vt_ld & 7) ||
        p.xa_k_ld < p.N || p.xa_vt_ld < (long long)(p.M / p.rows_per_batch) * p.xa_skv_alloc ||
        ((size_t)p.xa_k & 15) || ((size_t)p.xa_vt & 15) ||
        (size_t)(p.M / p.rows_per_batch) * p.xa_skv_alloc * p.xa_k_ld * 2 >= 0x7fffffffull || (size_t)p.N * p.xa_vt_ld * 2 >= 0x7fffffffull)
      return DA_ERR_UNSUPPORTED;
    return pp ? launch<2, 2, 2, 4, 4, 2, false, true, false, false, true, 5>(p, s)
              : launch<2, 2, 2, 4, 4, 2, false, false, false, false, true, 5>(p, s);
  }
  if (p.vt) {
    const int bn = tile == DA_TILE_K2_128x80 ? 80 : tile == DA_TILE_K2_128x160 ? 160 : tile == DA_TILE_K1_128x256 ? 256
                   : tile == DA_TILE_K1_256x128 ? 128 : 0;
    if (bn == 0 || geglu || p.stats_out || p.residual || p.gate || p.rowvec || p.bias_rows || p.out_f32 || p.act != DA_ACT_NONE ||
        p.vt_col0 <= 0 || p.vt_col0 >= p.N || (p.vt_col0 % bn) || (p.vt_col0 & 15) || p.ld_vt < p.M || p.out_scale != 1.0f)
      return DA_ERR_UNSUPPORTED;
  }
  if (p.stats_out) {
    // statistics come out of the row-contiguous store path only: 16-byte aligned bf16 rows (and residual rows)
    if (geglu || p.out_f32 || (p.ldc & 7) || ((size_t)p.C & 15) || (p.N & 7) || p.gate ||
        (p.residual && ((p.ldr & 7) || ((size_t)p.residual & 15))) || stats_band_cols(tile) == 0)
      return DA_ERR_UNSUPPORTED;
  }
  switch (tile) {
    case DA_TILE_K2_128x80:
      if (geglu) break;
      if (pp) return ns == 2 ? launch<2, 4, 1, 2, 5, 2, false, true, false, false, true>(p, s)
                             : launch<2, 4, 1, 2, 5, 3, false, true, false, false, true>(p, s);
      return ns == 2 ? launch<2, 4, 1, 2, 5, 2, false, false, false, false, true>(p, s)
                     : launch<2, 4, 1, 2, 5, 3, false, false, false, false, true>(p, s);
    case DA_TILE_K2_128x160:
      if (geglu || ns != 2) break;
      return pp ? launch<2, 2, 2, 4, 5, 2, false, true, false, false, true>(p, s)
                : launch<2, 2, 2, 4, 5, 2, false, false, false, false, true>(p, s);
    case DA_TILE_K1_128x320:
      if (!geglu || p.stats_out || ns != 2 || pp || (p.ldc & 7) || ((size_t)p.C & 15) || (p.N & 15)) break;
      return launch<1, 4, 2, 2, 10, 2, false, false, false, true, true>(p, s);
    // consumer-only instantiations of two eight-wave tiles: the fused Q | K | V projection (M 2048 x N 3840: 240 tiles of 128 x 256,
    // one round of the 256 CUs, where 128 x 160 needs 1.5 and 128 x 80 three)
    case DA_TILE_K1_128x256:
      if (geglu || p.stats_out || pp) break;
      return ns == 2 ? launch<1, 2, 4, 4, 4, 2, false, false, false, false, true>(p, s)
                     : launch<1, 2, 4, 4, 4, 3, false, false, false, false, true>(p, s);
    case DA_TILE_K1_256x128:
      if (geglu || p.stats_out || pp) break;
      return ns == 2 ? launch<1, 4, 2, 4, 4, 2, false, false, false, false, true>(p, s)
                     : launch<1, 4, 2, 4, 4, 3, false, false, false, false, true>(p, s);
  }
  return DA_ERR_UNSUPPORTED;
}

template <bool CONV>
int dispatch(const da_gemm_params& p, int tile, int staging, hipStream_t s) {
  if (p.split_k > 1 || !staging_fits(p)) return DA_ERR_UNSUPPORTED;
  if (p.stats_out || p.ln_stats || p.vt || p.xa_k) {
    if constexpr (CONV) return DA_ERR_UNSUPPORTED;
    else return dispatch_lnf(p, tile, staging, s);
  }
  const bool geglu = (p.act == DA_ACT_GEGLU || p.act == DA_ACT_GEGLU_TANH);
  // GEGLU: the value / gate column tiles of a 64-row group must share a wave (wave tiles whose width is a multiple of 64)
  if (geglu && tile != DA_TILE_K2_128x128 && tile != DA_TILE_K1_256x128 && tile != DA_TILE_K1_128x256 && tile != DA_TILE_K1_256x256 &&
      tile != DA_TILE_K1_256x320 && tile != DA_TILE_K1_128x320)
    return DA_ERR_UNSUPPORTED;
  const int ns = (staging == DA_STAGE_LDS_DIRECT || staging == DA_STAGE_PINGPONG) ? 2
                 : (staging == DA_STAGE_LDS_DIRECT3 || staging == DA_STAGE_PINGPONG3) ? 3 : 0;
  if (ns == 0) return DA_ERR_UNSUPPORTED;
  if (staging == DA_STAGE_PINGPONG || staging == DA_STAGE_PINGPONG3) {   // the K-groups half an iteration apart (KG == 2 tiles)
    switch (tile) {
      case DA_TILE_K2_128x128:
        if (ns == 2) return launch<2, 2, 2, 4, 4, 2, CONV, true>(p, s);
        break;
      case DA_TILE_K2_128x80:
        return ns == 2 ? launch<2, 4, 1, 2, 5, 2, CONV, true>(p, s) : launch<2, 4, 1, 2, 5, 3, CONV, true>(p, s);
      case DA_TILE_K2_128x160:
        if (ns == 2) return launch<2, 2, 2, 4, 5, 2, CONV, true>(p, s);
        break;
      case DA_TILE_K2_80x128:
        if constexpr (!CONV) return ns == 2 ? launch<2, 1, 4, 5, 2, 2, false, true>(p, s) : launch<2, 1, 4, 5, 2, 3, false, true>(p, s);
        break;
      case DA_TILE_K2_128x64:
        return ns == 2 ? launch<2, 2, 2, 4, 2, 2, CONV, true>(p, s) : launch<2, 2, 2, 4, 2, 3, CONV, true>(p, s);
    }
    return DA_ERR_UNSUPPORTED;
  }
  switch (tile) {
    case DA_TILE_K2_128x128:
      if (ns == 2) return launch<2, 2, 2, 4, 4, 2, CONV>(p, s);
      break;
    case DA_TILE_K2_128x80:
      return ns == 2 ? launch<2, 4, 1, 2, 5, 2, CONV>(p, s) : launch<2, 4, 1, 2, 5, 3, CONV>(p, s);
    case DA_TILE_K2_128x160:
      if (ns == 2) return launch<2, 2, 2, 4, 5, 2, CONV>(p, s);
      break;
    case DA_TILE_K2_80x128:
      if constexpr (!CONV) return ns == 2 ? launch<2, 1, 4, 5, 2, 2, false>(p, s) : launch<2, 1, 4, 5, 2, 3, false>(p, s);
      break;
    case DA_TILE_K2_128x64:
      return ns == 2 ? launch<2, 2, 2, 4, 2, 2, CONV>(p, s) : launch<2, 2, 2, 4, 2, 3, CONV>(p, s);
    // KG == 1: eight waves on every slice, large tiles
    case DA_TILE_K1_128x320:                                  // 4 x 2 waves of 32 x 160
      if (ns != 2) break;
      if (!geglu) return launch<1, 4, 2, 2, 10, 2, CONV>(p, s);
      // GEGLU: interleaved column ownership (a 160-column wave tile is not a whole number of [32 | 32] groups), whole output
      // rows from LDS (16-byte aligned output rows).  M 2048 x N 10240 = 512 tiles: two exact rounds of 256 workgroups.
      if constexpr (!CONV) {
        if ((p.ldc & 7) || ((size_t)p.C & 15) || (p.N & 15)) break;
        return launch<1, 4, 2, 2, 10, 2, false, false, false, true>(p, s);
      }
      break;
    case DA_TILE_K1_256x128:                                  // 4 x 2 waves of 64 x 64
      return ns == 2 ? launch<1, 4, 2, 4, 4, 2, CONV>(p, s) : launch<1, 4, 2, 4, 4, 3, CONV>(p, s);
    case DA_TILE_K1_128x256:                                  // 2 x 4 waves of 64 x 64
      return ns == 2 ? launch<1, 2, 4, 4, 4, 2, CONV>(p, s) : launch<1, 2, 4, 4, 4, 3, CONV>(p, s);
    case DA_TILE_K1_256x160:                                  // 4 x 2 waves of 64 x 80
      if (ns == 2) return launch<1, 4, 2, 4, 5, 2, CONV>(p, s);
      break;
    case DA_TILE_K1_256x320:                                  // 4 x 2 waves of 64 x 160, streaming-W loop (nn.Linear only)
      if constexpr (!CONV) {
        if (ns != 2) break;
        if (!geglu) return launch<1, 4, 2, 4, 10, 2, false, false, true, false>(p, s);
        // GEGLU: interleaved column ownership, whole output rows from LDS (16-byte aligned output rows)
        if ((p.ldc & 7) || ((size_t)p.C & 15) || (p.N & 15)) break;
        return launch<1, 4, 2, 4, 10, 2, false, false, true, true>(p, s);
      }
      break;
    case DA_TILE_K1_256x256:                                  // 2 x 4 waves of 128 x 64
      if constexpr (!CONV) {                                  // (the conv build of this tile needs 19 registers it does not have)
        if (ns == 2) return launch<1, 2, 4, 8, 4, 2, false>(p, s);
      }
      break;
  }
  return DA_ERR_UNSUPPORTED;
}

}  // namespace da_gemm2
