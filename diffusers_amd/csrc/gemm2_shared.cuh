// Helpers shared by the K2 / K1 family (gemm2_kernel.cuh) and the eight-phase 256 x 256 tile (gemm3.hip): the wave-uniform buffer
// descriptor, the fused epilogue of one 4-channel group, the XCD rectangle choice and the 31-bit staging budget.  (Split out of
// gemm2_kernel.cuh in round 5 so that gemm3.hip does not instantiate that header's kernels.)
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.cuh"
#include "diffusers_amd.h"

namespace da_gemm2 {

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, size_t bytes) {
  const uint64_t v = (uint64_t)base;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  const int n = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffffull ? 0x7fffffffull : bytes));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}
#endif

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// One 4-channel group of one output row: the fused epilogue of gemm_kernel.cuh (bias, per-row bias, per-batch channel
// vector, activation, gate, residual, output scale) with the same rounding points.  o[] in, o[] out (fp32).
// ACT / GATE are compile-time (ACT = -1: decided at run time, the gated launches): the epilogue of a launch is ONE straight
// line of code per output tile -- with the activation switch inside the unrolled tile loop every tile jumped through its own
// copy of a 4 KB switch and the epilogue, 650-820 cycles per tile, was bound by instruction fetch
// (profiles/r03e_gemm_stage_trace.md).  Absent operands are ZERO vectors (adding a bf16 zero / multiplying by 1.0f is exact),
// so there is no per-tile branch on them either.  FINISH = false stops in front of the residual: the row-contiguous store
// path adds it (and the output scale) after the values went through LDS -- the same fp32 operations in the same order.
// GATE: 0 none, 1 bf16 [B][ld_gate], 2 fp32.
template <int ACT, int GATE, bool FINISH>
__device__ __forceinline__ void epilogue4(const da_gemm_params& p, float* o, int n, int bidx, float brow, uint2 bv, uint2 rv, uint2 resv) {
  o[0] += bf_lo(bv.x); o[1] += bf_hi(bv.x); o[2] += bf_lo(bv.y); o[3] += bf_hi(bv.y);
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] += brow;
  o[0] += bf_lo(rv.x); o[1] += bf_hi(rv.x); o[2] += bf_lo(rv.y); o[3] += bf_hi(rv.y);
  const int act = ACT >= 0 ? ACT : p.act;
  if (act == DA_ACT_GELU_TANH) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = gelu_tanh_f(bf2f(f2bf(o[e])));
  } else if (act == DA_ACT_SILU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = silu_f(bf2f(f2bf(o[e])));
  } else if (act == DA_ACT_GELU_ERF) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = gelu_erf_f(bf2f(f2bf(o[e])));
  } else if (act == DA_ACT_QUICK_GELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xv = bf2f(f2bf(o[e]));
      const float tv = bf2f(f2bf(1.702f * xv));
      o[e] = xv * bf2f(f2bf(1.0f / (1.0f + __expf(-tv))));
    }
  }
  if constexpr (GATE == 2) {
    const float4 gv = *(const float4*)((const float*)p.gate + (size_t)bidx * p.ld_gate + n);
    o[0] = bf2f(f2bf(o[0])) * gv.x; o[1] = bf2f(f2bf(o[1])) * gv.y;
    o[2] = bf2f(f2bf(o[2])) * gv.z; o[3] = bf2f(f2bf(o[3])) * gv.w;
  } else if constexpr (GATE == 1) {
    const uint2 gv = *(const uint2*)((const uint16_t*)p.gate + (size_t)bidx * p.ld_gate + n);
    o[0] = bf2f(f2bf(bf2f(f2bf(o[0])) * bf_lo(gv.x)));
    o[1] = bf2f(f2bf(bf2f(f2bf(o[1])) * bf_hi(gv.x)));
    o[2] = bf2f(f2bf(bf2f(f2bf(o[2])) * bf_lo(gv.y)));
    o[3] = bf2f(f2bf(bf2f(f2bf(o[3])) * bf_hi(gv.y)));
  }
  if constexpr (FINISH) {
    o[0] += bf_lo(resv.x); o[1] += bf_hi(resv.x); o[2] += bf_lo(resv.y); o[3] += bf_hi(resv.y);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] *= p.out_scale;
  }
}

// ---- host side ----
// `wrow`: bytes of one weight row relative to one activation row (1 for nn.Linear; a k x k conv's weight row is k * k activation rows
// long, and the taps' re-reads of the activations are L2 hits as long as the XCD's rows of a tap sweep stay resident: the rectangle's
// weight rows are weighed by it -- fewer XCDs then fetch each weight row of a deep-K conv; xcd_conv_weighting() below).
inline int choose_xcd_gx2(int tiles_m, int tiles_n, int BM, int BN, int wrow = 1, long long arow_bytes = 0) {   // as da_gemm::choose_xcd_gx
  // experiments: DA_XCD_GX = 1 / 2 / 4 / 8 pins the number of XCD columns (1: every XCD owns whole row panels -- it reads rows the
  // previous launch's same-numbered XCD wrote; 8: whole column panels -- each weight row is fetched by one XCD only)
  static const int forced = [] { const char* v = getenv("DA_XCD_GX"); return v ? atoi(v) : 0; }();
  if (forced == 1 || forced == 2 || forced == 4 || forced == 8) return forced;
  int best = 1;
  double best_cost = 1e300;
  for (int gx = 1; gx <= 8; gx *= 2) {
    const int gy = 8 / gx;
    const int tm_per = (tiles_m + gy - 1) / gy, tn_per = (tiles_n + gx - 1) / gx;
    const double inflation = (double)(8 * tm_per * tn_per) / ((double)tiles_m * tiles_n);
    double cost = ((double)tm_per * BM + (double)tn_per * BN * wrow) * inflation * inflation;
    // weighted form only: a rectangle whose activation rows (all channels) do not stay in the 4 MiB L2 pays them once per tap
    if (wrow > 1 && (double)tm_per * BM * (double)arow_bytes > 2.75 * 1048576.0) cost = ((double)tm_per * BM + (double)tn_per * BN) * wrow * inflation * inflation;
    if (cost < best_cost) {
      best_cost = cost;
      best = gx;
    }
  }
  return best;
}

// Channel chunk of a k x k conv's K order, in 64-wide slices (gemm2_kernel.cuh "K order of a k x k conv: CHUNKED").  First form: ONE
// chunk size for every k x k conv, DA_CONV_CHUNK = n channels (multiples of 64).  Measured (round 6, same box, SDXL
// image, images/s): tap-major 0.9869 / 0.9923, chunks of 128 channels 0.9689 / 0.9710, 256: 0.9802, 512: 0.9885 -- the chunked order
// cuts the fetched bytes of the deep-K convs by up to 5.5x (profiles/r06_sdxl_traffic_conv_chunk128.md: 404 -> 74 MB at
// M 32768 x N 320 x K 5760) and LOSES 2 % of the image, because a tap change recomputes the per-lane gather offsets (now every slice
// pair instead of every C / 64 slices) and these launches were not bound by the fabric in the first place.
// Round 6, second form (DEFAULT; DA_CONV_CHUNK=0 restores the tap-major order everywhere, DA_CONV_CHUNK=<n> pins chunks of n channels):
// chunk only the launches whose tap sweep does NOT stay in the L2 -- the rows of the tiles co-resident on an XCD (`rows_xcd`: 32 CUs,
// one tile each, the rectangle's column tiles sharing rows) x all channels x 2 B above 3 MiB -- with the largest chunk whose sweep is
// <= 2 MiB, and only if that chunk is >= 256 channels (a tap change, with its per-lane offset recompute, at most every fourth slice).
// Shallow convs and the VAE's 128 / 256-channel levels keep the tap-major order.  Measured in situ (profiles/r06c_conv_knobs.md):
// M 32768 x N 320 x K 5760 fetches 74 MB instead of 403 and runs 114 instead of 120 us, K 8640: 111 instead of 605 MB, 166 instead of
// 173 us; M 8192 x N 640 x K 17280 235 instead of 454 MB at the same time; every other launch of the step is untouched; the image is
// the same speed within the pair-to-pair spread (same box: 0.9870 / 0.9863 tap-major, 0.9865 / 0.9897 auto).
inline int conv_chunk_slices(const da_gemm_params& p, long long rows_xcd = 0) {
  static const int forced = [] { const char* v = getenv("DA_CONV_CHUNK"); return v ? (v[0] == 'a' ? -1 : atoi(v)) : -1; }();
  if (p.conv <= 1) return 0;
  if (forced < 0) {
    const long long ctot = p.C1 + p.C2;
    if (rows_xcd <= 0 || rows_xcd * ctot * 2 <= 3ll * 1048576) return 0;
    const long long ch = (2ll * 1048576 / (rows_xcd * 2)) / 64;
    return (ch < 4 || ch * 64 >= ctot) ? 0 : (int)ch;
  }
  return forced > 0 ? forced / 64 : 0;
}
// XCD rectangle of a k x k conv (DEFAULT on; DA_XCD_CONV=0 = the nn.Linear rule for every launch): weight rows count k * k times.  A
// mapping only -- every tile computes the same sums -- so results are bit-identical.  In situ: M 2048 x N 1280 x K 11520 (ten launches
// of a step) fetches 87 MB instead of 134, K 5760 36 instead of 65, M 8192 x N 640 x K 5760 60 instead of 80; times within 2 %.
inline int xcd_conv_weighting(const da_gemm_params& p) {
  static const int on = [] { const char* v = getenv("DA_XCD_CONV"); return v ? atoi(v) : 1; }();
  return (on && p.conv > 1) ? p.conv * p.conv : 1;
}

// 31-bit offset budget of the buffer-addressed staging (as da_gemm::buffer_staging_fits, for tiles up to 256 rows)
inline bool staging_fits(const da_gemm_params& p) {
  const size_t lim = 0x3fffffffull;
  if ((size_t)p.ldw * 512 >= lim || (size_t)p.K * 2 >= lim) return false;
  if (!p.conv) return (size_t)p.lda * 512 < lim;
  const size_t cmax = (size_t)(p.C1 > p.C2 ? p.C1 : p.C2);
  const size_t span = (size_t)256 * p.stride * p.stride + (size_t)(6 + 2 * p.stride) * p.Win + 64;
  return span * cmax * 2 < lim && (size_t)p.M / ((size_t)p.Hout * p.Wout) * p.Hin * p.Win < 0x7fffffffull;
}

}  // namespace da_gemm2
