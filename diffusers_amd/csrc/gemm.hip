// bf16 MFMA GEMM / implicit-GEMM conv3x3 for gfx950 (MI355X).
//
// One kernel template serves every dense contraction on the denoising hot path:
//   * nn.Linear           out[M][N] = X[M][K] . W[N][K]^T           (reference: torch F.linear call sites,
//                          diffusers models/attention_processor.py:2743-2777, models/activations.py:113-124,
//                          models/attention.py:1717-1742)
//   * nn.Conv2d 3x3 / s1,s2 / pad 1 on channels-last activations as an implicit GEMM
//                          (reference: models/resnet.py:340-373, models/downsampling.py:145,
//                          models/upsampling.py:177-190 with the nearest-2x gather fused into the A loader,
//                          models/unets/unet_2d_blocks.py:2444 with the skip concat read from two base pointers)
//
// Design (CDNA4): 256 threads = 4 waves (2x2), each wave owns MT x NT tiles of 32x32 computed with
// v_mfma_f32_32x32x16_bf16.  The MFMA "A" operand is the WEIGHT tile and the "B" operand the ACTIVATION tile, so
// each lane ends up holding 4 consecutive output channels of one output row -> 8-byte packed bf16 stores and
// per-lane fused epilogues (bias, per-batch channel vector, residual, GEGLU / GELU / SiLU).
// K is consumed in 64-wide slices (one 128-byte line per tile row) through a double-buffered LDS image whose
// 16-byte slots are XOR-swizzled with ((row>>1)&7) so every ds_read_b128 lane group is bank-conflict free.
// Staging is either register-staged (global_load_dwordx4 -> ds_write_b128) or direct-to-LDS (LDS-DMA: buffer_load ... lds
// through a per-tile descriptor, or global_load_lds_dwordx4 with per-lane pointers; swizzle applied on the per-lane SOURCE
// offset, LDS image lane-linear) -- see the staging-mode notes in gemm_kernel.cuh.
#include "gemm_kernel.cuh"

namespace da_gemm {
int dispatch_conv(const da_gemm_params& p, int tile, int staging, hipStream_t s);  // gemm_conv.hip
}
namespace da_gemm2 {   // the K2 / K1 family (gemm2_kernel.cuh) compiles in its own two translation units
int dispatch_lin(const da_gemm_params& p, int tile, int staging, hipStream_t s);   // gemm2_lin.hip
int dispatch_conv(const da_gemm_params& p, int tile, int staging, hipStream_t s);  // gemm2_conv.hip
}
namespace da_gemm3 {   // the eight-phase 256 x 256 tile (gemm3.hip), nn.Linear only
int dispatch_lin(const da_gemm_params& p, int tile, int staging, hipStream_t s);
}

namespace {

struct TileShape {
  int bm, bn, waves;
};
constexpr TileShape kTiles[] = {{0, 0, 0},      {128, 128, 4}, {64, 128, 4},  {128, 64, 4}, {64, 64, 4},
                                {256, 128, 8},  {128, 256, 8}, {256, 256, 8}, {128, 128, 8},
                                // K2 family (gemm2_kernel.cuh)
                                {128, 128, 8},  {128, 80, 8},  {128, 160, 8}, {80, 128, 8}, {128, 64, 8},
                                {128, 320, 8},  {256, 128, 8}, {128, 256, 8}, {256, 160, 8}, {256, 256, 8}, {256, 320, 8},
                                // K3 (gemm3.hip)
                                {256, 256, 8},  {256, 320, 8}};
static_assert(sizeof(kTiles) / sizeof(kTiles[0]) == DA_TILE_COUNT, "kTiles / DA_TILE_* mismatch");
inline bool is_k3(int tile) { return tile == DA_TILE_K3_256x256 || tile == DA_TILE_K3_256x320; }
inline bool is_k2(int tile) { return tile >= DA_TILE_K2_128x128 && !is_k3(tile); }
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

// Untuned fallback: fewest bytes staged per flop among the tiles that still give every CU a block; output-channel
// counts that are a multiple of 64 but not of 128 (320, 960, 1920) take 64-wide tiles so no MFMA column is wasted.
int pick_tile(const da_gemm_params& p) {
  auto nblk = [&](int t) {
    return (long)((p.M + kTiles[t].bm - 1) / kTiles[t].bm) * ((p.N + kTiles[t].bn - 1) / kTiles[t].bn);
  };
  const bool geglu = (p.act == DA_ACT_GEGLU || p.act == DA_ACT_GEGLU_TANH);
  const bool narrow = !geglu && ((p.N & 127) != 0) && ((p.N % 128) <= 64);
  if (narrow) return (nblk(DA_TILE_128x64) >= 200) ? DA_TILE_128x64 : DA_TILE_64x64;
  if (nblk(DA_TILE_256x128) >= 480) return DA_TILE_256x128;
  if (nblk(DA_TILE_128x128) >= 128) return DA_TILE_128x128;
  if (nblk(DA_TILE_64x128) >= 128 || geglu) return DA_TILE_64x128;
  return DA_TILE_64x64;
}

int validate(da_gemm_params& p) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return DA_ERR_INVALID;
  if ((p.K & 63) || (p.N & 3) || (p.ldc & 3)) return DA_ERR_UNSUPPORTED;
  if (!p.A || !p.W || !p.C) return DA_ERR_INVALID;
  if (p.residual && (p.ldr & 3)) return DA_ERR_UNSUPPORTED;
  if (p.rowvec && (p.rows_per_batch <= 0 || (p.ld_rowvec & 3))) return DA_ERR_INVALID;
  if (p.gate && (p.rows_per_batch <= 0 || (p.ld_gate & 3))) return DA_ERR_INVALID;
  if (p.alpha == 0.0f) p.alpha = 1.0f;
  if (p.out_scale == 0.0f) p.out_scale = 1.0f;
  if (p.conv) {
    if (p.C1 <= 0 || (p.C1 & 63) || (p.C2 & 63) || p.C2 < 0) return DA_ERR_UNSUPPORTED;
    if (p.C2 > 0 && !p.A2) return DA_ERR_INVALID;
    if (p.conv != 1 && p.conv != 3) return DA_ERR_UNSUPPORTED;
    if (p.K != p.conv * p.conv * (p.C1 + p.C2)) return DA_ERR_INVALID;
    if (p.stride != 1 && p.stride != 2) return DA_ERR_UNSUPPORTED;
    if (p.up != 0 && p.up != 1) return DA_ERR_INVALID;
    if (p.Hin <= 0 || p.Win <= 0 || p.Hout <= 0 || p.Wout <= 0) return DA_ERR_INVALID;
    if (p.M % (p.Hout * p.Wout)) return DA_ERR_INVALID;
  } else {
    if ((p.lda & 7) || (p.ldw & 7)) return DA_ERR_UNSUPPORTED;
  }
  if ((p.act == DA_ACT_GEGLU || p.act == DA_ACT_GEGLU_TANH) &&
      ((p.N & 127) || p.out_f32 || p.residual || p.rowvec || p.gate || p.bias_rows))
    return DA_ERR_UNSUPPORTED;
  if (p.act < 0 || p.act > DA_ACT_GEGLU_TANH) return DA_ERR_INVALID;
  if (p.split_k < 0 || p.split_k > DA_SPLITK_MAX) return DA_ERR_INVALID;
  if (p.split_k > 1 && (p.stats_out || p.ln_stats)) return DA_ERR_UNSUPPORTED;   // no split-K build of the LayerNorm fold
  if (p.k_valid < 0 || (p.k_valid > 0 && (p.k_valid > (p.conv ? p.C1 : p.K) || (p.conv && p.C2 != 0)))) return DA_ERR_INVALID;
  if (p.stats_out && (p.conv || p.out_f32 || p.act == DA_ACT_GEGLU || p.act == DA_ACT_GEGLU_TANH || p.stats_ld <= 0 || (p.stats_ld & 1)))
    return DA_ERR_UNSUPPORTED;
  if (p.ln_stats && (p.conv || !p.ln_s || !p.ln_c || p.ln_parts <= 0 || p.ln_parts > 4 * DA_LN_PAIR_LOADS ||
                     p.ln_stats_ld < 2 * DA_LN_MAX_PARTS || (p.ln_stats_ld & 3)))
    return DA_ERR_INVALID;
  return DA_OK;
}

int stats_parts(const da_gemm_params& p, int tile) {   // one partial per column tile (K2 family: per 80-column wave band)
  if (is_k2(tile)) {
    const int band = (tile == DA_TILE_K2_128x80 || tile == DA_TILE_K2_128x160) ? 80 : 0;
    return band ? (p.N + band - 1) / band : 0;
  }
  return (p.N + kTiles[tile].bn - 1) / kTiles[tile].bn;
}

bool tile_ok(const da_gemm_params& p, int tile) {
  if (tile <= 0 || tile >= kNumTiles) return false;
  if (is_k3(tile)) {  // nn.Linear, plain or GEGLU epilogue, nothing that needs the other families' extra instantiations
    if (p.conv || p.split_k > 1 || p.stats_out || (p.ln_stats && tile != DA_TILE_K3_256x320) || p.vt || p.xa_k) return false;
    if (tile == DA_TILE_K3_256x320)   // the GEGLU projection's tile: whole tiles, aligned output rows (round 6: also as a LayerNorm-fold consumer)
      return (p.act == DA_ACT_GEGLU || p.act == DA_ACT_GEGLU_TANH) && (p.M % 256) == 0 && (p.N % 320) == 0 && !(p.ldc & 7) &&
             !((size_t)p.C & 15);
    return true;
  }
  if (is_k2(tile)) {
    const bool geglu = (p.act == DA_ACT_GEGLU || p.act == DA_ACT_GEGLU_TANH);
    const bool geglu_ok = tile == DA_TILE_K2_128x128 || tile == DA_TILE_K1_256x128 || tile == DA_TILE_K1_128x256 ||
                          tile == DA_TILE_K1_256x256 || tile == DA_TILE_K1_256x320 || (tile == DA_TILE_K1_128x320 && !p.conv);
    // LayerNorm fold (round 4): nn.Linear, the tiles of the SDXL transformer blocks (gemm2_kernel.cuh dispatch_lnf)
    const bool lnf_lin = tile == DA_TILE_K2_128x80 || tile == DA_TILE_K2_128x160;          // producer and consumer
    const bool lnf_cons = lnf_lin || tile == DA_TILE_K1_128x256 || tile == DA_TILE_K1_256x128;   // consumer / transposed block only
    if (p.xa_k) return !p.conv && p.split_k <= 1 && !geglu && !p.stats_out && !p.vt && tile == DA_TILE_K2_128x128;
    if (p.vt) return !p.conv && p.split_k <= 1 && !geglu && !p.stats_out && lnf_cons;
    if (p.stats_out || p.ln_stats) {
      if (p.conv || p.split_k > 1) return false;
      if (p.stats_out && (geglu || !lnf_lin)) return false;
      return geglu ? tile == DA_TILE_K1_128x320 : lnf_cons;
    }
    return p.split_k <= 1 && (!geglu || geglu_ok) &&
           !(p.conv && (tile == DA_TILE_K2_80x128 || tile == DA_TILE_K1_256x256 || tile == DA_TILE_K1_256x320));
  }
  if (p.vt || p.xa_k) return false;   // the transposed column block / cross-attention epilogue exist in the second family only
  // GEGLU pairs (value, gate) 32-column tiles inside one wave: the wave must own an even number of them
  if ((p.act == DA_ACT_GEGLU || p.act == DA_ACT_GEGLU_TANH) &&
      (tile == DA_TILE_128x64 || tile == DA_TILE_64x64 || tile == DA_TILE_128x128_W8))
    return false;
  // the 4 x 2 wave tile has no registers for the LayerNorm fold (either side)
  if (tile == DA_TILE_256x256 && (p.stats_out || p.ln_stats)) return false;
  return true;
}

// Reads n 16-byte chunks (grid-stride) and keeps nothing: brings a tensor back into the memory-side cache for da_gemm_tune.
__global__ __launch_bounds__(256) void touch_kernel(const uint4* __restrict__ src, size_t n, unsigned* __restrict__ sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint4 v = src[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x9e3779b9u) sink[0] = acc;   // (never in practice: keeps the loads alive)
}

int run(const da_gemm_params& p, int tile, int staging, hipStream_t s, const da_gemm_params* pb = nullptr) {
  if (is_k3(tile)) return pb ? DA_ERR_UNSUPPORTED : da_gemm3::dispatch_lin(p, tile, staging, s);
  if (is_k2(tile)) {
    if (pb) return DA_ERR_UNSUPPORTED;
    return p.conv ? da_gemm2::dispatch_conv(p, tile, staging, s) : da_gemm2::dispatch_lin(p, tile, staging, s);
  }
  if (p.conv) return pb ? DA_ERR_UNSUPPORTED : da_gemm::dispatch_conv(p, tile, staging, s);
  return da_gemm::dispatch<false>(p, tile, staging, s, pb);
}

}  // namespace

extern "C" int da_gemm_bf16(const da_gemm_params* pp, void* stream) {
  if (!pp) return DA_ERR_INVALID;
  da_gemm_params p = *pp;
  const int v = validate(p);
  if (v != DA_OK) return v;
  int tile = p.tile;
  if (tile == DA_TILE_AUTO) tile = pick_tile(p);
  if (!tile_ok(p, tile)) return DA_ERR_UNSUPPORTED;
  if (p.stats_out && (stats_parts(p, tile) <= 0 || p.stats_ld < 2 * stats_parts(p, tile))) return DA_ERR_INVALID;
  return run(p, tile, p.staging, (hipStream_t)stream);
}

extern "C" int da_gemm_stats_parts(const da_gemm_params* pp) {
  if (!pp || pp->N <= 0) return 0;
  int tile = pp->tile;
  if (tile == DA_TILE_AUTO) tile = pick_tile(*pp);
  if (tile <= 0 || tile >= kNumTiles) return 0;
  return stats_parts(*pp, tile);
}

extern "C" int da_gemm_pair_bf16(const da_gemm_params* pa, const da_gemm_params* pb, void* stream) {
  if (!pa || !pb) return DA_ERR_INVALID;
  da_gemm_params a = *pa, b = *pb;
  int v = validate(a);
  if (v != DA_OK) return v;
  v = validate(b);
  if (v != DA_OK) return v;
  if (a.conv || b.conv || a.split_k > 1 || b.split_k > 1) return DA_ERR_UNSUPPORTED;
  int tile = a.tile;
  if (tile == DA_TILE_AUTO) tile = pick_tile(a);
  if (!tile_ok(a, tile) || !tile_ok(b, tile)) return DA_ERR_UNSUPPORTED;
  return run(a, tile, a.staging, (hipStream_t)stream, &b);
}

// Times every (tile, staging[, split_k]) variant that can run this problem on `stream` (HIP events, min of `iters`
// launches each after one warm launch) and returns the fastest.  All variants of one split factor walk K in the same
// order with the same MFMA, so they produce bit-identical C: tuning changes speed only (a split factor > 1 changes the
// fp32 summation order; it is only considered when the caller asks for it).  Must not be called while the stream is
// being captured.
extern "C" int da_gemm_tune(const da_gemm_params* pp, const da_gemm_params* pair, void* stream, int iters, void* scratch,
                            size_t scratch_bytes, int* best_tile, int* best_staging, int* best_split, float* best_us) {
  if (!pp || !best_tile || !best_staging) return DA_ERR_INVALID;
  da_gemm_params p = *pp;
  int v = validate(p);
  if (v != DA_OK) return v;
  da_gemm_params pb{};
  if (pair) {
    pb = *pair;
    v = validate(pb);
    if (v != DA_OK) return v;
    if (p.conv || pb.conv) return DA_ERR_UNSUPPORTED;
  }
  if (iters <= 0) iters = 3;
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return DA_ERR_LAUNCH;
  float best = 3.0e38f;
  int bt = 0, bs = 0, bk = 1;
  static const int stagings[] = {DA_STAGE_LDS_DIRECT, DA_STAGE_LDS_DIRECT3, DA_STAGE_LDS_DIRECT4, DA_STAGE_LDS_DIRECT6,
                                 DA_STAGE_LDS_DIRECT8, DA_STAGE_PINGPONG, DA_STAGE_PINGPONG3};
  constexpr int n_stagings = sizeof(stagings) / sizeof(stagings[0]);
  // split factors tried next to the unsplit variants (round 4: up to 8 -- the 8 x 8 / 16 x 16 levels of the SD1.5 and DDPM
  // U-Nets are 3 x 3 convs with M = 64 .. 512 rows and K = 4.6 k .. 23 k: 8 .. 80 tiles walking 72 .. 360 K slices each)
  // Round 6: 12, 16 and 24 as well -- an M = 64 .. 128 conv of those levels is 10 .. 20 tiles streaming 10 .. 60 MB of weights, and at split 8
  // only 80 .. 160 workgroups have loads in flight (0.8 TB/s: bound by bytes in flight, not by the fabric).
  static const int splits[] = {1, 2, 3, 4, 6, 8, 12, 16, 24};
  const int n_splits = (best_split && !pair && p.workspace && p.sync_flags) ? (int)(sizeof(splits) / sizeof(splits[0])) : 1;
  const int family = pp->tile;   // DA_TILE_AUTO: every variant; DA_TILE_FAMILY_1 / DA_TILE_FAMILY_K2: one kernel family only
  for (int spi = 0; spi < n_splits; ++spi) {
    const int split = splits[spi];
    p.split_k = split;
    for (int tile = 1; tile < kNumTiles; ++tile) {
      // (the K3 tile sums K in the K1 / K2 family's order: it competes wherever that family does)
      if ((family == DA_TILE_FAMILY_1 && (is_k2(tile) || is_k3(tile))) || (family == DA_TILE_FAMILY_K2 && !is_k2(tile) && !is_k3(tile))) continue;
      if (!tile_ok(p, tile) || (pair && !tile_ok(pb, tile))) continue;
      // a tile more than twice the problem in either dimension only wastes MFMA rows
      if (kTiles[tile].bm >= 2 * p.M + 64 || kTiles[tile].bn >= 2 * p.N + 64) continue;
      if (split > 1) {
        // splitting pays only when the unsplit launch leaves CUs idle
        const long tiles = (long)((p.M + kTiles[tile].bm - 1) / kTiles[tile].bm) * ((p.N + kTiles[tile].bn - 1) / kTiles[tile].bn);
        if (tiles >= 256 || (p.K >> 6) < 4 * split) continue;
      }
      for (int si = 0; si < n_stagings; ++si) {
        const int st = stagings[si];
        int rc = run(p, tile, st, s, pair ? &pb : nullptr);  // warm launch (also sets the LDS attribute once)
        if (rc == DA_ERR_UNSUPPORTED || (split > 1 && rc == DA_ERR_INVALID)) continue;
        if (rc != DA_OK) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
        float tmin = 3.0e38f;
        for (int it = 0; it < iters; ++it) {
          // evict the operands from L2 / Infinity Cache: in the denoising loop the weights of a layer were last touched
          // one whole step (5 GB of other weights) ago, so the variant must be chosen for HBM-latency operands
          if (scratch && scratch_bytes) {
            (void)hipMemsetAsync(scratch, 0, scratch_bytes, s);
            // ... but the ACTIVATIONS were written by the kernel in front of this one: they sit in the memory-side cache when
            // the launch starts.  Re-reading them (touch_kernel) restores that state; timing with
            // everything cold picked variants that lost in the loop (FF-down: 42 us in situ against 36 for the runner-up).
            const da_gemm_params* both[2] = {&p, pair ? &pb : nullptr};
            for (int w = 0; w < 2; ++w) {
              if (!both[w]) continue;
              const da_gemm_params& q = *both[w];
              const size_t na = q.conv ? (size_t)(q.M / ((size_t)q.Hout * q.Wout)) * q.Hin * q.Win * q.C1 * 2
                                       : ((size_t)(q.M - 1) * q.lda + q.K) * 2;
              if (na <= ((size_t)192 << 20))
                DA_LAUNCH(touch_kernel, dim3(1024), dim3(256), 0, s, (const uint4*)q.A, na / 16, (unsigned*)scratch);
            }
          }
          (void)hipEventRecord(e0, s);
          rc = run(p, tile, st, s, pair ? &pb : nullptr);
          (void)hipEventRecord(e1, s);
          if (rc != DA_OK || hipEventSynchronize(e1) != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return DA_ERR_LAUNCH; }
          float ms = 0.f;
          (void)hipEventElapsedTime(&ms, e0, e1);
          if (ms < tmin) tmin = ms;
        }
        // a split variant must win by a clear margin (3 %): it costs workspace traffic the microbenchmark under-weighs
        const float score = split > 1 ? tmin * 1.03f : tmin;
        if (score < best) { best = score; bt = tile; bs = st; bk = split; }
      }
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (bt == 0) return DA_ERR_UNSUPPORTED;
  *best_tile = bt;
  *best_staging = bs;
  if (best_split) *best_split = bk;
  if (best_us) *best_us = best * 1000.0f;
  return DA_OK;
}
