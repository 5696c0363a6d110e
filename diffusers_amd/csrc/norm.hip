// Normalisation kernels for gfx950 (MI355X): GroupNorm(+SiLU) on channels-last images, LayerNorm on token rows,
// row softmax.  All are HBM-bound: 16-byte (8 x bf16) accesses per lane, fp32 statistics, one read + one write of
// the tensor for the apply pass and one extra read for the GroupNorm statistics pass.
//
// Reference call sites (diffusers src/diffusers/):
//   nn.GroupNorm(32, C)  models/resnet.py:326,:350  models/transformers/transformer_2d.py:466
//                        models/unets/unet_2d_condition.py:1228  models/autoencoders/vae.py:305
//                        models/attention_processor.py:2740 ; followed by SiLU at resnet.py:327,:362 etc.
//   nn.LayerNorm(C)      models/attention.py:986,:1030,:1056 (BasicTransformerBlock norm1/2/3)
#include <cstdlib>

#include "common.cuh"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// GroupNorm statistics: grid (nblk, B).  Thread t owns channel chunk cc = t % (C/8) (8 channels) and pixel lane
// prow = t / (C/8); it walks pixels prow, prow+k, ... of the block's slab.  Per-thread channel sums go to LDS and
// G threads fold them per group in a fixed order (deterministic), writing one (sum, sumsq) partial per block.
// ------------------------------------------------------------------------------------------------------------------
__global__ void gn_stats_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2, int C1,
                                float* __restrict__ ws, int HW, int C, int G, int pix_per_blk, int krows) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [krows][C][2]
  const int cpr = C >> 3;
  const int t = threadIdx.x;
  const int cc = t % cpr, prow = t / cpr;
  const int b = blockIdx.y, blk = blockIdx.x;
  const int p0 = blk * pix_per_blk;
  const int p1 = min(HW, p0 + pix_per_blk);
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
  // channels [0, C1) come from x ([B][HW][C1]), channels [C1, C) from x2 ([B][HW][C-C1]): fused torch.cat
  const bool second = (cc * 8 >= C1);
  const int Cs = second ? (C - C1) : C1;
  const uint16_t* xb = (second ? x2 + (size_t)(cc * 8 - C1) : x + (size_t)cc * 8) + (size_t)b * HW * Cs;
  for (int pix = p0 + prow; pix < p1; pix += 4 * krows) {  // 4 independent 16-byte loads in flight per lane
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = pix + u * krows;
      v[u] = (pu < p1) ? *(const uint4*)(xb + (size_t)pu * Cs) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float f[8];
      unpack8(v[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s[e] += f[e];
        q[e] += f[e] * f[e];
      }
    }
  }
  float* my = sm + ((size_t)prow * C + cc * 8) * 2;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    my[2 * e] = s[e];
    my[2 * e + 1] = q[e];
  }
  __syncthreads();
  if (t < G) {
    const int cpg = C / G;
    float ts = 0.f, tq = 0.f;
    // eight LDS reads in flight, added in index order (the sequential sum, bit for bit; a slot past the group adds + 0.0f)
    for (int r = 0; r < krows; ++r) {
      const float2* row = (const float2*)(sm + ((size_t)r * C + t * cpg) * 2);
      for (int c0 = 0; c0 < cpg; c0 += 8) {
        float2 pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) pv[u] = (c0 + u < cpg) ? row[c0 + u] : make_float2(0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          ts += pv[u].x;
          tq += pv[u].y;
        }
      }
    }
    float* o = ws + (((size_t)b * gridDim.x + blk) * G + t) * 2;
    o[0] = ts;
    o[1] = tq;
  }
}

// GroupNorm apply (+ optional SiLU): y = act((x - mean_g) * rstd_g * gamma_c + beta_c)
__global__ void gn_apply_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2, int C1,
                                const uint16_t* __restrict__ gamma,
                                const uint16_t* __restrict__ beta, uint16_t* __restrict__ y,
                                const float* __restrict__ ws, int nblk_stats, int HW, int C, int G, float eps, int act,
                                int pix_per_blk, int krows) {
  __shared__ float s_mean[64], s_rstd[64];
  __shared__ double s_part[8][64][2];
  const int cpr = C >> 3;
  const int t = threadIdx.x;
  const int cc = t % cpr, prow = t / cpr;
  const int b = blockIdx.y, blk = blockIdx.x;
  // Every block folds the statistics partials itself (they are L2-resident); the fold is spread over SL slices of G
  // threads so no thread walks more than nblk_stats / SL entries, always in the same order (deterministic).
  int SL = (int)blockDim.x / G;
  if (SL > 8) SL = 8;
  if (SL < 1) SL = 1;
  if (t < G * SL) {
    const int g = t % G, sl = t / G;
    double ts = 0.0, tq = 0.0;
    const float* w = ws + ((size_t)b * nblk_stats * G + g) * 2;
    // 16 partials in flight per thread, added in index order (the sequential sum, bit for bit): with one load per iteration
    // the fold was a chain of 26-49 dependent L2 round trips in front of every block's first pixel
    for (int i = sl; i < nblk_stats; i += 16 * SL) {
      float2 pq[16];
      // (UNCONDITIONAL loads from a clamped index, zeroed by a select: written as `cond ? load : 0` every load sat in its own
      // exec-masked branch with its own s_waitcnt vmcnt(0) -- the sixteen "in flight" were sixteen serialised L2 round trips, twice
      // over for 256 partials, in front of every block's first pixel; round 6, found in the ISA)
      float2 raw[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) raw[u] = *(const float2*)(w + (size_t)min(i + u * SL, nblk_stats - 1) * G * 2);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int idx = i + u * SL;                       // past the end: + 0.0, which changes nothing
        pq[u] = (idx < nblk_stats) ? raw[u] : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        ts += (double)pq[u].x;
        tq += (double)pq[u].y;
      }
    }
    s_part[sl][g][0] = ts;
    s_part[sl][g][1] = tq;
  }
  __syncthreads();
  if (t < G) {
    double ts = 0.0, tq = 0.0;
    for (int sl = 0; sl < SL; ++sl) {
      ts += s_part[sl][t][0];
      tq += s_part[sl][t][1];
    }
    const double n = (double)HW * (double)(C / G);
    const double mean = ts / n;
    double var = tq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    s_mean[t] = (float)mean;
    s_rstd[t] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int cpg = C / G;
  float a[8], c[8];
  {
    float gf[8], bf[8];
    unpack8(*(const uint4*)(gamma + cc * 8), gf);
    unpack8(*(const uint4*)(beta + cc * 8), bf);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (cc * 8 + e) / cpg;
      a[e] = s_rstd[g] * gf[e];
      c[e] = bf[e] - s_mean[g] * a[e];
    }
  }
  const int p0 = blk * pix_per_blk;
  const int p1 = min(HW, p0 + pix_per_blk);
  const size_t base = (size_t)b * HW * C + (size_t)cc * 8;
  const bool second = (cc * 8 >= C1);
  const int Cs = second ? (C - C1) : C1;
  const uint16_t* xb = (second ? x2 + (size_t)(cc * 8 - C1) : x + (size_t)cc * 8) + (size_t)b * HW * Cs;
  for (int pix = p0 + prow; pix < p1; pix += 4 * krows) {  // 4 independent 16-byte loads in flight per lane
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = pix + u * krows;
      v[u] = (pu < p1) ? *(const uint4*)(xb + (size_t)pu * Cs) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = pix + u * krows;
      if (pu >= p1) break;
      float f[8];
      unpack8(v[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float o = f[e] * a[e] + c[e];
        if (act) {
          o = bf2f(f2bf(o));  // reference rounds the GroupNorm output to bf16 before SiLU
          o = silu_f(o);
        }
        f[e] = o;
      }
      *(uint4*)(y + base + (size_t)pu * C) = pack8(f);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// GroupNorm in ONE launch (round 5) for tensors whose per-(batch, group set) slab is small: the two-kernel form above is bound by
// its two launches there (~8 us each for the 5 MB tensors of SDXL's 32 x 32 level, 17 us per GroupNorm against ~2 us of traffic;
// every GroupNorm of the SD1.5 U-Net at 64 x 64 latents is of that kind).  One workgroup owns ALL pixels of one batch for a set of
// GS whole groups whose channel span CW = GS * (C / G) is a multiple of 8 (cpg 10 -> four groups = 40 channels = five 16-byte
// chunks): pass 1 streams the slab once (16-byte chunks, chunk index fastest: a pixel's CW channels are contiguous), parks it in
// LDS when it fits (RES) and accumulates per-thread (sum, sum of squares) per group; the block reduction is a fixed butterfly per
// wave + an in-order fp64 fold over the waves (deterministic); pass 2 normalises out of LDS (or re-reads the slab -- it is
// L2-resident -- when it did not fit).  A thread keeps the same chunk position for the whole kernel (T % nch == 0), so its eight
// channels' group ids, gamma and beta are loop constants.  Same formula and rounding points as gn_apply_kernel.
// ------------------------------------------------------------------------------------------------------------------
//
// Round 6: the same kernel with the pixels of a slab dealt to `nparts` workgroups (gridDim.z; a power of two <= 32), for slabs that
// do not fit one CU's LDS but whose tensor fits the chip's (B * G / GS * nparts <= CUs workgroups of <= 144 KB: every one of them is
// resident, so waiting for each other cannot deadlock).  Each part parks ITS pixels in LDS, reduces its (sum, sum of squares) per group
// in fp64 as before and publishes them with 8-byte agent-scope (write-through) stores; thread 0 adds 32 / nparts to the slab's
// arrival counter and waits -- one lane, relaxed loads, s_sleep, bounded -- until the counter reaches the end of its generation (every
// launch adds exactly 32 per slab, so generations start at multiples of 32 whatever nparts the launches sharing the counter use;
// the counter is monotonic and never reset); then every part folds ALL parts' partials in part order (a fixed order: every part
// computes the same mean / rstd bits) and normalises out of LDS.  The tensor is read ONCE and written once.
constexpr int kGnSyncCounters = 4096;                       // slabs (batch x group set) a sync workspace serves
constexpr int kGnSyncPartBytes = 32 * 4 * 16;               // per slab: 32 parts x up to 4 groups x (sum, sumsq) fp64
constexpr size_t kGnSyncBytes = (size_t)kGnSyncCounters * 4 + 64 + (size_t)kGnSyncCounters * kGnSyncPartBytes;
template <int GS, bool RES>
__global__ __launch_bounds__(1024) void gn_fused_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2, int C1,
                                                        const uint16_t* __restrict__ gamma, const uint16_t* __restrict__ beta,
                                                        uint16_t* __restrict__ y, int HW_all, int C, int G, float eps, int act,
                                                        unsigned char* __restrict__ sync) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  __shared__ float s_part[16][2 * GS];
  __shared__ float s_a[GS], s_c[GS];                        // mean, rstd per group
  __shared__ double s_tot[GS][2];
  // this part's pixels [pix0, pix0 + HW) of the batch's HW_all (one part: everything)
  const int nparts = (int)gridDim.z, part = (int)blockIdx.z;
  const int per_part = (HW_all + nparts - 1) / nparts;
  const int pix0 = part * per_part;
  const int HW = max(0, min(HW_all, pix0 + per_part) - pix0);
  const int T = (int)blockDim.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = T >> 6;
  const int cpg = C / G, CW = GS * cpg, nch = CW >> 3;
  const int b = blockIdx.y, g0 = blockIdx.x * GS, c0 = g0 * cpg;
  const int ch = t % nch;                                   // this thread's 16-byte chunk of every pixel it visits
  const int cabs = c0 + ch * 8;                             // first of its eight channels
  const bool second = cabs >= C1;
  const int Cs = second ? (C - C1) : C1;
  const uint16_t* xb = (second ? x2 + (size_t)(cabs - C1) : x + (size_t)cabs) + ((size_t)b * HW_all + pix0) * Cs;
  int gid[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) gid[e] = (ch * 8 + e) / cpg;  // 0 .. GS - 1
  const int ppr = T / nch;                                  // pixels per round of the whole block
  const int p_first = t / nch;
  float s[GS], q[GS];
#pragma unroll
  for (int g = 0; g < GS; ++g) s[g] = q[g] = 0.f;
  uint4* slab = (uint4*)gsm;
  for (int pix = p_first; pix < HW; pix += 4 * ppr) {       // 4 independent 16-byte loads in flight per lane
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = pix + u * ppr;
      v[u] = (pu < HW) ? *(const uint4*)(xb + (size_t)pu * Cs) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = pix + u * ppr;
      if (pu >= HW) break;
      if constexpr (RES) slab[(size_t)pu * nch + ch] = v[u];
      float f[8];
      unpack8(v[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int g = 0; g < GS; ++g) {
          const float m = (GS == 1 || gid[e] == g) ? f[e] : 0.f;
          s[g] += m;
          q[g] += m * m;
        }
    }
  }
  // wave butterfly (fixed pattern), then the waves in index order in fp64
#pragma unroll
  for (int g = 0; g < GS; ++g) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s[g] += __shfl_xor(s[g], o, 64);
      q[g] += __shfl_xor(q[g], o, 64);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int g = 0; g < GS; ++g) s_part[wave][2 * g] = s[g], s_part[wave][2 * g + 1] = q[g];
  }
  __syncthreads();
  if (t < GS) {
    double ts = 0.0, tq = 0.0;
    for (int w = 0; w < nw; ++w) {
      ts += (double)s_part[w][2 * t];
      tq += (double)s_part[w][2 * t + 1];
    }
    s_tot[t][0] = ts;
    s_tot[t][1] = tq;
  }
  if (nparts > 1) {
    // ---- the parts of a slab meet (see the kernel comment): publish, arrive, wait, fold in part order ----
    const int slab_id = b * (int)gridDim.x + (int)blockIdx.x;
    unsigned int* cnt = (unsigned int*)sync + slab_id;
    unsigned long long* parts = (unsigned long long*)(sync + (size_t)kGnSyncCounters * 4 + 64 + (size_t)slab_id * kGnSyncPartBytes);
    if (t < GS) {
      __hip_atomic_store(parts + (part * 4 + t) * 2 + 0, (unsigned long long)__double_as_longlong(s_tot[t][0]), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(parts + (part * 4 + t) * 2 + 1, (unsigned long long)__double_as_longlong(s_tot[t][1]), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the storing lanes drain before the arrival
    }
    __syncthreads();
    if (t == 0) {
      const unsigned int inc = 32u / (unsigned int)nparts;
      const unsigned int old = __hip_atomic_fetch_add(cnt, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int target = (old & ~31u) + 32u;
      int spins = 0;
      while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1 << 22)) {                          // a lost part must not hang the GPU: flag it and go on (wrong statistics)
          __hip_atomic_store((unsigned int*)(sync + (size_t)kGnSyncCounters * 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
    if (t < GS) {
      double ts = 0.0, tq = 0.0;
      for (int pz = 0; pz < nparts; ++pz) {
        ts += __longlong_as_double((long long)__hip_atomic_load(parts + (pz * 4 + t) * 2 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        tq += __longlong_as_double((long long)__hip_atomic_load(parts + (pz * 4 + t) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
      s_tot[t][0] = ts;
      s_tot[t][1] = tq;
    }
  }
  if (t < GS) {
    const double ts = s_tot[t][0], tq = s_tot[t][1];
    const double n = (double)HW_all * (double)cpg;
    const double mean = ts / n;
    double var = tq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    s_a[t] = (float)mean;
    s_c[t] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  float a[8], c[8];
  {
    float gf[8], bf[8];
    unpack8(*(const uint4*)(gamma + cabs), gf);
    unpack8(*(const uint4*)(beta + cabs), bf);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float mean = s_a[0], rstd = s_c[0];
#pragma unroll
      for (int g = 1; g < GS; ++g)
        if (gid[e] == g) mean = s_a[g], rstd = s_c[g];
      a[e] = rstd * gf[e];
      c[e] = bf[e] - mean * a[e];
    }
  }
  uint16_t* yb = y + ((size_t)b * HW_all + pix0) * C + cabs;
  for (int pix = p_first; pix < HW; pix += 4 * ppr) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = pix + u * ppr;
      if constexpr (RES) v[u] = (pu < HW) ? slab[(size_t)pu * nch + ch] : make_uint4(0, 0, 0, 0);   // its own writes: no barrier
      else v[u] = (pu < HW) ? *(const uint4*)(xb + (size_t)pu * Cs) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = pix + u * ppr;
      if (pu >= HW) break;
      float f[8];
      unpack8(v[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float o = f[e] * a[e] + c[e];
        if (act) {
          o = bf2f(f2bf(o));  // reference rounds the GroupNorm output to bf16 before SiLU
          o = silu_f(o);
        }
        f[e] = o;
      }
      *(uint4*)(yb + (size_t)pu * C) = pack8(f);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// LayerNorm over the last dim: one wave per row, the row lives in registers (NCH chunks of 8 per lane).
// Optional AdaLN modulation: y = LN(x) * (1 + scale[b]) + shift[b]  (normalization.py:157-170, :194-202, :346-351)
// ------------------------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const uint16_t* __restrict__ x,
                                                        const uint16_t* __restrict__ gamma,
                                                        const uint16_t* __restrict__ beta, uint16_t* __restrict__ y,
                                                        const uint16_t* __restrict__ mod_scale,
                                                        const uint16_t* __restrict__ mod_shift, int mod_ld, int mod_f32,
                                                        int rows_per_batch, int M, int C, int ldx, int ldy, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nchunks = C >> 3;
  float v[NCH][8];
  const uint16_t* xr = x + (size_t)row * ldx;
  // the affine parameters are fetched WITH the row (they do not depend on it): their round trip runs under the two wave
  // reductions instead of behind them -- the kernel is latency-bound (10 MB through a 6 us launch)
  uint4 gq[NCH], bq[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = min(lane + 64 * i, nchunks - 1);
    gq[i] = gamma ? *(const uint4*)(gamma + ch * 8) : make_uint4(0, 0, 0, 0);
    bq[i] = beta ? *(const uint4*)(beta + ch * 8) : make_uint4(0, 0, 0, 0);
  }
  float sum = 0.f;
  // (the row's chunks are loaded UNCONDITIONALLY from clamped offsets and zeroed by a select: behind `if (ch < nchunks)` every chunk's
  // load sat in its own exec-masked branch with its own s_waitcnt vmcnt(0) -- NCH serialised round trips per row instead of one; round 6)
  uint4 xq[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) xq[i] = *(const uint4*)(xr + min(lane + 64 * i, nchunks - 1) * 8);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    unpack8(xq[i], v[i]);
    if (ch < nchunks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[i][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  sum = wave_sum(sum);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nchunks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[i][e] - mean;
        sq += d * d;
      }
    }
  }
  sq = wave_sum(sq);
  const float rstd = rsqrtf(sq / (float)C + eps);
  uint16_t* yr = y + (size_t)row * ldy;
  const int bidx = mod_scale ? row / rows_per_batch : 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nchunks) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd;
      if (gamma) {
        float g[8];
        unpack8(gq[i], g);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] *= g[e];
      }
      if (beta) {
        float bb[8];
        unpack8(bq[i], bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += bb[e];
      }
      if (mod_scale && mod_f32) {
        const float* scp = (const float*)mod_scale + (size_t)bidx * mod_ld + ch * 8;
        const float* shp = (const float*)mod_shift + (size_t)bidx * mod_ld + ch * 8;
        const float4 s0 = *(const float4*)scp, s1 = *(const float4*)(scp + 4);
        const float4 h0 = *(const float4*)shp, h1 = *(const float4*)(shp + 4);
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = o[e] * (1.0f + sc[e]) + sh[e];
      } else if (mod_scale) {
        float sc[8], sh[8];
        unpack8(*(const uint4*)(mod_scale + (size_t)bidx * mod_ld + ch * 8), sc);
        unpack8(*(const uint4*)(mod_shift + (size_t)bidx * mod_ld + ch * 8), sh);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = bf2f(f2bf(o[e])) * (1.0f + sc[e]) + sh[e];
      }
      *(uint4*)(yr + ch * 8) = pack8(o);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// RMSNorm over the head dim (torch.nn.RMSNorm(head_dim): Flux q/k norm) or over all heads ("rms_norm_across_heads":
// Wan) followed by the interleaved-pair rotary embedding, IN PLACE on `parts` column blocks (q, k) of a token-major
// buffer.  One wave per (row, part): lane l owns 8-channel chunks l, l+64, ...; a head of D channels is D/8
// consecutive lanes, so the per-head mean square is a shuffle reduction inside aligned lane groups.
//   norm:  y = bf16( x * rsqrt(mean(x^2) + eps) * w )                  (normalization.py:510-569 / torch.nn.RMSNorm)
//   rope:  out[2i] = y[2i] cos[2i] - y[2i+1] sin[2i] ; out[2i+1] = y[2i+1] cos[2i+1] + y[2i] sin[2i+1]   (fp32 math,
//          embeddings.py:1216-1232 with use_real_unbind_dim = -1)
// ------------------------------------------------------------------------------------------------------------------
struct RopeParts {
  int col_off[4];
  const uint16_t* weight[4];
};

template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(uint16_t* __restrict__ x, int ld, int rows,
                                                           int rows_per_batch, int heads, int D, int parts,
                                                           RopeParts pp, float eps, const float* __restrict__ cosT,
                                                           const float* __restrict__ sinT, int rope_row0, int do_norm) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= rows * parts) return;
  const int row = item / parts, part = item - row * parts;
  const int C = heads * D, nchunks = C >> 3, cph = D >> 3;  // chunks per head
  uint16_t* xr = x + (size_t)row * ld + pp.col_off[part];
  float v[NCH][8];
  float ss[NCH];
  uint4 xq[NCH];                                   // (unconditional clamped loads: see layernorm_kernel)
#pragma unroll
  for (int i = 0; i < NCH; ++i) xq[i] = *(const uint4*)(xr + min(lane + 64 * i, nchunks - 1) * 8);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    ss[i] = 0.f;
    unpack8(xq[i], v[i]);
    if (ch < nchunks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) ss[i] += v[i][e] * v[i][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  if (do_norm == 1) {  // per head: reduce over the cph consecutive lanes of a head
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      for (int o = cph >> 1; o > 0; o >>= 1) ss[i] += __shfl_xor(ss[i], o, 64);
      ss[i] = rsqrtf(ss[i] / (float)D + eps);
    }
  } else if (do_norm == 2) {  // across heads: one statistic for the whole row
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) t += ss[i];
    t = wave_sum(t);
    const float r = rsqrtf(t / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) ss[i] = r;
  }
  const uint16_t* w = pp.weight[part];
  const float* cr = cosT ? cosT + (size_t)(rope_row0 + row % rows_per_batch) * D : nullptr;
  const float* sr = sinT ? sinT + (size_t)(rope_row0 + row % rows_per_batch) * D : nullptr;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch >= nchunks) continue;
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = v[i][e];
    if (do_norm) {
      float wf[8];
      if (w) unpack8(*(const uint4*)(w + (do_norm == 1 ? (ch % cph) * 8 : ch * 8)), wf);
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = bf2f(f2bf(y[e] * ss[i] * (w ? wf[e] : 1.0f)));
    }
    if (cr) {
      const int d0 = (ch % cph) * 8;
      const float4 c0 = *(const float4*)(cr + d0), c1 = *(const float4*)(cr + d0 + 4);
      const float4 s0 = *(const float4*)(sr + d0), s1 = *(const float4*)(sr + d0 + 4);
      const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
      const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        o[e] = y[e] * cc[e] - y[e + 1] * sn[e];
        o[e + 1] = y[e + 1] * cc[e + 1] + y[e] * sn[e + 1];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = o[e];
    }
    *(uint4*)(xr + ch * 8) = pack8(y);
  }
}

// Row softmax: fp32 scores [M][ld] -> bf16 probabilities [M][ldo]; one block per row (N up to 2^20).
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, uint16_t* __restrict__ pr,
                                                           int N, long long ld, long long ldo) {
  __shared__ float red[8];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const float* row = s + (size_t)blockIdx.x * ld;
  uint16_t* orow = pr + (size_t)blockIdx.x * ldo;
  float mx = -3.0e38f;
  for (int i = t * 4; i < N; i += 1024) {
    const float4 v = *(const float4*)(row + i);
    mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
  }
  mx = wave_max(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = t * 4; i < N; i += 1024) {
    const float4 v = *(const float4*)(row + i);
    sum += __expf(v.x - mx) + __expf(v.y - mx) + __expf(v.z - mx) + __expf(v.w - mx);
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + w] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int i = t * 4; i < N; i += 1024) {
    const float4 v = *(const float4*)(row + i);
    uint2 pk;
    pk.x = pack_bf2(__expf(v.x - mx) * inv, __expf(v.y - mx) * inv);
    pk.y = pack_bf2(__expf(v.z - mx) * inv, __expf(v.w - mx) * inv);
    *(uint2*)(orow + i) = pk;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// WanRMS_norm over the channels of channels-last rows (+ SiLU): models/autoencoders/autoencoder_kl_wan.py:198-206 and
// the nonlinearity that follows it at :352-353,:367-368,:899-900.  A group of LPR lanes owns one row (NCH 16-byte chunks
// per lane), 64 / LPR rows per wave; the rounding points are the reference's (normalize in fp32 -> bf16, * sqrt(C) ->
// bf16, * gamma -> bf16, SiLU -> bf16).  Zero-padded channels (gamma 0) stay exactly 0.
// ------------------------------------------------------------------------------------------------------------------
template <int LPR, int NCH>
__global__ __launch_bounds__(256) void rms_channels_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ gamma,
                                                           uint16_t* __restrict__ y, long long rows, int C, float scale,
                                                           int act) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rsel = lane / LPR;
  const int chunks = C >> 3;
  float g[NCH][8];
#pragma unroll
  for (int u = 0; u < NCH; ++u) {
    const int ch = sub + u * LPR;
    if (ch < chunks) {
      unpack8(*(const uint4*)(gamma + (size_t)ch * 8), g[u]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) g[u][e] = 0.f;
    }
  }
  const long long stride = (long long)gridDim.x * 4 * RPW;
  for (long long r = ((long long)blockIdx.x * 4 + wave) * RPW + rsel; r < rows + rsel; r += stride) {
    // (rows + rsel bound keeps every lane of a row group in the loop for the shuffles; out-of-range rows do no I/O)
    const bool live = r < rows;
    float f[NCH][8];
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int ch = sub + u * LPR;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (live && ch < chunks) v = *(const uint4*)(x + (size_t)r * C + (size_t)ch * 8);
      unpack8(v, f[u]);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += f[u][e] * f[u][e];
    }
#pragma unroll
    for (int m = LPR >> 1; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int ch = sub + u * LPR;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = bf2f(f2bf(f[u][e] / nrm));
        v = bf2f(f2bf(v * scale));
        v = bf2f(f2bf(v * g[u][e]));
        o[e] = (act == DA_ACT_SILU) ? silu_f(v) : v;
      }
      if (live && ch < chunks) *(uint4*)(y + (size_t)r * C + (size_t)ch * 8) = pack8(o);
    }
  }
}

template <int LPR, int NCH>
int launch_rms_channels(const void* x, const void* gamma, void* y, long long rows, int C, float scale, int act,
                        hipStream_t s) {
  constexpr int RPW = 64 / LPR;
  long long blocks = (rows + 4 * RPW - 1) / (4 * RPW);
  if (blocks > 8192) blocks = 8192;
  DA_LAUNCH((rms_channels_kernel<LPR, NCH>), dim3((unsigned)blocks), dim3(256), 0, s, (const uint16_t*)x,
            (const uint16_t*)gamma, (uint16_t*)y, rows, C, scale, act);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

struct GnPlan {
  int threads, krows, nblk, pix_per_blk;
};
// Plan knobs (experiments only: DA_GN_THREADS / DA_GN_MINPIX / DA_GN_MAXBLK override the defaults below; read once)
int gn_knob(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
GnPlan gn_plan(int B, int HW, int C) {
  static const int e_threads = gn_knob("DA_GN_THREADS", 0), e_minpix = gn_knob("DA_GN_MINPIX", 0),
                   k_maxblk = gn_knob("DA_GN_MAXBLK", 512), k_cap = gn_knob("DA_GN_CAP", 2048);
  // Measured on the 14 GroupNorm shapes of an SDXL denoising step (profiles/r04c_norms.jsonl; 454 -> 369 us over the 14): from
  // ~4 M elements up 512-thread blocks (wide tensors had ONE pixel row per block: C = 1280 -> 160 threads), from ~10 M up 16
  // pixels per thread row; the small tensors (two ~8 us latency-bound launches) and the VAE's very large ones keep the old plan.
  const long long elems = (long long)B * HW * C;
  const bool mid = elems >= (4ll << 20) && elems <= (48ll << 20);
  const int k_threads = e_threads ? e_threads : (mid ? 512 : 256);
  const int k_minpix = e_minpix ? e_minpix : (mid && elems >= (10ll << 20) ? 16 : 8);
  GnPlan g;
  const int cpr = C / 8;
  int k = k_threads / cpr;
  if (k < 1) k = 1;
  while (k > 1 && cpr * k > 512) --k;
  g.krows = k;
  g.threads = cpr * k;
  int nblk = (HW + k_minpix * k - 1) / (k_minpix * k);  // at least 8 pixels per thread row (two rounds of 4 loads in flight)
  const int cap = k_cap / (B > 0 ? B : 1);
  if (nblk > cap) nblk = cap;
  if (nblk > k_maxblk) nblk = k_maxblk;
  if (nblk < 1) nblk = 1;
  int ppb = (HW + nblk - 1) / nblk;
  ppb = ((ppb + k - 1) / k) * k;
  g.pix_per_blk = ppb;
  g.nblk = (HW + ppb - 1) / ppb;
  return g;
}

}  // namespace

// One-launch plan (gn_fused_kernel): GS = the smallest number of whole groups whose channel span is a multiple of 8, a block size
// that is a multiple of both 64 and the span's chunk count, and the slab either LDS-resident or small enough to be re-read from
// L2 by its one workgroup.  Returns GS (0: keep the two-kernel form).  DA_GN_FUSED = 0 switches it off, DA_GN_FUSED_KB pins the
// largest slab (KiB) that is re-read from L2 instead (default 0: none -- it measured slower), DA_GN_FUSED_MINWG the fewest workgroups
// worth launching.
struct GnFused { int gs, threads, resident; size_t lds; int parts; };
GnFused gn_fused_plan(int B, int HW, int C, int C1, int G, bool have_sync) {
  // (read at every call -- a getenv is noise next to a launch: tests and tools/bench_norms_r5.py time both forms in one process)
  // Measured (profiles/r05c_groupnorm_one_launch.jsonl, chained launches from a HIP graph): LDS-resident slabs win -- 17.4 -> 12.6 us
  // for SDXL's 14 GroupNorms over (2, 1024 px, 1280 ch), 10.4-14.2 -> 4.1-7.0 us for the 8 x 8 / 16 x 16 levels of the SD1.5 U-Net;
  // re-reading a larger slab from L2 by its one workgroup LOSES (40-109 us against 15-30): the default reach is the LDS.
  const int on = gn_knob("DA_GN_FUSED", 1), max_kb = gn_knob("DA_GN_FUSED_KB", 0), min_wg = gn_knob("DA_GN_FUSED_MINWG", 16);
  const int multi = gn_knob("DA_GN_MULTI", 1);
  GnFused f{0, 0, 0, 0, 1};
  if (!on) return f;
  const int cpg = C / G;
  int gs = 1;
  while (gs <= 4 && ((gs * cpg) & 7)) ++gs;
  if (gs > 4 || (G % gs)) return f;
  const int cw = gs * cpg, nch = cw / 8;
  // (two sources: C1 is a multiple of 8, so no 16-byte chunk straddles them; a group SET may -- each thread picks its source by
  // its own chunk)
  int lcm = 64;
  while (lcm % nch) lcm += 64;
  if (lcm > 1024) return f;
  const int threads = (1024 / lcm) * lcm;
  const size_t slab = (size_t)HW * cw * 2;
  const long long wgs = (long long)B * (G / gs);
  if (wgs < min_wg) return f;
  const bool res = slab <= 144 * 1024;
  // (group sets of 3 / 4 groups -- 10 or 30 channels per group -- LOSE in this form: the masked per-group accumulation is 4x the VALU
  //  work per chunk, measured 15.1 -> 18.4 us at (2, 4096 px, 320 ch), profiles/r06_groupnorm_several_workgroups.jsonl)
  if (!res && multi && have_sync && gs <= 2 && wgs <= kGnSyncCounters) {
    // round 6: deal the slab's pixels to a power-of-two number of workgroups, each LDS-resident, all of them resident at once
    // (<= one per CU): as many parts as the chip has CUs for, at least enough for the LDS
    int parts = 2;
    while (parts <= 32 && (size_t)((HW + parts - 1) / parts) * cw * 2 > 144 * 1024) parts *= 2;
    while (parts * 2 <= 32 && wgs * parts * 2 <= 256 && (HW + parts * 2 - 1) / (parts * 2) >= 4 * (threads / nch)) parts *= 2;
    if (parts <= 32 && wgs * parts <= 256) {
      f.gs = gs, f.threads = threads, f.resident = 1, f.parts = parts;
      f.lds = (size_t)((HW + parts - 1) / parts) * cw * 2;
      return f;
    }
  }
  if (!res && slab > (size_t)max_kb * 1024) return f;
  f.gs = gs, f.threads = threads, f.resident = res, f.lds = res ? slab : 0;
  return f;
}

extern "C" size_t da_groupnorm_workspace_bytes(int B, int HW, int C, int G) {
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0) return 0;
  GnPlan g = gn_plan(B, HW, C);
  return (size_t)B * g.nblk * G * 2 * sizeof(float);
}

extern "C" size_t da_groupnorm_sync_bytes(void) { return kGnSyncBytes; }

extern "C" int da_groupnorm_nhwc_bf16(const void* x, const void* x2, int C1, const void* gamma, const void* beta,
                                      void* y, void* workspace, int B, int HW, int C, int G, float eps, int act,
                                      void* sync, void* stream) {
  if (!x || !gamma || !beta || !y || !workspace) return DA_ERR_INVALID;
  if (!x2) C1 = C;
  if (C1 <= 0 || C1 > C || (C1 & 7) || (x2 == nullptr && C1 != C)) return DA_ERR_INVALID;
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > 64 || (C % G) || (C & 7)) return DA_ERR_UNSUPPORTED;
  if (C / 8 > 512) return DA_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const GnFused ff = gn_fused_plan(B, HW, C, C1, G, sync != nullptr);
  if (ff.gs) {
    const dim3 grid(G / ff.gs, B, ff.parts), block(ff.threads);
#define DA_GNF(GS_, RES_)                                                                                              \
  do {                                                                                                                 \
    auto kern = gn_fused_kernel<GS_, RES_>;                                                                            \
    static bool attr_set = false;                                                                                      \
    if (RES_ && !attr_set) {                                                                                           \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) != hipSuccess) \
        return DA_ERR_LAUNCH;                                                                                          \
      attr_set = true;                                                                                                 \
    }                                                                                                                  \
    DA_LAUNCH(kern, grid, block, ff.lds, s, (const uint16_t*)x, (const uint16_t*)x2, C1, (const uint16_t*)gamma,       \
              (const uint16_t*)beta, (uint16_t*)y, HW, C, G, eps, act, (unsigned char*)sync);                          \
  } while (0)
    if (ff.resident) {
      if (ff.gs == 1) DA_GNF(1, true); else if (ff.gs == 2) DA_GNF(2, true); else if (ff.gs == 3) DA_GNF(3, true); else DA_GNF(4, true);
    } else {
      if (ff.gs == 1) DA_GNF(1, false); else if (ff.gs == 2) DA_GNF(2, false); else if (ff.gs == 3) DA_GNF(3, false); else DA_GNF(4, false);
    }
#undef DA_GNF
    DA_CHECK_LAUNCH();
    return DA_OK;
  }
  GnPlan g = gn_plan(B, HW, C);
  const size_t lds = (size_t)g.krows * C * 2 * sizeof(float);
  if (lds > 64 * 1024) return DA_ERR_UNSUPPORTED;
  DA_LAUNCH(gn_stats_kernel, dim3(g.nblk, B), dim3(g.threads), lds, s, (const uint16_t*)x, (const uint16_t*)x2,
                     C1, (float*)workspace, HW, C, G, g.pix_per_blk, g.krows);
  DA_CHECK_LAUNCH();
  DA_LAUNCH(gn_apply_kernel, dim3(g.nblk, B), dim3(g.threads), 0, s, (const uint16_t*)x,
                     (const uint16_t*)x2, C1, (const uint16_t*)gamma, (const uint16_t*)beta, (uint16_t*)y, (const float*)workspace, g.nblk, HW,
                     C, G, eps, act, g.pix_per_blk, g.krows);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// T5LayerNorm: one wave per row, the row in registers (as layernorm_kernel); no mean, no bias, two bf16 roundings.
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ gamma,
                                                           uint16_t* __restrict__ y, int M, int C, int ldx, int ldy, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nchunks = C >> 3;
  float v[NCH][8];
  const uint16_t* xr = x + (size_t)row * ldx;
  float sq = 0.f;
  uint4 xq[NCH];                                   // (unconditional clamped loads: see layernorm_kernel)
#pragma unroll
  for (int i = 0; i < NCH; ++i) xq[i] = *(const uint4*)(xr + min(lane + 64 * i, nchunks - 1) * 8);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    unpack8(xq[i], v[i]);
    if (ch < nchunks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sq += v[i][e] * v[i][e];
    }
  }
  sq = wave_sum(sq);
  const float rstd = rsqrtf(sq / (float)C + eps);
  uint16_t* yr = y + (size_t)row * ldy;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nchunks) {
      float g[8], o[8];
      unpack8(*(const uint4*)(gamma + ch * 8), g);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = g[e] * bf2f(f2bf(v[i][e] * rstd));
      *(uint4*)(yr + ch * 8) = pack8(o);
    }
  }
}

extern "C" int da_rmsnorm_bf16(const void* x, const void* gamma, void* y, int M, int C, int ldx, int ldy, float eps,
                               void* stream) {
  if (!x || !gamma || !y) return DA_ERR_INVALID;
  if (M <= 0 || C <= 0 || (C & 7) || (ldx & 7) || (ldy & 7)) return DA_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int nch = (C / 8 + 63) / 64;
  dim3 grid((M + 3) / 4), block(256);
#define DA_RMS(N) \
  DA_LAUNCH(rmsnorm_rows_kernel<N>, grid, block, 0, s, (const uint16_t*)x, (const uint16_t*)gamma, (uint16_t*)y, M, C, ldx, ldy, eps)
  if (nch <= 1) DA_RMS(1);
  else if (nch <= 2) DA_RMS(2);
  else if (nch <= 4) DA_RMS(4);
  else if (nch <= 8) DA_RMS(8);
  else return DA_ERR_UNSUPPORTED;
#undef DA_RMS
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_layernorm_bf16(const void* x, const void* gamma, const void* beta, void* y, const void* mod_scale,
                                 const void* mod_shift, int mod_ld, int mod_f32, int rows_per_batch, int M, int C,
                                 int ldx, int ldy, float eps, void* stream) {
  if (!x || !y) return DA_ERR_INVALID;
  if (M <= 0 || C <= 0 || (C & 7) || (ldx & 7) || (ldy & 7)) return DA_ERR_UNSUPPORTED;
  if ((mod_scale == nullptr) != (mod_shift == nullptr)) return DA_ERR_INVALID;
  if (mod_scale && (rows_per_batch <= 0 || (mod_ld & 7))) return DA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  const int nch = (C / 8 + 63) / 64;
  dim3 grid((M + 3) / 4), block(256);
#define DA_LN(N)                                                                                                  \
  DA_LAUNCH(layernorm_kernel<N>, grid, block, 0, s, (const uint16_t*)x, (const uint16_t*)gamma,          \
                     (const uint16_t*)beta, (uint16_t*)y, (const uint16_t*)mod_scale, (const uint16_t*)mod_shift, \
                     mod_ld, mod_f32, rows_per_batch, M, C, ldx, ldy, eps)
  if (nch <= 1) DA_LN(1);
  else if (nch <= 2) DA_LN(2);
  else if (nch <= 3) DA_LN(3);
  else if (nch <= 4) DA_LN(4);
  else if (nch <= 6) DA_LN(6);
  else if (nch <= 8) DA_LN(8);
  else return DA_ERR_UNSUPPORTED;
#undef DA_LN
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_rmsnorm_rope_bf16(void* x, int ld, int rows, int rows_per_batch, int heads, int D, int parts,
                                    const int* col_off, const void* const* weight, float eps, const float* cosT,
                                    const float* sinT, int rope_row0, int do_norm, void* stream) {
  if (!x || !col_off || rows <= 0 || heads <= 0 || parts <= 0 || parts > 4) return DA_ERR_INVALID;
  if ((cosT == nullptr) != (sinT == nullptr)) return DA_ERR_INVALID;
  if (D != 64 && D != 128) return DA_ERR_UNSUPPORTED;
  if ((ld & 7) || do_norm < 0 || do_norm > 2) return DA_ERR_UNSUPPORTED;
  if (rows_per_batch <= 0) rows_per_batch = rows;
  RopeParts pp;
  for (int j = 0; j < 4; ++j) {
    pp.col_off[j] = j < parts ? col_off[j] : 0;
    pp.weight[j] = (j < parts && weight) ? (const uint16_t*)weight[j] : nullptr;
    if (j < parts && (col_off[j] & 7)) return DA_ERR_UNSUPPORTED;
  }
  const int nch = (heads * D / 8 + 63) / 64;
  dim3 grid((rows * parts + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
#define DA_RR(N)                                                                                                   \
  DA_LAUNCH(rmsnorm_rope_kernel<N>, grid, block, 0, s, (uint16_t*)x, ld, rows, rows_per_batch, heads, D, parts, pp, \
            eps, cosT, sinT, rope_row0, do_norm)
  if (nch <= 1) DA_RR(1);
  else if (nch <= 2) DA_RR(2);
  else if (nch <= 3) DA_RR(3);
  else if (nch <= 4) DA_RR(4);
  else if (nch <= 6) DA_RR(6);
  else if (nch <= 8) DA_RR(8);
  else return DA_ERR_UNSUPPORTED;
#undef DA_RR
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_softmax_rows_f32_bf16(const void* scores, void* probs, int M, int N, long long ld, long long ldo,
                                        void* stream) {
  if (!scores || !probs || M <= 0 || N <= 0 || (N & 3) || (ld & 3) || (ldo & 3)) return DA_ERR_INVALID;
  DA_LAUNCH(softmax_rows_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const float*)scores,
                     (uint16_t*)probs, N, ld, ldo);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_rmsnorm_channels_bf16(const void* x, const void* gamma, void* y, long long rows, int C, float scale,
                                        int act, void* stream) {
  if (!x || !gamma || !y || rows <= 0 || C <= 0 || (C & 7)) return DA_ERR_INVALID;
  if (act != DA_ACT_NONE && act != DA_ACT_SILU) return DA_ERR_UNSUPPORTED;
  if (C > 1024) return DA_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int chunks = C >> 3;
  if (chunks <= 8) return launch_rms_channels<8, 1>(x, gamma, y, rows, C, scale, act, s);
  if (chunks <= 16) return launch_rms_channels<16, 1>(x, gamma, y, rows, C, scale, act, s);
  if (chunks <= 32) return launch_rms_channels<32, 1>(x, gamma, y, rows, C, scale, act, s);
  if (chunks <= 64) return launch_rms_channels<64, 1>(x, gamma, y, rows, C, scale, act, s);
  return launch_rms_channels<64, 2>(x, gamma, y, rows, C, scale, act, s);
}
