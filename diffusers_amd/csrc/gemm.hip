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
// Staging is either register-staged (global_load_dwordx4 -> ds_write_b128) or direct-to-LDS
// (global_load_lds_dwordx4, swizzle applied on the per-lane SOURCE address, LDS image lane-linear).
#include "common.cuh"
#include "diffusers_amd.h"

namespace {

__device__ uint4 g_zero_line[8];  // 128 B of zeros: source for out-of-bounds rows in the direct-to-LDS path

struct RowInfo {      // per staged activation row (implicit GEMM gather state)
  int base;           // linear: row index (or -1 if out of range); conv: b*Hin
  int oy, ox;         // conv: oy*stride - pad, ox*stride - pad
};

// WM x WN waves, each wave owns MT x NT MFMA tiles of 32x32  ->  block tile (32*MT*WM) x (32*NT*WN), K slices of 64.
// STAGES = LDS ring depth: 2 = prefetch distance 1 (one slice in flight under the MFMAs of the current one),
// 3 = prefetch distance 2 with a COUNTED s_waitcnt vmcnt so one slice stays in flight across every barrier.
template <int WM, int WN, int MT, int NT, int STAGES, bool CONV, bool GLDS>
__global__ __launch_bounds__(64 * WM * WN) void igemm_bf16_kernel(const da_gemm_params p) {
  constexpr int NW = WM * WN, NTHR = 64 * NW, RP = NTHR / 8;  // RP = tile rows staged per pass (one 1 KiB piece per wave)
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  constexpr int XR = BM / RP, WR = BN / RP;  // staged rows per thread
  constexpr int XBYTES = BM * 128, WBYTES = BN * 128, STAGE = XBYTES + WBYTES;
  constexpr int PD = STAGES - 1;             // prefetch distance
  constexpr int LOADS = XR + WR;             // LDS-DMA instructions per wave per K slice
  static_assert(XR >= 1 && XR <= 4 && WR >= 1 && WR <= 4, "tile / thread-count combination not stageable");
  static_assert(GLDS || (STAGES == 2 && NW == 4), "register staging exists for the 4-wave 2-stage tiles only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int l31 = lane & 31, hi = lane >> 5;

  // ---- XCD-aware tile mapping: block b runs on XCD b%8; give each XCD a contiguous run of tiles ----
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
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

  const int nk = p.K >> 6;
  const int Ctot = CONV ? (p.C1 + p.C2) : 0;

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 xg0, xg1, xg2, xg3, wg0, wg1, wg2, wg3;  // named (not arrays) so they never land in scratch
  xg0 = xg1 = xg2 = xg3 = wg0 = wg1 = wg2 = wg3 = make_uint4(0, 0, 0, 0);

  // K-slice cursor of the NEXT slice to issue.  Slices are issued strictly in order, so the (tap, channel) position
  // of the implicit GEMM advances incrementally (no integer division in the loop): K index = tap * Ctot + c.
  int is_kh = 0, is_kw = 0, is_c0 = 0;  // conv: kernel row / column of the tap, first channel of the slice
  size_t is_k = 0;                      // linear / weights: element offset of the slice inside a row

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
    if (GLDS) {                                                                                                        \
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
    /* advance the cursor to the next K slice */                                                                       \
    is_k += 64;                                                                                                        \
    if (CONV) {                                                                                                        \
      is_c0 += 64;                                                                                                     \
      if (is_c0 >= Ctot) {                                                                                             \
        is_c0 = 0;                                                                                                     \
        if (++is_kw >= p.conv) {                                                                                       \
          is_kw = 0;                                                                                                   \
          ++is_kh;                                                                                                     \
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
  // Wait until at most PENDING later slices of THIS wave's LDS-DMA are still in flight, then rendezvous.  The raw
  // s_barrier (not __syncthreads, whose fence would drain vmcnt to 0) lets one slice stay in flight across the barrier;
  // the asm "memory" clobbers keep the compiler from moving LDS accesses across the rendezvous.
#define DA_STAGE_WAIT(PENDING)                                                                                         \
  do {                                                                                                                 \
    if (GLDS) {                                                                                                        \
      if (PENDING) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LOADS) : "memory");                             \
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                 \
      __builtin_amdgcn_s_barrier();                                                                                    \
      asm volatile("" ::: "memory");                                                                                   \
    } else {                                                                                                           \
      __syncthreads();                                                                                                 \
    }                                                                                                                  \
  } while (0)

  // fragment read offsets (bytes) inside a tile: row (l31) * 128 + ((2*ks+hi) ^ ((l31>>1)&7)) * 16
  const int fsw = (l31 >> 1) & 7;
  const int frow = l31 * 128;

  auto compute = [&](int buf) {
    const unsigned char* xb = smem + buf * STAGE + (wm * MT * 32) * 128 + frow;
    const unsigned char* wb = smem + buf * STAGE + XBYTES + (wn * NT * 32) * 128 + frow;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int off = ((2 * ks + hi) ^ fsw) << 4;
      bf16x8_t wf[NT], xf[MT];
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = *(const bf16x8_t*)(wb + j * 32 * 128 + off);
#pragma unroll
      for (int i = 0; i < MT; ++i) xf[i] = *(const bf16x8_t*)(xb + i * 32 * 128 + off);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    }
  };

  // ---- main loop: LDS ring of STAGES slices, one rendezvous per K slice ----
  DA_STAGE_ISSUE(0);
  DA_STAGE_COMMIT(0);
  if (PD == 2 && nk > 1) DA_STAGE_ISSUE(1);
  if (PD == 2 && nk > 1) DA_STAGE_WAIT(1);
  else DA_STAGE_WAIT(0);
  int cur = 0;                                   // ring slot of slice kt
  int nxt = (PD == 2) ? 2 : 1;                   // ring slot the next issued slice goes to
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + PD < nk);
    if (more) DA_STAGE_ISSUE(nxt);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch in flight under this slice's MFMAs
    compute(cur);
    if (kt + 1 < nk) {
      if (more) DA_STAGE_COMMIT(nxt);
      if (PD == 2 && more) DA_STAGE_WAIT(1);
      else DA_STAGE_WAIT(0);
    }
    cur = (cur + 1 == STAGES) ? 0 : cur + 1;
    nxt = (nxt + 1 == STAGES) ? 0 : nxt + 1;
  }
#undef DA_STAGE_ISSUE
#undef DA_STAGE_COMMIT
#undef DA_STAGE_WAIT

  // ---- epilogue: lane holds, for output row m (= lane&31 within the 32-tile), channels 8*(r>>2)+4*hi+(r&3) ----
  const uint16_t* __restrict__ bias = (const uint16_t*)p.bias;
  const uint16_t* __restrict__ rowvec = (const uint16_t*)p.rowvec;
  const uint16_t* __restrict__ resid = (const uint16_t*)p.residual;
  const bool geglu = (p.act == DA_ACT_GEGLU);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + (wm * MT + i) * 32 + l31;
    if (m >= p.M) continue;
    const int bidx = (rowvec != nullptr) ? (m / p.rows_per_batch) : 0;
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
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float hv = acc[i][2 * jp][4 * g + e] * p.alpha;
              float gv = acc[i][2 * jp + 1][4 * g + e] * p.alpha;
              if (bias) {
                hv += bf2f(bias[nv + e]);
                gv += bf2f(bias[nv + 32 + e]);
              }
              // reference rounds the projection to bf16 before chunk/gelu/mul (activations.py:113-124)
              hv = bf2f(f2bf(hv));
              gv = bf2f(f2bf(gv));
              o[e] = hv * bf2f(f2bf(gelu_erf_f(gv)));
            }
            uint2 pk;
            pk.x = pack_bf2(o[0], o[1]);
            pk.y = pack_bf2(o[2], o[3]);
            *(uint2*)((uint16_t*)p.C + (size_t)m * p.ldc + no) = pk;
          }
        }
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + (wn * NT + j) * 32 + 8 * g + 4 * hi;
        if (n >= p.N) continue;  // N is a multiple of 4 (checked on the host)
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[i][j][4 * g + e] * p.alpha;
        if (bias) {
          const uint2 bv = *(const uint2*)(bias + n);
          o[0] += bf_lo(bv.x); o[1] += bf_hi(bv.x); o[2] += bf_lo(bv.y); o[3] += bf_hi(bv.y);
        }
        if (rowvec) {
          const uint2 rv = *(const uint2*)(rowvec + (size_t)bidx * p.ld_rowvec + n);
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
        }
        if (resid) {
          const uint2 rv = *(const uint2*)(resid + (size_t)m * p.ldr + n);
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
        }
      }
    }
  }
}

struct TileShape {
  int bm, bn, waves;
};
constexpr TileShape kTiles[] = {{0, 0, 0},      {128, 128, 4}, {64, 128, 4},  {128, 64, 4},
                                {64, 64, 4},    {256, 128, 8}, {128, 256, 8}, {256, 256, 8}};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

template <int WM, int WN, int MT, int NT, int STAGES, bool CONV, bool GLDS>
int launch(const da_gemm_params& p, hipStream_t s) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const size_t lds = (size_t)(BM + BN) * 128 * STAGES;
  auto kern = igemm_bf16_kernel<WM, WN, MT, NT, STAGES, CONV, GLDS>;
  if (lds > 48 * 1024) {
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DA_ERR_LAUNCH;
      attr_set = true;
    }
  }
  DA_LAUNCH(kern, dim3(tiles), dim3(64 * WM * WN), lds, s, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

template <bool CONV>
int dispatch(const da_gemm_params& p, int tile, int staging, hipStream_t s) {
  if (staging == DA_STAGE_REGISTER) {
    switch (tile) {
      case DA_TILE_128x128: return launch<2, 2, 2, 2, 2, CONV, false>(p, s);
      case DA_TILE_64x128: return launch<2, 2, 1, 2, 2, CONV, false>(p, s);
      case DA_TILE_128x64: return launch<2, 2, 2, 1, 2, CONV, false>(p, s);
      case DA_TILE_64x64: return launch<2, 2, 1, 1, 2, CONV, false>(p, s);
    }
    return DA_ERR_UNSUPPORTED;
  }
  if (staging == DA_STAGE_LDS_DIRECT) {
    switch (tile) {
      case DA_TILE_128x128: return launch<2, 2, 2, 2, 2, CONV, true>(p, s);
      case DA_TILE_64x128: return launch<2, 2, 1, 2, 2, CONV, true>(p, s);
      case DA_TILE_128x64: return launch<2, 2, 2, 1, 2, CONV, true>(p, s);
      case DA_TILE_64x64: return launch<2, 2, 1, 1, 2, CONV, true>(p, s);
      case DA_TILE_256x128: return launch<4, 2, 2, 2, 2, CONV, true>(p, s);
      case DA_TILE_128x256: return launch<2, 4, 2, 2, 2, CONV, true>(p, s);
      case DA_TILE_256x256: return launch<2, 4, 4, 2, 2, CONV, true>(p, s);
    }
    return DA_ERR_UNSUPPORTED;
  }
  if (staging == DA_STAGE_LDS_DIRECT3) {
    switch (tile) {
      case DA_TILE_128x128: return launch<2, 2, 2, 2, 3, CONV, true>(p, s);
      case DA_TILE_64x128: return launch<2, 2, 1, 2, 3, CONV, true>(p, s);
      case DA_TILE_128x64: return launch<2, 2, 2, 1, 3, CONV, true>(p, s);
      case DA_TILE_64x64: return launch<2, 2, 1, 1, 3, CONV, true>(p, s);
      case DA_TILE_256x128: return launch<4, 2, 2, 2, 3, CONV, true>(p, s);
      case DA_TILE_128x256: return launch<2, 4, 2, 2, 3, CONV, true>(p, s);
    }
    return DA_ERR_UNSUPPORTED;  // 256x256 x 3 stages would need 192 KiB of LDS
  }
  return DA_ERR_INVALID;
}

// Untuned fallback: fewest bytes staged per flop among the tiles that still give every CU a block; output-channel
// counts that are a multiple of 64 but not of 128 (320, 960, 1920) take 64-wide tiles so no MFMA column is wasted.
int pick_tile(const da_gemm_params& p) {
  auto nblk = [&](int t) {
    return (long)((p.M + kTiles[t].bm - 1) / kTiles[t].bm) * ((p.N + kTiles[t].bn - 1) / kTiles[t].bn);
  };
  const bool geglu = (p.act == DA_ACT_GEGLU);
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
  if (p.act == DA_ACT_GEGLU && ((p.N & 127) || p.out_f32 || p.residual || p.rowvec)) return DA_ERR_UNSUPPORTED;
  return DA_OK;
}

bool tile_ok(const da_gemm_params& p, int tile) {
  if (tile <= 0 || tile >= kNumTiles) return false;
  // GEGLU pairs (value, gate) 32-column tiles inside one wave: the wave must own an even number of them
  if (p.act == DA_ACT_GEGLU && (tile == DA_TILE_128x64 || tile == DA_TILE_64x64)) return false;
  return true;
}

int run(const da_gemm_params& p, int tile, int staging, hipStream_t s) {
  return p.conv ? dispatch<true>(p, tile, staging, s) : dispatch<false>(p, tile, staging, s);
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
  return run(p, tile, p.staging, (hipStream_t)stream);
}

// Times every (tile, staging) variant that can run this problem on `stream` (HIP events, min of `iters` launches each
// after one warm launch) and returns the fastest.  All variants walk K in the same order with the same MFMA, so they
// produce bit-identical C: tuning changes speed only.  Must not be called while the stream is being captured.
extern "C" int da_gemm_tune(const da_gemm_params* pp, void* stream, int iters, int* best_tile, int* best_staging,
                            float* best_us) {
  if (!pp || !best_tile || !best_staging) return DA_ERR_INVALID;
  da_gemm_params p = *pp;
  const int v = validate(p);
  if (v != DA_OK) return v;
  if (iters <= 0) iters = 3;
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return DA_ERR_LAUNCH;
  float best = 3.0e38f;
  int bt = 0, bs = 0;
  static const int stagings[2] = {DA_STAGE_LDS_DIRECT, DA_STAGE_LDS_DIRECT3};
  for (int tile = 1; tile < kNumTiles; ++tile) {
    if (!tile_ok(p, tile)) continue;
    // a tile more than twice the problem in either dimension only wastes MFMA rows
    if (kTiles[tile].bm >= 2 * p.M + 64 || kTiles[tile].bn >= 2 * p.N + 64) continue;
    for (int si = 0; si < 2; ++si) {
      const int st = stagings[si];
      int rc = run(p, tile, st, s);  // warm launch (also sets the LDS attribute once)
      if (rc == DA_ERR_UNSUPPORTED) continue;
      if (rc != DA_OK) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
      float tmin = 3.0e38f;
      for (int it = 0; it < iters; ++it) {
        (void)hipEventRecord(e0, s);
        rc = run(p, tile, st, s);
        (void)hipEventRecord(e1, s);
        if (rc != DA_OK || hipEventSynchronize(e1) != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return DA_ERR_LAUNCH; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < tmin) tmin = ms;
      }
      if (tmin < best) { best = tmin; bt = tile; bs = st; }
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (bt == 0) return DA_ERR_UNSUPPORTED;
  *best_tile = bt;
  *best_staging = bs;
  if (best_us) *best_us = best * 1000.0f;
  return DA_OK;
}
