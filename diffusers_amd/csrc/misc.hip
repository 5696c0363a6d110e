// Small hot-path kernels that are not GEMM-shaped: sinusoidal timestep embedding, skinny (M <= 8) linear layers for
// the embedding MLPs, and the two "thin" convolutions at the ends of the U-Net / VAE (Cin <= 16 or Cout <= 16) which
// also convert between the reference's NCHW tensors and the engine's channels-last layout.
//
// Reference (diffusers src/diffusers/):
//   get_timestep_embedding       models/embeddings.py:27-78 (+ Timesteps :1310-1326)
//   TimestepEmbedding            models/embeddings.py:1262-1308   (Linear -> SiLU -> Linear, M = batch)
//   time_emb_proj                models/resnet.py:345-349         (Linear(SiLU(temb)))
//   conv_in / conv_out           models/unets/unet_2d_condition.py:1108,:1230 ; models/autoencoders/vae.py:286,:309
//   post_quant_conv              models/autoencoders/autoencoder_kl.py:204
#include <cstdlib>

#include "common.cuh"

namespace {

// out[b][:] = [sin | cos] (or [cos | sin] when flip) of t_b * exp(-ln(max_period) * i / (half - shift)), fp32 math.
__global__ void timestep_embedding_kernel(const float* __restrict__ t, const float* __restrict__ table,
                                          const int* __restrict__ step_idx, void* __restrict__ out, int B, int dim,
                                          int flip_sin_to_cos, float shift, float scale, float max_period, int out_f32) {
  const int half = dim / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * half) return;
  const int b = idx / half, i = idx - b * half;
  float tv;
  if (table) tv = table[(size_t)(*step_idx) * 8 + 7];
  else tv = t[b];
  const float exponent = (-logf(max_period) * (float)i) / ((float)half - shift);
  const float arg = scale * (tv * expf(exponent));
  const float sv = sinf(arg), cv = cosf(arg);
  const size_t o0 = (size_t)b * dim + (flip_sin_to_cos ? half + i : i);  // sin slot
  const size_t o1 = (size_t)b * dim + (flip_sin_to_cos ? i : half + i);  // cos slot
  if (out_f32) {
    ((float*)out)[o0] = sv;
    ((float*)out)[o1] = cv;
  } else {
    ((uint16_t*)out)[o0] = f2bf(sv);
    ((uint16_t*)out)[o1] = f2bf(cv);
  }
}

// Skinny linear: out[m][n] = act_out( sum_k act_in(x[m][k]) * W[n][k] + bias[n] ) (+ res[m][n]); M <= 8.
// One wave per output column; lanes stride K with 16-byte loads; W is streamed once (HBM-bound GEMV regime).
template <int MMAX>
__global__ __launch_bounds__(256) void linear_small_m_kernel(const uint16_t* __restrict__ x,
                                                             const uint16_t* __restrict__ W,
                                                             const uint16_t* __restrict__ bias,
                                                             const uint16_t* __restrict__ res,
                                                             uint16_t* __restrict__ out, int M, int N, int K, int ldx,
                                                             int ldo, int ldr, int act_in, int act_out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc[MMAX];
#pragma unroll
  for (int m = 0; m < MMAX; ++m) acc[m] = 0.f;
  const uint16_t* wr = W + (size_t)n * K;
  // Four K chunks per trip: the weight row is met cold (a GEMV streams W once), and with one 16-byte load in flight per lane the
  // launch was K / 512 serialised HBM round trips (12.7 us for the 7 MB of SDXL's time-embedding layers).  The loads are issued
  // unconditionally from clamped offsets (chunks past K / rows past M are never accumulated / never stored); the accumulation order
  // per lane -- chunks in k order -- is unchanged, so the results are bit-identical.
  for (int k0 = lane * 8; k0 < K; k0 += 4 * 512) {
    uint4 wq[4], xq[4][MMAX];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kc = min(k0 + u * 512, K - 8);
      wq[u] = *(const uint4*)(wr + kc);
#pragma unroll
      for (int m = 0; m < MMAX; ++m) xq[u][m] = *(const uint4*)(x + (size_t)min(m, M - 1) * ldx + kc);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (k0 + u * 512 >= K) break;
      float wf[8];
      unpack8(wq[u], wf);
#pragma unroll
      for (int m = 0; m < MMAX; ++m) {
        float xf[8];
        unpack8(xq[u][m], xf);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float xv = xf[e];
          if (act_in == DA_ACT_SILU) xv = bf2f(f2bf(silu_f(xv)));
          acc[m] += xv * wf[e];
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MMAX; ++m) acc[m] = wave_sum(acc[m]);
  if (lane == 0) {
    const float bv = bias ? bf2f(bias[n]) : 0.f;
    for (int m = 0; m < M; ++m) {
      float o = acc[m] + bv;
      if (act_out == DA_ACT_SILU) o = silu_f(bf2f(f2bf(o)));
      else if (act_out == DA_ACT_GELU_TANH) o = gelu_tanh_f(bf2f(f2bf(o)));
      if (res) o = bf2f(f2bf(o)) + bf2f(res[(size_t)m * ldr + n]);
      out[(size_t)m * ldo + n] = f2bf(o);
    }
  }
}

// Thin-input conv: Cin <= 16, k in {1,3}, stride 1, pad (k-1)/2.  Input NCHW or NHWC, output NHWC [B][H][W][Cout].
// Thread = (pixel, 8 output channels).  weights: [Cout][k][k][Cin] bf16.  blockIdx.y selects a chunk of `coc` output
// channels whose weights are staged in LDS as float, k-major ([k*k*Cin][coc]): the 8 channels a thread owns are 32
// contiguous bytes (two ds_read_b128) and the lanes of a wave (consecutive channel groups of one pixel) read
// consecutive addresses instead of one bank.
__global__ __launch_bounds__(256) void conv_thin_in_kernel(const uint16_t* __restrict__ x,
                                                           const uint16_t* __restrict__ w,
                                                           const uint16_t* __restrict__ bias,
                                                           uint16_t* __restrict__ y, int B, int H, int W, int Cin,
                                                           int Cout, int ks, int in_nchw, float in_div, float in_add,
                                                           int coc) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];  // [ks*ks*Cin][coc]
  const int kk = ks * ks * Cin;
  const int co0 = blockIdx.y * coc;
  const int ncoc = min(coc, Cout - co0);
  for (int i = threadIdx.x; i < ncoc * kk; i += blockDim.x) {
    const int co = i / kk, k = i - co * kk;
    wsm[k * coc + co] = bf2f(w[(size_t)(co0 + co) * kk + k]);
  }
  __syncthreads();
  const int cgroups = ncoc >> 3;
  const size_t total = (size_t)B * H * W * cgroups;
  const int pad = (ks - 1) / 2;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cgroups);
    const size_t pix = idx / cgroups;
    const int xw = (int)(pix % W);
    const int yh = (int)((pix / W) % H);
    const int b = (int)(pix / ((size_t)W * H));
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bias ? bf2f(bias[co0 + cg * 8 + e]) : 0.f;
    for (int kh = 0; kh < ks; ++kh) {
      const int iy = yh + kh - pad;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kw = 0; kw < ks; ++kw) {
        const int ix = xw + kw - pad;
        if ((unsigned)ix >= (unsigned)W) continue;
        for (int c = 0; c < Cin; ++c) {
          const size_t off = in_nchw ? (((size_t)b * Cin + c) * H + iy) * W + ix : (((size_t)b * H + iy) * W + ix) * Cin + c;
          float xv = bf2f(x[off]);
          if (in_div != 1.0f) xv = bf2f(f2bf(__fdiv_rn(xv, in_div)));  // e.g. latents / vae.config.scaling_factor
          if (in_add != 0.0f) xv = bf2f(f2bf(__fadd_rn(xv, in_add)));  // + vae.config.shift_factor (Flux)
          const float* wp = wsm + (size_t)((kh * ks + kw) * Cin + c) * coc + cg * 8;
          const float4 w0 = *(const float4*)wp, w1 = *(const float4*)(wp + 4);
          acc[0] += xv * w0.x; acc[1] += xv * w0.y; acc[2] += xv * w0.z; acc[3] += xv * w0.w;
          acc[4] += xv * w1.x; acc[5] += xv * w1.y; acc[6] += xv * w1.z; acc[7] += xv * w1.w;
        }
      }
    }
    *(uint4*)(y + pix * Cout + co0 + cg * 8) = pack8(acc);
  }
}

// Round 6: the same conv with FOUR consecutive output pixels of a row per thread (3 x 3, Cin <= 4: U-Net conv_in 4 -> 320, the DDPM
// U-Net's 3 -> 128).  The kernel above reads the two 16-byte weight vectors of every (tap, input channel) from LDS and ONE input
// value from memory per 8 multiply-adds; here the weight vectors feed 32 multiply-adds (four pixels) and a kernel row's six input
// columns are loaded once for the four pixels that share them.  Per output element the operations and their order are the ones of
// the kernel above (bias, then taps in (kh, kw, c) order, the same input conversions): bit-identical.
template <int CIN>
__global__ __launch_bounds__(256) void conv_thin_in4_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                            const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int B, int H,
                                                            int W, int Cout, int in_nchw, float in_div, float in_add, int coc) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];  // [9 * CIN][coc]
  constexpr int kk = 9 * CIN;
  const int co0 = blockIdx.y * coc;
  const int ncoc = min(coc, Cout - co0);
  // (consecutive lanes take consecutive OUTPUT CHANNELS of one (tap, input channel): their LDS words are consecutive.  With consecutive k
  // per lane -- the coalesced order of the global side -- the 64 lanes of a store hit ONE bank, coc * 4 bytes being a multiple of 256)
  for (int i = threadIdx.x; i < ncoc * kk; i += blockDim.x) {
    const int k = i / ncoc, co = i - k * ncoc;
    wsm[k * coc + co] = bf2f(w[(size_t)(co0 + co) * kk + k]);
  }
  __syncthreads();
  const int cgroups = ncoc >> 3;
  const size_t sB = (size_t)CIN * H * W, sC = in_nchw ? (size_t)H * W : 1, sY = in_nchw ? (size_t)W : (size_t)W * CIN, sX = in_nchw ? 1 : CIN;
  const bool convert_in = (in_div != 1.0f) || (in_add != 0.0f);
  const int wq = (W + 3) >> 2;                                 // pixel quads per row
  const size_t total = (size_t)B * H * wq * cgroups;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cgroups);
    const size_t quad = idx / cgroups;
    const int x0 = (int)(quad % wq) * 4;
    const int yh = (int)((quad / wq) % H);
    const int b = (int)(quad / ((size_t)wq * H));
    float acc[4][8];
    {
      float bvs[8];
      // (one 16-byte load when the eight channels' bias is 16-byte aligned; eight dependent 2-byte loads each waited for otherwise)
      if (bias && (((size_t)(bias + co0 + cg * 8)) & 15) == 0) {
        unpack8(*(const uint4*)(bias + co0 + cg * 8), bvs);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bvs[e] = bias ? bf2f(bias[co0 + cg * 8 + e]) : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int px = 0; px < 4; ++px) acc[px][e] = bvs[e];
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = yh + kh - 1;
      const bool row_ok = (unsigned)iy < (unsigned)H;          // (a padded row adds nothing: skipped in the kernel above as well)
      float xin[6][CIN];
      // The 6 * CIN input values of this kernel row are loaded UNCONDITIONALLY from clamped (always valid) addresses and zeroed by a
      // select afterwards: behind `if (ok)` every load sat in its own exec-masked branch with its own s_waitcnt -- 72 serialised round
      // trips per thread, which WAS the kernel (U-Net conv_in 4 -> 320 at 128 x 128: 58 us for 21 MB of output; round 6).  Same values.
      // (layout strides instead of a per-element NCHW / NHWC select, and ONE uniform branch around the conversions: straight-line loads)
      uint16_t raw[6][CIN];
      const int iyc = min(max(iy, 0), H - 1);
      const uint16_t* xrow = x + (size_t)b * sB + (size_t)iyc * sY;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int ixc = min(max(x0 + j - 1, 0), W - 1);
#pragma unroll
        for (int c = 0; c < CIN; ++c) raw[j][c] = xrow[(size_t)ixc * sX + (size_t)c * sC];
      }
      if (convert_in) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const bool ok = row_ok && (unsigned)(x0 + j - 1) < (unsigned)W;
#pragma unroll
          for (int c = 0; c < CIN; ++c) {
            float xv = bf2f(raw[j][c]);
            if (in_div != 1.0f) xv = bf2f(f2bf(__fdiv_rn(xv, in_div)));
            if (in_add != 0.0f) xv = bf2f(f2bf(__fadd_rn(xv, in_add)));
            xin[j][c] = ok ? xv : 0.f;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const bool ok = row_ok && (unsigned)(x0 + j - 1) < (unsigned)W;
#pragma unroll
          for (int c = 0; c < CIN; ++c) xin[j][c] = ok ? bf2f(raw[j][c]) : 0.f;
        }
      }
      if (!row_ok) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
          const float* wp = wsm + (size_t)((kh * 3 + kw) * CIN + c) * coc + cg * 8;
          const float4 w0 = *(const float4*)wp, w1 = *(const float4*)(wp + 4);
#pragma unroll
          for (int px = 0; px < 4; ++px) {
            // a tap that falls into the left / right padding is SKIPPED above (not multiplied by zero): keep that -- x * w with
            // x = 0 is exact, but skipping and adding + 0.0f can differ in the sign of a zero sum; the predicate keeps the bits
            const int ix = x0 + px + kw - 1;
            if ((unsigned)ix >= (unsigned)W) continue;
            const float xv = xin[px + kw][c];
            acc[px][0] += xv * w0.x; acc[px][1] += xv * w0.y; acc[px][2] += xv * w0.z; acc[px][3] += xv * w0.w;
            acc[px][4] += xv * w1.x; acc[px][5] += xv * w1.y; acc[px][6] += xv * w1.z; acc[px][7] += xv * w1.w;
          }
        }
    }
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      if (x0 + px >= W) break;
      const size_t pix = ((size_t)b * H + yh) * W + x0 + px;
      *(uint4*)(y + pix * Cout + co0 + cg * 8) = pack8(acc[px]);
    }
  }
}

// Thin-output conv 3x3 / pad 1: Cout <= 16, input NHWC [B][H][W][Cin] (Cin % 8 == 0), output NCHW (bf16 or fp32).
// Thread = one output pixel, all Cout channels; weights [Cout][3][3][Cin] staged in LDS as bf16.
template <int COUT>
__global__ __launch_bounds__(256) void conv_thin_out_kernel(const uint16_t* __restrict__ x,
                                                            const uint16_t* __restrict__ w,
                                                            const uint16_t* __restrict__ bias, void* __restrict__ y,
                                                            int B, int H, int W, int Cin, int out_f32) {
  extern __shared__ __attribute__((aligned(16))) uint16_t wsh[];  // [COUT][9*Cin]
  const int kk = 9 * Cin;
  for (int i = threadIdx.x * 8; i < COUT * kk; i += blockDim.x * 8) *(uint4*)(wsh + i) = *(const uint4*)(w + i);
  __syncthreads();
  const size_t total = (size_t)B * H * W;
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (size_t)gridDim.x * blockDim.x) {
    const int xw = (int)(pix % W);
    const int yh = (int)((pix / W) % H);
    const int b = (int)(pix / ((size_t)W * H));
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = bias ? bf2f(bias[o]) : 0.f;
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = yh + kh - 1;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = xw + kw - 1;
        if ((unsigned)ix >= (unsigned)W) continue;
        const uint16_t* xp = x + (((size_t)b * H + iy) * W + ix) * Cin;
        const uint16_t* wp = wsh + (kh * 3 + kw) * Cin;
        for (int c = 0; c < Cin; c += 8) {
          float xf[8];
          unpack8(*(const uint4*)(xp + c), xf);
#pragma unroll
          for (int o = 0; o < COUT; ++o) {
            float wf[8];
            unpack8(*(const uint4*)(wp + (size_t)o * kk + c), wf);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o] += xf[e] * wf[e];
          }
        }
      }
    }
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
      const size_t off = (((size_t)b * COUT + o) * H + yh) * W + xw;
      if (out_f32) ((float*)y)[off] = acc[o];
      else ((uint16_t*)y)[off] = f2bf(acc[o]);
    }
  }
}

// out[b][i] = a[i] + m[b][i] in fp32 (scale_shift_table + temb.float())
__global__ void bcast_add_f32_kernel(const float* __restrict__ a, const uint16_t* __restrict__ m, float* __restrict__ out,
                                     int B, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n) return;
  out[i] = a[i % n] + bf2f(m[i]);
}

// Patch gather / scatter for a Conv3d whose kernel equals its stride.  One thread per 8 output elements would need the
// patch to be 8 wide; patches here are (1,2,2), so this is a plain element kernel: 2.7 MB per call at Wan 480p.
__global__ void patchify3d_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ tok, int B, int C, int F, int H,
                                  int W, int pt, int ph, int pw) {
  const int f = F / pt, h = H / ph, w = W / pw, K = C * pt * ph * pw;
  const size_t total = (size_t)B * f * h * w * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int k = (int)(i % K);
    size_t s = i / K;
    const int dw = k % pw; k /= pw;
    const int dh = k % ph; k /= ph;
    const int dt = k % pt; k /= pt;
    const int c = k;
    const int xw = (int)(s % w); s /= w;
    const int yh = (int)(s % h); s /= h;
    const int tf = (int)(s % f); s /= f;
    const int b = (int)s;
    tok[i] = x[((((size_t)b * C + c) * F + tf * pt + dt) * H + yh * ph + dh) * W + xw * pw + dw];
  }
}
__global__ void unpatchify3d_kernel(const uint16_t* __restrict__ tok, uint16_t* __restrict__ x, int B, int C, int F,
                                    int H, int W, int pt, int ph, int pw) {
  const int f = F / pt, h = H / ph, w = W / pw, K = C * pt * ph * pw;
  const size_t total = (size_t)B * C * F * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int X = (int)(r % W); r /= W;
    const int Y = (int)(r % H); r /= H;
    const int T = (int)(r % F); r /= F;
    const int c = (int)(r % C); r /= C;
    const int b = (int)r;
    const int tf = T / pt, dt = T % pt, yh = Y / ph, dh = Y % ph, xw = X / pw, dw = X % pw;
    const size_t s = (((size_t)b * f + tf) * h + yh) * w + xw;
    x[i] = tok[s * K + ((size_t)(dt * ph + dh) * pw + dw) * C + c];
  }
}

// out[c][r] = in[r][c] (bf16), 64x64 tiles through LDS (padded rows: conflict-free column reads).  Used to hand a
// (B, S, H, D) value tensor to the flash kernel, which consumes V^T (keys contiguous).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                             int R, int Cc, long long ldi, long long ldo) {
  __shared__ uint16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cc) ? in[(size_t)r * ldi + c] : (uint16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cc && r < R) out[(size_t)c * ldo + r] = tile[tx][i];
  }
}

// out[b][c][p] = in[b][p][c] for c < cout: the first `cout` (<= 8) channels of a channels-last tensor whose rows hold `cpad`
// channels, as NCHW planes.  Tail of a thin-output conv run on the MFMA implicit-GEMM kernel with its output channels
// zero-padded to 16 (conv_out of the U-Nets and of the VAE decoder, models/unets/unet_2d_condition.py:1230,
// models/autoencoders/vae.py:309): one 16-byte read per pixel, `cout` coalesced 2-byte plane writes.
// MODE >= 0 fuses VaeImageProcessor.postprocess (image_processor.py:738-786; the modes and the arithmetic of
// image_postprocess_kernel below, applied to the same bf16-rounded conv output) into this pass, so the decoded image is
// written once, in the layout and type the caller asked for: 0 = NCHW fp32 in [0, 1], 1 = NHWC fp32, 2 = NHWC uint8.
template <int MODE>
__global__ __launch_bounds__(256) void nhwc_take_nchw_kernel(const uint16_t* __restrict__ in, void* __restrict__ outv,
                                                             long long B, long long HW, int cpad, int cout) {
  const long long total = B * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long b = i / HW, px = i - b * HW;
    const uint4 v = *(const uint4*)(in + (size_t)i * cpad);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c >= cout) break;
      const uint16_t h = (uint16_t)((c & 1) ? (w[c >> 1] >> 16) : (w[c >> 1] & 0xffffu));
      if (MODE < 0) {
        ((uint16_t*)outv)[((size_t)b * cout + c) * HW + px] = h;
      } else {
        const float y = fminf(fmaxf(bf2f(h) * 0.5f + 0.5f, 0.f), 1.f);
        if (MODE == 0) ((float*)outv)[((size_t)b * cout + c) * HW + px] = y;
        else if (MODE == 1) ((float*)outv)[(size_t)i * cout + c] = y;
        else ((uint8_t*)outv)[(size_t)i * cout + c] = (uint8_t)rintf(y * 255.f);
      }
    }
  }
}

// dst[d0][d2][d1][:] = src[d0][d1][d2][:] in 16-byte chunks (D3 a multiple of 8 bf16).  WanResample 'upsample3d'
// (autoencoder_kl_wan.py:297-299): time_conv produces 2C channels per position, [frame][pos][2][C] here, and the two
// halves become consecutive frames, [frame][2][pos][C].  Pure copy: HBM-bound, chunk index = destination order.
__global__ __launch_bounds__(256) void permute_0213_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                           long long D0, int D1, int D2, int ch) {
  const long long total = D0 * D1 * D2 * ch;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % ch);
    long long t = i / ch;
    const int d1 = (int)(t % D1);
    t /= D1;
    const int d2 = (int)(t % D2);
    const long long d0 = t / D2;
    dst[i] = src[((d0 * D1 + d1) * D2 + d2) * ch + c];
  }
}

// Channels-last frames [B*T][HW][Cs] (first C <= 4 channels used) -> video [B][C][T][HW], clamped to [lo, hi]
// (AutoencoderKLWan._decode: torch.clamp(out, -1, 1), autoencoder_kl_wan.py:1210).  One thread per position.
template <bool F32>
__global__ __launch_bounds__(256) void frames_to_ncthw_kernel(const uint16_t* __restrict__ src, void* __restrict__ dst,
                                                              int B, int T, long long HW, int Cs, int C, float lo,
                                                              float hi) {
  const long long total = (long long)B * T * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long p = i % HW;
    const long long bt = i / HW;
    const int t = (int)(bt % T);
    const long long b = bt / T;
    const uint2 v = *(const uint2*)(src + (size_t)i * Cs);
    const float f[4] = {bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y)};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c >= C) break;
      const float o = fminf(fmaxf(f[c], lo), hi);
      const size_t at = (((size_t)b * C + c) * T + t) * (size_t)HW + (size_t)p;
      if (F32) ((float*)dst)[at] = o;
      else ((uint16_t*)dst)[at] = f2bf(o);
    }
  }
}

// VaeImageProcessor.postprocess (image_processor.py:738-786): denormalize (x * 0.5 + 0.5).clamp(0, 1) (:222-234), then
// "pt" keeps NCHW, "np" moves channels last (pt_to_numpy, :191-204), "pil" additionally (x * 255).round() -> uint8
// (numpy_to_pil, :128-149; numpy rounds half to even, so does v_rndne).  One thread per position, C <= 4 planes.
template <bool IN_F32, int MODE>  // MODE 0: NCHW fp32, 1: NHWC fp32, 2: NHWC uint8
__global__ __launch_bounds__(256) void image_postprocess_kernel(const void* __restrict__ img, void* __restrict__ out,
                                                                int B, int C, long long HW) {
  const long long total = (long long)B * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long b = i / HW, p = i - b * HW;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c >= C) break;
      const size_t src = ((size_t)b * C + c) * (size_t)HW + (size_t)p;
      const float x = IN_F32 ? ((const float*)img)[src] : bf2f(((const uint16_t*)img)[src]);
      const float v = fminf(fmaxf(x * 0.5f + 0.5f, 0.f), 1.f);
      if (MODE == 0) ((float*)out)[src] = v;
      else if (MODE == 1) ((float*)out)[(size_t)i * C + c] = v;
      else ((uint8_t*)out)[(size_t)i * C + c] = (uint8_t)rintf(v * 255.f);
    }
  }
}

}  // namespace

extern "C" int da_transpose_bf16(const void* in, void* out, int R, int Cc, long long ldi, long long ldo, void* stream) {
  if (!in || !out || R <= 0 || Cc <= 0 || ldi < Cc || ldo < R) return DA_ERR_INVALID;
  DA_LAUNCH(transpose_bf16_kernel, dim3((Cc + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream,
            (const uint16_t*)in, (uint16_t*)out, R, Cc, ldi, ldo);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

static int take_launch(const void* in, void* out, long long B, long long HW, int cpad, int cout, int mode, void* stream) {
  const long long total = B * HW;
  const unsigned blocks = (unsigned)((total + 255) / 256 < 65535 * 16 ? (total + 255) / 256 : 65535 * 16);
  hipStream_t s = (hipStream_t)stream;
#define DA_TAKE(M_) DA_LAUNCH(nhwc_take_nchw_kernel<M_>, dim3(blocks), dim3(256), 0, s, (const uint16_t*)in, out, B, HW, cpad, cout)
  if (mode < 0) DA_TAKE(-1); else if (mode == 0) DA_TAKE(0); else if (mode == 1) DA_TAKE(1); else DA_TAKE(2);
#undef DA_TAKE
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_nhwc_take_nchw_bf16(const void* in, void* out, long long B, long long HW, int cpad, int cout, void* stream) {
  if (!in || !out || B <= 0 || HW <= 0 || cpad < 8 || (cpad & 7) || cout <= 0 || cout > 8) return DA_ERR_INVALID;
  return take_launch(in, out, B, HW, cpad, cout, -1, stream);
}

extern "C" int da_nhwc_take_postprocess(const void* in, void* out, long long B, long long HW, int cpad, int cout, int mode,
                                        void* stream) {
  if (!in || !out || B <= 0 || HW <= 0 || cpad < 8 || (cpad & 7) || cout <= 0 || cout > 4 || mode < 0 || mode > 2)
    return DA_ERR_INVALID;
  return take_launch(in, out, B, HW, cpad, cout, mode, stream);
}

extern "C" int da_bcast_add_f32(const float* a, const void* m, float* out, int B, int n, void* stream) {
  if (!a || !m || !out || B <= 0 || n <= 0) return DA_ERR_INVALID;
  DA_LAUNCH(bcast_add_f32_kernel, dim3((B * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, (const uint16_t*)m,
            out, B, n);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

static int patch_args_ok(const void* a, const void* b, int B, int C, int F, int H, int W, int pt, int ph, int pw) {
  if (!a || !b || B <= 0 || C <= 0 || F <= 0 || H <= 0 || W <= 0 || pt <= 0 || ph <= 0 || pw <= 0) return 0;
  return (F % pt) == 0 && (H % ph) == 0 && (W % pw) == 0;
}

extern "C" int da_patchify3d_bf16(const void* x, void* tokens, int B, int C, int F, int H, int W, int pt, int ph, int pw,
                                  void* stream) {
  if (!patch_args_ok(x, tokens, B, C, F, H, W, pt, ph, pw)) return DA_ERR_INVALID;
  size_t total = (size_t)B * C * F * H * W, blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  DA_LAUNCH(patchify3d_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
            (uint16_t*)tokens, B, C, F, H, W, pt, ph, pw);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_unpatchify3d_bf16(const void* tokens, void* x, int B, int C, int F, int H, int W, int pt, int ph,
                                    int pw, void* stream) {
  if (!patch_args_ok(tokens, x, B, C, F, H, W, pt, ph, pw)) return DA_ERR_INVALID;
  size_t total = (size_t)B * C * F * H * W, blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  DA_LAUNCH(unpatchify3d_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)tokens,
            (uint16_t*)x, B, C, F, H, W, pt, ph, pw);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_timestep_embedding(const float* t, const float* table, const int* step_idx, void* out, int B,
                                     int dim, int flip_sin_to_cos, float shift, float scale, float max_period,
                                     int out_f32, void* stream) {
  if ((!t && !table) || !out || B <= 0 || dim <= 0 || (dim & 1)) return DA_ERR_INVALID;
  if (table && !step_idx) return DA_ERR_INVALID;
  const int total = B * (dim / 2);
  DA_LAUNCH(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, table,
                     step_idx, out, B, dim, flip_sin_to_cos, shift, scale, max_period, out_f32);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_linear_small_m_bf16(const void* x, const void* W, const void* bias, const void* res, void* out, int M,
                                      int N, int K, int ldx, int ldo, int ldr, int act_in, int act_out, void* stream) {
  if (!x || !W || !out || M <= 0 || M > 8 || N <= 0 || K <= 0) return DA_ERR_INVALID;
  if ((K & 7) || (ldx & 7)) return DA_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((N + 3) / 4), block(256);
  if (M <= 2)
    DA_LAUNCH(linear_small_m_kernel<2>, grid, block, 0, s, (const uint16_t*)x, (const uint16_t*)W,
                       (const uint16_t*)bias, (const uint16_t*)res, (uint16_t*)out, M, N, K, ldx, ldo, ldr, act_in,
                       act_out);
  else
    DA_LAUNCH(linear_small_m_kernel<8>, grid, block, 0, s, (const uint16_t*)x, (const uint16_t*)W,
                       (const uint16_t*)bias, (const uint16_t*)res, (uint16_t*)out, M, N, K, ldx, ldo, ldr, act_in,
                       act_out);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_conv_thin_in_bf16(const void* x, const void* w, const void* bias, void* y, int B, int H, int W,
                                    int Cin, int Cout, int ksize, int in_nchw, float in_div, float in_add, void* stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0) return DA_ERR_INVALID;
  if (Cin <= 0 || Cin > 16 || Cout <= 0 || (Cout & 7) || (ksize != 1 && ksize != 3)) return DA_ERR_UNSUPPORTED;
  if (in_div == 0.f) in_div = 1.f;
  // output channels per block column: as many as fit 64 KiB of LDS (no opt-in attribute, two blocks per CU)
  const int kk = ksize * ksize * Cin;
  int coc = ((64 * 1024) / (kk * (int)sizeof(float))) & ~7;
  if (coc > Cout) coc = Cout;
  if (coc < 8) return DA_ERR_UNSUPPORTED;
  const int nchunk = (Cout + coc - 1) / coc;
  const size_t lds = (size_t)coc * kk * sizeof(float);
  size_t total = (size_t)B * H * W * (coc / 8);
  size_t blocks = (total + 255) / 256;
  // every block first stages its chunk's weights into LDS (up to 64 KiB, transposed, one integer division per element): with
  // thousands of blocks that staging, not the convolution, was the kernel (U-Net conv_in 4 -> 320 at 128 x 128: 140 us for 21 MB
  // of output).  Three blocks per CU (what 46-64 KiB of LDS each admits), grid-stride over the pixels.
  const size_t cap = 768 / nchunk > 0 ? 768 / nchunk : 1;
  if (blocks > cap) blocks = cap;
  const char* quad_env = getenv("DA_CONV_IN_QUAD");             // (read per call: tests compare the two kernels in one process)
  const int quad = quad_env ? atoi(quad_env) : 1;
  if (quad && ksize == 3 && (Cin == 3 || Cin == 4)) {          // four pixels per thread (round 6; bit-identical)
    size_t qblocks = ((size_t)B * H * ((W + 3) / 4) * (coc / 8) + 255) / 256;
    if (qblocks > cap) qblocks = cap;
    if (Cin == 4)
      DA_LAUNCH(conv_thin_in4_kernel<4>, dim3((unsigned)qblocks, (unsigned)nchunk), dim3(256), lds, (hipStream_t)stream, (const uint16_t*)x,
                (const uint16_t*)w, (const uint16_t*)bias, (uint16_t*)y, B, H, W, Cout, in_nchw, in_div, in_add, coc);
    else
      DA_LAUNCH(conv_thin_in4_kernel<3>, dim3((unsigned)qblocks, (unsigned)nchunk), dim3(256), lds, (hipStream_t)stream, (const uint16_t*)x,
                (const uint16_t*)w, (const uint16_t*)bias, (uint16_t*)y, B, H, W, Cout, in_nchw, in_div, in_add, coc);
    DA_CHECK_LAUNCH();
    return DA_OK;
  }
  DA_LAUNCH(conv_thin_in_kernel, dim3((unsigned)blocks, (unsigned)nchunk), dim3(256), lds, (hipStream_t)stream,
            (const uint16_t*)x, (const uint16_t*)w, (const uint16_t*)bias, (uint16_t*)y, B, H, W, Cin, Cout, ksize,
            in_nchw, in_div, in_add, coc);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_conv_thin_out_bf16(const void* x, const void* w, const void* bias, void* y, int B, int H, int W,
                                     int Cin, int Cout, int out_f32, void* stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0) return DA_ERR_INVALID;
  if (Cin <= 0 || (Cin & 7) || Cout <= 0) return DA_ERR_UNSUPPORTED;
  const size_t lds = (size_t)Cout * 9 * Cin * sizeof(uint16_t);
  if (lds > 150 * 1024) return DA_ERR_UNSUPPORTED;
  size_t total = (size_t)B * H * W;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t s = (hipStream_t)stream;
#define DA_CO(N)                                                                                                      \
  do {                                                                                                                \
    auto kern = conv_thin_out_kernel<N>;                                                                              \
    static size_t lds_enabled = 48 * 1024;                                                                            \
    if (lds > lds_enabled) {                                                                                          \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) !=           \
          hipSuccess)                                                                                                 \
        return DA_ERR_LAUNCH;                                                                                         \
      lds_enabled = 150 * 1024;                                                                                       \
    }                                                                                                                 \
    DA_LAUNCH(kern, dim3((unsigned)blocks), dim3(256), lds, s, (const uint16_t*)x, (const uint16_t*)w,       \
                       (const uint16_t*)bias, y, B, H, W, Cin, out_f32);                                              \
  } while (0)
  switch (Cout) {
    case 3: DA_CO(3); break;
    case 4: DA_CO(4); break;
    case 8: DA_CO(8); break;
    case 16: DA_CO(16); break;
    default: return DA_ERR_UNSUPPORTED;
  }
#undef DA_CO
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_permute_0213_bf16(const void* src, void* dst, long long D0, int D1, int D2, int D3, void* stream) {
  if (!src || !dst || D0 <= 0 || D1 <= 0 || D2 <= 0 || D3 <= 0 || (D3 & 7)) return DA_ERR_INVALID;
  const long long total = D0 * D1 * D2 * (D3 >> 3);
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  DA_LAUNCH(permute_0213_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src,
            (uint4*)dst, D0, D1, D2, D3 >> 3);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_frames_to_ncthw_bf16(const void* src, void* dst, int B, int T, long long HW, int Cs, int C, float lo,
                                       float hi, int out_f32, void* stream) {
  if (!src || !dst || B <= 0 || T <= 0 || HW <= 0 || C <= 0 || C > 4 || Cs < 4 || (Cs & 3) || !(lo <= hi))
    return DA_ERR_INVALID;
  const long long total = (long long)B * T * HW;
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (out_f32) {
    DA_LAUNCH(frames_to_ncthw_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
              (const uint16_t*)src, dst, B, T, HW, Cs, C, lo, hi);
  } else {
    DA_LAUNCH(frames_to_ncthw_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
              (const uint16_t*)src, dst, B, T, HW, Cs, C, lo, hi);
  }
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_image_postprocess(const void* img, void* out, int B, int C, long long HW, int in_f32, int mode,
                                    void* stream) {
  if (!img || !out || B <= 0 || C <= 0 || C > 4 || HW <= 0 || mode < 0 || mode > 2) return DA_ERR_INVALID;
  const long long total = (long long)B * HW;
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipStream_t s = (hipStream_t)stream;
#define DA_PP(F_, M_)                                                                                              \
  DA_LAUNCH((image_postprocess_kernel<F_, M_>), dim3((unsigned)blocks), dim3(256), 0, s, img, out, B, C, HW)
  if (in_f32) {
    if (mode == 0) DA_PP(true, 0); else if (mode == 1) DA_PP(true, 1); else DA_PP(true, 2);
  } else {
    if (mode == 0) DA_PP(false, 0); else if (mode == 1) DA_PP(false, 1); else DA_PP(false, 2);
  }
#undef DA_PP
  DA_CHECK_LAUNCH();
  return DA_OK;
}
