// Fused scheduler.step + classifier-free-guidance kernels for gfx950 (MI355X).
//
// Reference (diffusers src/diffusers/):
//   CFG combine        pipelines/stable_diffusion/pipeline_stable_diffusion.py:1054-1055,
//                      pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl.py:1223-1225
//   Euler              schedulers/scheduling_euler_discrete.py:326-348 (scale_model_input), :685-800 (step)
//   DDIM               schedulers/scheduling_ddim.py:384-514
//   DDPM               schedulers/scheduling_ddpm.py:461-567
//   FlowMatch Euler    schedulers/scheduling_flow_match_euler_discrete.py:423-523
//
// The reference issues ~10 elementwise launches per step, each rounding to the tensor dtype.  These kernels do the
// whole update in one pass but reproduce the reference's rounding points exactly (bf16 tensors round after every
// torch op; 0-d fp32 scalars promote nothing), so results are bit-identical to the reference for identical inputs.
// Per-step scalars live in a device table (8 floats per step) indexed by a device-resident step counter, so a whole
// denoising step can be captured once in a HIP graph and replayed for every step.
#include "common.cuh"

namespace {

template <typename T>
struct IO;
template <>
struct IO<uint16_t> {
  static __device__ __forceinline__ float ld(const uint16_t* p, size_t i) { return bf2f(p[i]); }
  static __device__ __forceinline__ void st(uint16_t* p, size_t i, float v) { p[i] = f2bf(v); }
  static __device__ __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }  // torch rounds each op's result
};
template <>
struct IO<float> {
  static __device__ __forceinline__ float ld(const float* p, size_t i) { return p[i]; }
  static __device__ __forceinline__ void st(float* p, size_t i, float v) { p[i] = v; }
  static __device__ __forceinline__ float rnd(float v) { return v; }
};

// noise_pred = uncond + g * (text - uncond), every op in the tensor dtype
template <typename T>
__device__ __forceinline__ float cfg_combine(float u, float c, float g) {
  const float d = IO<T>::rnd(__fsub_rn(c, u));
  const float gd = IO<T>::rnd(__fmul_rn(g, d));
  return IO<T>::rnd(__fadd_rn(u, gd));
}

// eps layout with CFG: [2][n] = (uncond, cond); without: [n]
template <typename T, bool CFG>
__device__ __forceinline__ float load_eps(const T* eps, size_t i, size_t n, float g) {
  if (CFG) return cfg_combine<T>(IO<T>::ld(eps, i), IO<T>::ld(eps, n + i), g);
  return IO<T>::ld(eps, i);
}

// rescale_noise_cfg (pipelines/stable_diffusion/pipeline_stable_diffusion.py:69-92), used when guidance_rescale > 0:
//   std_text = noise_pred_text.std(dims 1..)   std_cfg = noise_cfg.std(dims 1..)         (unbiased, per sample)
//   noise_cfg = guidance_rescale * (noise_cfg * (std_text / std_cfg)) + (1 - guidance_rescale) * noise_cfg
// Pass 1: one workgroup per sample sums x and x^2 of the text prediction and of the CFG combination (fp64 accumulators:
// torch reduces a bf16 tensor in fp32 and rounds the RESULT to the tensor dtype, so what has to match is the rounded std)
// and stores ratio = rnd(rnd(std_text) / rnd(std_cfg)).  Pass 2: the combination again (same roundings as cfg_combine), then
// every product / sum of the formula rounded in the tensor dtype.  eps layout [2][B][n_per] = (uncond, cond).
template <typename T>
__global__ __launch_bounds__(1024) void cfg_rescale_stats_kernel(const T* __restrict__ eps, float* __restrict__ ratio, float g,
                                                                 size_t n_per, int B) {
  const int b = blockIdx.x;
  const T* u = eps + (size_t)b * n_per;
  const T* c = eps + ((size_t)B + b) * n_per;
  double st = 0, st2 = 0, sc = 0, sc2 = 0;
  for (size_t i = threadIdx.x; i < n_per; i += blockDim.x) {
    const float tv = IO<T>::ld(c, i);
    const float cv = cfg_combine<T>(IO<T>::ld(u, i), tv, g);
    st += tv; st2 += (double)tv * tv;
    sc += cv; sc2 += (double)cv * cv;
  }
  __shared__ double red[4][16];
  double v[4] = {st, st2, sc, sc2};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[0][wave] = v[0]; red[1][wave] = v[1]; red[2][wave] = v[2]; red[3][wave] = v[3]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[4] = {0, 0, 0, 0};
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { t[0] += red[0][w]; t[1] += red[1][w]; t[2] += red[2][w]; t[3] += red[3][w]; }
    const double n = (double)n_per;
    const double var_t = fmax((t[1] - t[0] * t[0] / n) / (n - 1.0), 0.0), var_c = fmax((t[3] - t[2] * t[2] / n) / (n - 1.0), 0.0);
    const float std_t = IO<T>::rnd((float)sqrt(var_t)), std_c = IO<T>::rnd((float)sqrt(var_c));
    ratio[b] = IO<T>::rnd(__fdiv_rn(std_t, std_c));
  }
}

template <typename T>
__global__ void cfg_rescale_apply_kernel(const T* __restrict__ eps, T* __restrict__ out, const float* __restrict__ ratio,
                                         float g, float gr, float one_minus_gr, size_t n_per, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float cv = cfg_combine<T>(IO<T>::ld(eps, i), IO<T>::ld(eps, n + i), g);
    const float resc = IO<T>::rnd(__fmul_rn(cv, ratio[i / n_per]));
    const float a = IO<T>::rnd(__fmul_rn(gr, resc));
    const float b = IO<T>::rnd(__fmul_rn(one_minus_gr, cv));
    IO<T>::st(out, i, __fadd_rn(a, b));
  }
}

// Euler (gamma = 0): table row = [sigma, sigma_next, dt, sqrt(sigma^2+1), c_out, sigma^2+1, -, timestep]
// PRED: 0 epsilon, 1 v_prediction, 2 sample (scheduling_euler_discrete.py:760-775)
template <typename T, bool CFG, int PRED>
__global__ void euler_step_kernel(const T* __restrict__ eps, const T* __restrict__ x, T* __restrict__ out,
                                  const float* __restrict__ table, const int* __restrict__ step_idx, float g,
                                  size_t n) {
  const float* row = table + (size_t)(*step_idx) * 8;
  const float sigma = row[0], dt = row[2], c_out = row[4], s2p1 = row[5];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float e = load_eps<T, CFG>(eps, i, n, g);
    const float s = IO<T>::ld(x, i);                       // sample.to(float32)
    float x0;                                              // pred_original_sample (fp32)
    if (PRED == 0) {
      const float se = IO<T>::rnd(__fmul_rn(sigma, e));    // sigma_hat * model_output  (model dtype)
      x0 = __fsub_rn(s, se);
    } else if (PRED == 1) {
      // model_output * (-sigma / sqrt(sigma^2+1)) (model dtype) + sample / (sigma^2+1) (fp32)
      x0 = __fadd_rn(IO<T>::rnd(__fmul_rn(e, c_out)), __fdiv_rn(s, s2p1));
    } else {
      x0 = e;
    }
    const float der = __fdiv_rn(__fsub_rn(s, x0), sigma);  // derivative
    const float prev = __fadd_rn(s, __fmul_rn(der, dt));
    IO<T>::st(out, i, prev);
  }
}

// scale_model_input for Euler, replicated `rep` times along batch (torch.cat([latents] * 2))
template <typename T>
__global__ void euler_scale_input_kernel(const T* __restrict__ x, T* __restrict__ out, const float* __restrict__ table,
                                         const int* __restrict__ step_idx, int rep, size_t n) {
  const float den = table[(size_t)(*step_idx) * 8 + 3];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = __fdiv_rn(IO<T>::ld(x, i), den);
    for (int r = 0; r < rep; ++r) IO<T>::st(out, (size_t)r * n + i, v);
  }
}

// DDIM / DDPM: row = [sqrt(beta_t), sqrt(alpha_t), k0, ke, kx, kn, clip_range(0=off), timestep]
//   PRED 0 (epsilon)       x0 = (x - sqrt(beta_t) * out) / sqrt(alpha_t)                 pred_eps = out
//   PRED 1 (v_prediction)  x0 = sqrt(alpha_t) * x - sqrt(beta_t) * out                   pred_eps = sqrt(alpha_t) * out + sqrt(beta_t) * x
//   PRED 2 (sample)        x0 = out                                                      pred_eps = (x - sqrt(alpha_t) * x0) / sqrt(beta_t)
//   (scheduling_ddim.py:455-468, scheduling_ddpm.py:505-517; pred_eps is formed from the UNCLIPPED x0, as the reference
//   does unless use_clipped_model_output)
//   prev = k0*x0 [clamped] (+ ke*pred_eps) (+ kx*x) (+ kn*noise)     each product / sum rounded in the tensor dtype
template <typename T, bool CFG, int PRED>
__global__ void x0_linear_step_kernel(const T* __restrict__ eps, const T* __restrict__ x, const T* __restrict__ noise,
                                      T* __restrict__ out, const float* __restrict__ table,
                                      const int* __restrict__ step_idx, float g, size_t n, size_t noise_step_stride) {
  const float* row = table + (size_t)(*step_idx) * 8;
  if (noise) noise += (size_t)(*step_idx) * noise_step_stride;  // pre-drawn per-step noise: graph-replay safe
  const float cb = row[0], ca = row[1], k0 = row[2], ke = row[3], kx = row[4], kn = row[5], clip = row[6];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float e = load_eps<T, CFG>(eps, i, n, g);
    const float s = IO<T>::ld(x, i);
    float x0, pe;
    if (PRED == 0) {
      x0 = IO<T>::rnd(__fmul_rn(cb, e));
      x0 = IO<T>::rnd(__fsub_rn(s, x0));
      x0 = IO<T>::rnd(__fdiv_rn(x0, ca));
      pe = e;
    } else if (PRED == 1) {
      x0 = IO<T>::rnd(__fsub_rn(IO<T>::rnd(__fmul_rn(ca, s)), IO<T>::rnd(__fmul_rn(cb, e))));
      pe = IO<T>::rnd(__fadd_rn(IO<T>::rnd(__fmul_rn(ca, e)), IO<T>::rnd(__fmul_rn(cb, s))));
    } else {
      x0 = e;
      pe = IO<T>::rnd(__fdiv_rn(IO<T>::rnd(__fsub_rn(s, IO<T>::rnd(__fmul_rn(ca, x0)))), cb));
    }
    if (clip > 0.f) x0 = fminf(fmaxf(x0, -clip), clip);
    float acc = IO<T>::rnd(__fmul_rn(k0, x0));
    if (ke != 0.f) acc = IO<T>::rnd(__fadd_rn(acc, IO<T>::rnd(__fmul_rn(ke, pe))));
    if (kx != 0.f) acc = IO<T>::rnd(__fadd_rn(acc, IO<T>::rnd(__fmul_rn(kx, s))));
    if (kn != 0.f && noise) acc = IO<T>::rnd(__fadd_rn(acc, IO<T>::rnd(__fmul_rn(kn, IO<T>::ld(noise, i)))));
    IO<T>::st(out, i, acc);
  }
}

// FlowMatch Euler: row = [sigma, sigma_next, dt, -, -, -, -, timestep]; prev = float(x) + (dt * v in model dtype), stored in
// the MODEL OUTPUT's dtype (scheduling_flow_match_euler_discrete.py:484,:517: sample.to(float32) ... .to(model_output.dtype)).
// TX = dtype of the sample, TV = dtype of the model output and of `out` (TX = float, TV = bf16 is what the reference's
// Wan loop hands over on its first step).
template <typename TX, typename TV, bool CFG>
__global__ void flowmatch_step_kernel(const TV* __restrict__ v, const TX* __restrict__ x, TV* __restrict__ out,
                                      const float* __restrict__ table, const int* __restrict__ step_idx, float g,
                                      size_t n) {
  const float dt = table[(size_t)(*step_idx) * 8 + 2];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float e = load_eps<TV, CFG>(v, i, n, g);
    const float s = IO<TX>::ld(x, i);
    const float de = IO<TV>::rnd(__fmul_rn(dt, e));
    IO<TV>::st(out, i, __fadd_rn(s, de));
  }
}

// UniPC (B(h), predict-x0, flow prediction) step, orders 1 / 2 -- schedulers/scheduling_unipc_multistep.py:760-1300.
// One pass does convert_model_output, the corrector (multistep_uni_c_bh_update) and the predictor
// (multistep_uni_p_bh_update) and rolls the history IN PLACE: x <- next sample, last <- corrected sample,
// m2 <- m1, m1 <- this step's x0 prediction.  Every tensor op of the reference is reproduced with its rounding point
// (TX = dtype of the sample / history, TV = dtype of the model output; 0-d fp32 scalars never promote a tensor).
// coef row (16 floats): [sigma, use_corr, order_c, c1c, c2c, c3c, rk_c, rho0_c, rho_last_c, order_p, c1p, c2p, c3p, rk_p, -, -]
//   c1 = sigma_t / sigma_s0, c2 = alpha_t * h_phi_1, c3 = alpha_t * B_h   (corrector: t = i, s0 = i-1; predictor: t = i+1, s0 = i)
template <typename TX, typename TV, bool CFG>
__global__ void unipc_flow_step_kernel(const TV* __restrict__ v, TX* __restrict__ x, TX* __restrict__ last,
                                       TX* __restrict__ m1, TX* __restrict__ m2, const float* __restrict__ coef,
                                       const int* __restrict__ step_idx, float g, size_t n) {
  const float* r = coef + (size_t)(*step_idx) * 16;
  const float sigma = r[0];
  const bool use_corr = r[1] != 0.f;
  const int order_c = (int)r[2], order_p = (int)r[9];
  const float c1c = r[3], c2c = r[4], c3c = r[5], rk_c = r[6];
  const float rho0 = IO<TX>::rnd(r[7]), rhol = IO<TX>::rnd(r[8]);   // rhos_c is a tensor of the sample dtype
  const float c1p = r[10], c2p = r[11], c3p = r[12], rk_p = r[13];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float e = load_eps<TV, CFG>(v, i, n, g);
    const float xs = IO<TX>::ld(x, i);
    const float m1o = IO<TX>::ld(m1, i);
    const float se = IO<TV>::rnd(__fmul_rn(sigma, e));
    const float mn = IO<TX>::rnd(__fsub_rn(xs, se));                        // x0_pred = sample - sigma * model_output
    float xc = xs;
    if (use_corr) {
      const float t1 = IO<TX>::rnd(__fmul_rn(c1c, IO<TX>::ld(last, i)));
      const float t2 = IO<TX>::rnd(__fmul_rn(c2c, m1o));
      const float xt = IO<TX>::rnd(__fsub_rn(t1, t2));
      const float d1t = IO<TX>::rnd(__fsub_rn(mn, m1o));
      float inner = IO<TX>::rnd(__fmul_rn(rhol, d1t));
      if (order_c == 2) {
        float d = IO<TX>::rnd(__fsub_rn(IO<TX>::ld(m2, i), m1o));
        d = IO<TX>::rnd(__fdiv_rn(d, rk_c));
        const float corr = IO<TX>::rnd(__fmul_rn(rho0, d));
        inner = IO<TX>::rnd(__fadd_rn(corr, inner));
      }
      xc = IO<TX>::rnd(__fsub_rn(xt, IO<TX>::rnd(__fmul_rn(c3c, inner))));
    }
    const float p1 = IO<TX>::rnd(__fmul_rn(c1p, xc));
    const float p2 = IO<TX>::rnd(__fmul_rn(c2p, mn));
    float xn = IO<TX>::rnd(__fsub_rn(p1, p2));
    if (order_p == 2) {
      float d = IO<TX>::rnd(__fsub_rn(m1o, mn));
      d = IO<TX>::rnd(__fdiv_rn(d, rk_p));
      const float pred = IO<TX>::rnd(__fmul_rn(0.5f, d));
      xn = IO<TX>::rnd(__fsub_rn(xn, IO<TX>::rnd(__fmul_rn(c3p, pred))));
    }
    IO<TX>::st(m2, i, m1o);
    IO<TX>::st(m1, i, mn);
    IO<TX>::st(last, i, xc);
    IO<TX>::st(x, i, xn);
  }
}

__global__ void advance_step_kernel(int* step_idx) { *step_idx += 1; }

// out = x * s in the tensor dtype (latents * init_noise_sigma, pipeline_stable_diffusion.py:713)
template <typename T>
__global__ void mul_scalar_kernel(const T* __restrict__ x, T* __restrict__ out, float sc, int rep, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = __fmul_rn(IO<T>::ld(x, i), sc);
    for (int r = 0; r < rep; ++r) IO<T>::st(out, (size_t)r * n + i, v);  // rep > 1: torch.cat([x] * rep) fused
  }
}

// latents.to(transformer_dtype) (pipeline_wan.py:600) fused with the CFG batch doubling: fp32 -> bf16, `rep` copies
__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ out, int rep, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint16_t v = f2bf(x[i]);
    for (int r = 0; r < rep; ++r) out[(size_t)r * n + i] = v;
  }
}

inline dim3 ew_grid(size_t n) {
  size_t b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

}  // namespace

extern "C" int da_euler_step(const void* eps, const void* x, void* out, const float* table, const int* step_idx,
                             int cfg, float guidance, long long n_, int dtype, int pred_type, void* stream) {
  if (!eps || !x || !out || !table || !step_idx || n_ <= 0) return DA_ERR_INVALID;
  if (pred_type < 0 || pred_type > 2) return DA_ERR_INVALID;
  const size_t n = (size_t)n_;
  hipStream_t s = (hipStream_t)stream;
#define DA_EU(T, C, P) \
  DA_LAUNCH((euler_step_kernel<T, C, P>), ew_grid(n), dim3(256), 0, s, (const T*)eps, (const T*)x, (T*)out, table, step_idx, guidance, n)
#define DA_EU_P(T, C) \
  do { if (pred_type == 0) DA_EU(T, C, 0); else if (pred_type == 1) DA_EU(T, C, 1); else DA_EU(T, C, 2); } while (0)
  if (dtype == DA_DTYPE_BF16) {
    if (cfg) DA_EU_P(uint16_t, true); else DA_EU_P(uint16_t, false);
  } else if (dtype == DA_DTYPE_F32) {
    if (cfg) DA_EU_P(float, true); else DA_EU_P(float, false);
  } else {
    return DA_ERR_UNSUPPORTED;
  }
#undef DA_EU_P
#undef DA_EU
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_euler_scale_model_input(const void* x, void* out, const float* table, const int* step_idx, int rep,
                                          long long n_, int dtype, void* stream) {
  if (!x || !out || !table || !step_idx || n_ <= 0 || rep <= 0) return DA_ERR_INVALID;
  const size_t n = (size_t)n_;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DA_DTYPE_BF16)
    DA_LAUNCH((euler_scale_input_kernel<uint16_t>), ew_grid(n), dim3(256), 0, s, (const uint16_t*)x, (uint16_t*)out, table, step_idx, rep, n);
  else if (dtype == DA_DTYPE_F32)
    DA_LAUNCH((euler_scale_input_kernel<float>), ew_grid(n), dim3(256), 0, s, (const float*)x, (float*)out, table, step_idx, rep, n);
  else
    return DA_ERR_UNSUPPORTED;
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_x0_linear_step(const void* eps, const void* x, const void* noise, long long noise_step_stride,
                                 void* out, const float* table, const int* step_idx, int cfg, float guidance,
                                 long long n_, int dtype, int pred_type, void* stream) {
  if (!eps || !x || !out || !table || !step_idx || n_ <= 0 || noise_step_stride < 0) return DA_ERR_INVALID;
  if (pred_type < 0 || pred_type > 2) return DA_ERR_INVALID;
  const size_t n = (size_t)n_;
  hipStream_t s = (hipStream_t)stream;
#define DA_XL(T, C, P)                                                                                              \
  DA_LAUNCH((x0_linear_step_kernel<T, C, P>), ew_grid(n), dim3(256), 0, s, (const T*)eps, (const T*)x, (const T*)noise, \
            (T*)out, table, step_idx, guidance, n, (size_t)noise_step_stride)
#define DA_XL_P(T, C) \
  do { if (pred_type == 0) DA_XL(T, C, 0); else if (pred_type == 1) DA_XL(T, C, 1); else DA_XL(T, C, 2); } while (0)
  if (dtype == DA_DTYPE_BF16) {
    if (cfg) DA_XL_P(uint16_t, true); else DA_XL_P(uint16_t, false);
  } else if (dtype == DA_DTYPE_F32) {
    if (cfg) DA_XL_P(float, true); else DA_XL_P(float, false);
  } else {
    return DA_ERR_UNSUPPORTED;
  }
#undef DA_XL_P
#undef DA_XL
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_flowmatch_step(const void* v, const void* x, void* out, const float* table, const int* step_idx,
                                 int cfg, float guidance, long long n_, int dtype, int x_dtype, void* stream) {
  if (!v || !x || !out || !table || !step_idx || n_ <= 0) return DA_ERR_INVALID;
  const size_t n = (size_t)n_;
  hipStream_t s = (hipStream_t)stream;
#define DA_FM(TX, TV, C) \
  DA_LAUNCH((flowmatch_step_kernel<TX, TV, C>), ew_grid(n), dim3(256), 0, s, (const TV*)v, (const TX*)x, (TV*)out, table, step_idx, guidance, n)
  if (dtype == DA_DTYPE_BF16 && x_dtype == DA_DTYPE_BF16) { if (cfg) DA_FM(uint16_t, uint16_t, true); else DA_FM(uint16_t, uint16_t, false); }
  else if (dtype == DA_DTYPE_F32 && x_dtype == DA_DTYPE_F32) { if (cfg) DA_FM(float, float, true); else DA_FM(float, float, false); }
  else if (dtype == DA_DTYPE_BF16 && x_dtype == DA_DTYPE_F32) { if (cfg) DA_FM(float, uint16_t, true); else DA_FM(float, uint16_t, false); }
  else return DA_ERR_UNSUPPORTED;
#undef DA_FM
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_cfg_rescale(const void* eps, void* out, float* ratio_ws, int B, long long n_per_, float guidance,
                              float guidance_rescale, int dtype, void* stream) {
  if (!eps || !out || !ratio_ws || B <= 0 || n_per_ <= 1) return DA_ERR_INVALID;
  const size_t n_per = (size_t)n_per_, n = n_per * (size_t)B;
  hipStream_t s = (hipStream_t)stream;
  const float omg = (float)(1.0 - (double)guidance_rescale);   // the reference forms (1 - guidance_rescale) in Python doubles
  if (dtype == DA_DTYPE_BF16) {
    DA_LAUNCH((cfg_rescale_stats_kernel<uint16_t>), dim3(B), dim3(1024), 0, s, (const uint16_t*)eps, ratio_ws, guidance, n_per, B);
    DA_CHECK_LAUNCH();
    DA_LAUNCH((cfg_rescale_apply_kernel<uint16_t>), ew_grid(n), dim3(256), 0, s, (const uint16_t*)eps, (uint16_t*)out, ratio_ws,
              guidance, guidance_rescale, omg, n_per, n);
  } else if (dtype == DA_DTYPE_F32) {
    DA_LAUNCH((cfg_rescale_stats_kernel<float>), dim3(B), dim3(1024), 0, s, (const float*)eps, ratio_ws, guidance, n_per, B);
    DA_CHECK_LAUNCH();
    DA_LAUNCH((cfg_rescale_apply_kernel<float>), ew_grid(n), dim3(256), 0, s, (const float*)eps, (float*)out, ratio_ws, guidance,
              guidance_rescale, omg, n_per, n);
  } else {
    return DA_ERR_UNSUPPORTED;
  }
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_advance_step(int* step_idx, void* stream) {
  if (!step_idx) return DA_ERR_INVALID;
  DA_LAUNCH(advance_step_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_idx);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_mul_scalar(const void* x, void* out, float sc, int rep, long long n_, int dtype, void* stream) {
  if (!x || !out || n_ <= 0 || rep <= 0) return DA_ERR_INVALID;
  const size_t n = (size_t)n_;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DA_DTYPE_BF16)
    DA_LAUNCH((mul_scalar_kernel<uint16_t>), ew_grid(n), dim3(256), 0, s, (const uint16_t*)x, (uint16_t*)out, sc, rep, n);
  else if (dtype == DA_DTYPE_F32)
    DA_LAUNCH((mul_scalar_kernel<float>), ew_grid(n), dim3(256), 0, s, (const float*)x, (float*)out, sc, rep, n);
  else
    return DA_ERR_UNSUPPORTED;
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_unipc_flow_step(const void* v, void* x, void* last, void* m1, void* m2, const float* coef,
                                  const int* step_idx, int cfg, float guidance, long long n_, int x_dtype, int v_dtype,
                                  void* stream) {
  if (!v || !x || !last || !m1 || !m2 || !coef || !step_idx || n_ <= 0) return DA_ERR_INVALID;
  const size_t n = (size_t)n_;
  hipStream_t s = (hipStream_t)stream;
#define DA_UP(TX, TV, C)                                                                                       \
  DA_LAUNCH((unipc_flow_step_kernel<TX, TV, C>), ew_grid(n), dim3(256), 0, s, (const TV*)v, (TX*)x, (TX*)last, \
            (TX*)m1, (TX*)m2, coef, step_idx, guidance, n)
  if (x_dtype == DA_DTYPE_F32 && v_dtype == DA_DTYPE_F32) { if (cfg) DA_UP(float, float, true); else DA_UP(float, float, false); }
  else if (x_dtype == DA_DTYPE_F32 && v_dtype == DA_DTYPE_BF16) { if (cfg) DA_UP(float, uint16_t, true); else DA_UP(float, uint16_t, false); }
  else if (x_dtype == DA_DTYPE_BF16 && v_dtype == DA_DTYPE_BF16) { if (cfg) DA_UP(uint16_t, uint16_t, true); else DA_UP(uint16_t, uint16_t, false); }
  else return DA_ERR_UNSUPPORTED;
#undef DA_UP
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_cast_f32_bf16(const float* x, void* out, int rep, long long n_, void* stream) {
  if (!x || !out || n_ <= 0 || rep <= 0) return DA_ERR_INVALID;
  const size_t n = (size_t)n_;
  DA_LAUNCH(cast_f32_bf16_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)out, rep, n);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
