// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of diffusers_amd.
// Written for wave64 + MFMA only; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "diffusers_amd.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;


// hipGetLastError() is per-thread and sticky: PyTorch's allocator leaves benign hipErrorNotReady results from
// hipEventQuery behind, so the error slot is cleared right before every launch and read right after it.
// da_set_launch_events() (version.hip): a caller may arm a start / stop event pair for this thread's next launches; such a launch
// goes through hipExtLaunchKernelGGL, which stamps the events with the dispatch's OWN begin / end (what rocprofv3 reports as the
// kernel's duration) instead of the completion of separate marker packets either side of it.  One thread-local load otherwise.
// da_set_launch_flags() (version.hip, experiments only): the flags word of hipExtLaunchKernelGGL for this thread's launches
// (hipExtAnyOrderLaunch = the dispatch packet without its barrier bit); da_take_launch_events() reports it through *flags.
extern "C" int da_take_launch_events(hipEvent_t* start, hipEvent_t* stop, unsigned* flags);
template <typename F, typename... Args>
inline void da_launch_with_events(hipEvent_t start, hipEvent_t stop, unsigned flags, F kernel, const dim3& grid, const dim3& block,
                                  uint32_t shmem, hipStream_t stream, Args... args) {
  hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, start, stop, flags, args...);
}
#define DA_LAUNCH(...)                                      \
  do {                                                      \
    (void)hipGetLastError();                                \
    hipEvent_t es__ = nullptr, ee__ = nullptr;              \
    unsigned fl__ = 0;                                      \
    if (__builtin_expect(da_take_launch_events(&es__, &ee__, &fl__), 0)) \
      da_launch_with_events(es__, ee__, fl__, __VA_ARGS__); \
    else                                                    \
      hipLaunchKernelGGL(__VA_ARGS__);                      \
  } while (0)

extern "C" void da_set_last_error(int hip_error);  // version.hip: remembered for da_last_error()
#define DA_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) {                                \
      da_set_last_error((int)e__);                          \
      return DA_ERR_LAUNCH;                                 \
    }                                                       \
  } while (0)

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even fp32 -> bf16 (lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}
// two fp32 -> one packed bf16 pair, round-to-nearest-even: ONE v_cvt_pk_bf16_f32 (converting the halves separately and
// OR-ing them costs four instructions)
typedef __bf16 da_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float da_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const da_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, da_bf16x2_t));
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// unpack 8 bf16 held in a uint4 into 8 floats
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bf_lo(v.x); f[1] = bf_hi(v.x);
  f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
  f[4] = bf_lo(v.z); f[5] = bf_hi(v.z);
  f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]);
  v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
  return v;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// erf-GELU (torch F.gelu default, approximate="none"): 0.5 x (1 + erf(x / sqrt 2)).
// erf through Abramowitz & Stegun 7.1.26 in its erfc form, erfc(z) = (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z), z >= 0:
// 1 + erf(x / sqrt 2) = E for x < 0 and 2 - E for x >= 0 with E = erfc(|x| / sqrt 2).  Branch-free, 15 VALU instructions
// (v_rcp_f32 + v_exp_f32 + 8 fma / mul) against ~50 for the library erff with both of its range branches taken under
// divergence -- the GEGLU epilogue of the SDXL feed-forward evaluates it 10.5 M times per launch and was 21 us of an 84 us
// kernel (profiles/r03b_k2_microbench.md).  Accuracy: the erfc form keeps a RELATIVE error <= 2e-4 for |x| <= 3.5 and
// <= 4e-3 down to x = -6 (|gelu| < 1e-6 there), absolute <= 5e-7 everywhere; over all 33 314 bf16 inputs in [-9, 9] the
// bf16-rounded result differs from the exactly rounded fp64 GELU on 96 inputs (all with |gelu| < 4e-3, 73 of them below
// 5e-9), torch's own fp32 erf path on 129.
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(t, poly, 1.421413741f);
  poly = fmaf(t, poly, -0.284496736f);
  poly = fmaf(t, poly, 0.254829592f);
  const float e = (poly * t) * __builtin_amdgcn_exp2f((x * x) * (-0.5f * 1.4426950408889634f));
  return (0.5f * x) * (x >= 0.0f ? 2.0f - e : e);
}
// tanh-GELU (torch approximate="tanh"): 0.5 x (1 + tanh(u)), u = sqrt(2 / pi) (x + 0.044715 x^3), evaluated as x * sigmoid(2 u) =
// x / (1 + 2^(-2 u log2 e)) -- the same function (1 + tanh u = 2 / (1 + e^(-2u))) in 8 VALU instructions (v_exp_f32 + v_rcp_f32 + 6
// fma / mul) against 37 for the library tanhf with its range branches: the epilogue of Flux's proj_mlp evaluates it 56.6 M times
// per launch with the matrix pipe idle (round 5).  Over all 33 762 bf16 inputs in [-30, 30] the bf16-rounded result differs from the
// exactly rounded fp64 GELU on 72 inputs, the tanhf form (and torch's own fp32 kernel) on 92.  Large |x|: 2^(+big) = inf ->
// x * 0 = -0 (exact value: a denormal-sized negative), 2^(-big) = 0 -> x.
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float u = k0 * (x + k1 * x * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * (-2.0f * 1.4426950408889634f)));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
