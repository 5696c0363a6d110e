#include "common.cuh"

static thread_local int g_last_hip_error = 0;

extern "C" int da_version(void) { return DA_ABI_VERSION; }
extern "C" size_t da_sizeof_gemm_params(void) { return sizeof(da_gemm_params); }
extern "C" size_t da_sizeof_attention_params(void) { return sizeof(da_attention_params); }
extern "C" void da_set_last_error(int hip_error) { g_last_hip_error = hip_error; }
// Name of the HIP runtime error behind the calling thread's most recent DA_ERR_LAUNCH ("hipSuccess" if none).
extern "C" const char* da_last_error(void) { return hipGetErrorName((hipError_t)g_last_hip_error); }

// ---- dispatch-attached timing events (bench.py's roofline legs) ------------------------------------------------------------------
// The next launch of this thread takes `start` (then it is disarmed); every launch until the pair is cleared takes `stop`, so an
// entry point that issues several kernels is bracketed from the begin of its first to the end of its last dispatch.
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;

extern "C" int da_set_launch_events(void* start_event, void* stop_event) {
  if ((start_event == nullptr) != (stop_event == nullptr)) return DA_ERR_INVALID;
  g_ev_start = (hipEvent_t)start_event;
  g_ev_stop = (hipEvent_t)stop_event;
  return DA_OK;
}

static thread_local unsigned g_launch_flags = 0;
extern "C" int da_set_launch_flags(unsigned flags) {
  g_launch_flags = flags;
  return DA_OK;
}

extern "C" int da_take_launch_events(hipEvent_t* start, hipEvent_t* stop, unsigned* flags) {
  if (!g_ev_stop && !g_launch_flags) return 0;
  *start = g_ev_start;
  *stop = g_ev_stop;
  *flags = g_launch_flags;
  g_ev_start = nullptr;
  return 1;
}

// ---- box normaliser (bench.py `config.box_mfma_tflops`) ---------------------------------------------------------------------------
// A register-only MFMA loop: four independent v_mfma_f32_32x32x16_bf16 accumulator chains per wave, four waves per SIMD, no memory
// traffic inside the loop.  Its rate is (matrix-pipe issue rate) x (the clock this box sustains under matrix load) and nothing
// else, so the quotient of two boxes' rates is the quotient of their sustained clocks: lines measured on different boxes of the
// pool (+- 8 % on one build) can be normalised with it.  It is NOT a roofline: 2.5 PFLOP/s stays the peak of every fraction.
__global__ __launch_bounds__(256) void mfma_probe_kernel(int iters, float* sink) {
  const int lane = threadIdx.x & 63;
  bf16x8_t a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(float)((lane + i) & 3);
    b[i] = (__bf16)(float)((lane ^ i) & 1);
  }
  f32x16_t c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == -1.0f) sink[0] = s;       // never true (all terms are >= 0): keeps the chains alive without a store
}

// Launches the probe: `blocks` workgroups of four waves, `iters` x 16 MFMAs per wave.  *flop (may be NULL) receives the launch's
// floating-point work, 2 * 32 * 32 * 16 per MFMA.  `sink` is any device address of >= 4 bytes (never written).
extern "C" int da_mfma_probe(int blocks, int iters, void* sink, double* flop, void* stream) {
  if (blocks <= 0 || iters <= 0 || !sink) return DA_ERR_INVALID;
  if (flop) *flop = (double)blocks * 4.0 * (double)iters * 16.0 * 2.0 * 32.0 * 32.0 * 16.0;
  DA_LAUNCH(mfma_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, (float*)sink);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
