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

extern "C" int da_take_launch_events(hipEvent_t* start, hipEvent_t* stop) {
  if (!g_ev_stop) return 0;
  *start = g_ev_start;
  *stop = g_ev_stop;
  g_ev_start = nullptr;
  return 1;
}
