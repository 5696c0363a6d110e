#include "common.cuh"

static thread_local int g_last_hip_error = 0;

extern "C" int da_version(void) { return 1; }
extern "C" void da_set_last_error(int hip_error) { g_last_hip_error = hip_error; }
// Name of the HIP runtime error behind the calling thread's most recent DA_ERR_LAUNCH ("hipSuccess" if none).
extern "C" const char* da_last_error(void) { return hipGetErrorName((hipError_t)g_last_hip_error); }
