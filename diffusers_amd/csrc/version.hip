#include "common.cuh"
extern "C" int da_version(void) { return 1; }
