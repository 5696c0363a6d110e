// nn.Conv2d instantiations of the K2 implicit-GEMM kernel (gemm2_kernel.cuh), in their own translation unit so they compile
// in parallel with the nn.Linear ones.
#include "gemm2_kernel.cuh"

namespace da_gemm2 {
int dispatch_conv(const da_gemm_params& p, int tile, int staging, hipStream_t s) { return dispatch<true>(p, tile, staging, s); }
}  // namespace da_gemm2
