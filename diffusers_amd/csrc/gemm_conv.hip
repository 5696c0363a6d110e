// nn.Conv2d (1x1 / 3x3, stride 1|2, fused nearest-2x upsample and channel concat) instantiations of the implicit-GEMM
// kernel in gemm_kernel.cuh; kept in their own translation unit so they compile in parallel with the nn.Linear ones.
#include "gemm_kernel.cuh"

namespace da_gemm {
int dispatch_conv(const da_gemm_params& p, int tile, int staging, hipStream_t s) { return dispatch<true>(p, tile, staging, s); }
}  // namespace da_gemm
