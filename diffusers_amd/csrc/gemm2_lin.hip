// nn.Linear instantiations of the K2 / K1 GEMM kernel (gemm2_kernel.cuh), in their own translation unit so they compile in
// parallel with the first family (gemm.hip) and with the nn.Conv2d ones (gemm2_conv.hip).
#include "gemm2_kernel.cuh"

namespace da_gemm2 {
int dispatch_lin(const da_gemm_params& p, int tile, int staging, hipStream_t s) { return dispatch<false>(p, tile, staging, s); }
}  // namespace da_gemm2
