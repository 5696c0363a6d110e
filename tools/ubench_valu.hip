// VALU issue-rate microbenchmark (gfx950): cycles per wave64 instruction of v_fma_f32, v_exp_f32, v_rcp_f32, v_add_f32, v_cvt_pk_bf16_f32
// and v_max3_f32, for 1 / 2 / 3 / 4 waves per SIMD (256-thread blocks x 1..4 per CU), independent chains (8 accumulators per lane).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/_ubench_valu && tools/_ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 2048
template <int OP>
__global__ void k(float* out, unsigned long long* cyc, float seed) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < N_IT; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
      if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
      if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
      if (OP == 5) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
      if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(double*)&a[i & 6]));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 4 * 1024 * 4); hipMalloc(&cyc, 8 * 1024 * 8);
  for (int bpc = 1; bpc <= 4; ++bpc) {
    const int blocks = 256 * bpc;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks; ++i) m += h[i];
    m /= blocks;
    std::printf("%-22s %d wave(s)/SIMD: %.2f cycles per instruction per wave (%.2f per instruction per SIMD)\n", name, bpc, m / (N_IT * 8.0), m / (N_IT * 8.0) / bpc);
  }
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("v_fma_f32"); run<1>("v_exp_f32"); run<2>("v_rcp_f32"); run<3>("v_add_f32"); run<4>("v_cvt_pk_bf16_f32"); run<5>("v_max3_f32"); run<6>("v_pk_mul_f32");
  return 0;
}
