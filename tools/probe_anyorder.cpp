// Round 6 experiment: what would a dispatch that does NOT wait for its predecessor buy on the chains of small launches an SDXL step
// is made of?  Two questions, one program (plain HIP + the C ABI, no torch):
//
//  1. Does gfx950 honour hipExtAnyOrderLaunch (the AQL packet without its barrier bit)?  A one-workgroup kernel spins 200 us on the
//     device clock; a second kernel launched behind it on the SAME stream stamps the clock.  Stamp < end of the spin = overlapped.
//  2. Upper bound on the overlap: chains of da_gemm_bf16 launches of one SDXL projection shape over more weights than the memory-side
//     cache holds (every weight cold, as inside a step), (a) in order on one stream, (b) with the any-order flag, (c) dealt round-robin
//     to 2 / 4 streams.  (b) and (c) enforce NO dependency -- every launch reads the same x and writes its own y -- so the
//     difference to (a) is everything a software-enforced dependency (arrival counters) could hope to recover, and no more.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/probe_anyorder.cpp -Ldiffusers_amd/_C -ldiffusers_amd \
//         -Wl,-rpath,$PWD/diffusers_amd/_C -o /tmp/probe_anyorder && /tmp/probe_anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "diffusers_amd.h"

#define HIP_OK(x)                                                              \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
      return 2;                                                                \
    }                                                                          \
  } while (0)

__global__ void spin_kernel(long long ticks, long long* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  out[0] = t0;
  out[1] = wall_clock64();
}
__global__ void stamp_kernel(long long* out) { out[2] = wall_clock64(); }
__global__ void fill_kernel(uint16_t* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u + seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    // bf16 in [-1/32, 1/32): sign | exponent 120..121 | mantissa
    p[i] = (uint16_t)(((h & 1u) << 15) | ((120u + ((h >> 1) & 1u)) << 7) | ((h >> 2) & 0x7fu));
  }
}

struct Shape {
  const char* name;
  int M, N, K, tile, staging, nw, residual;
};

int main() {
  HIP_OK(hipSetDevice(0));
  int clk_khz = 0;
  HIP_OK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
  std::printf("{\"wall_clock_khz\": %d}\n", clk_khz);
  hipStream_t s[4];
  for (auto& x : s) HIP_OK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));

  // ---- 1. any-order on one stream ----
  long long* stamps;
  HIP_OK(hipMalloc(&stamps, 64));
  for (unsigned flag = 0; flag < 2; ++flag) {
    HIP_OK(hipMemset(stamps, 0, 64));
    const long long ticks = (long long)clk_khz * 200 / 1000;  // 200 us
    hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s[0], nullptr, nullptr, 0, ticks, stamps);
    hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s[0], nullptr, nullptr, flag, stamps);
    HIP_OK(hipStreamSynchronize(s[0]));
    long long h[3];
    HIP_OK(hipMemcpy(h, stamps, 24, hipMemcpyDeviceToHost));
    std::printf("{\"test\": \"anyorder_same_stream\", \"flag\": %u, \"spin_us\": %.1f, \"stamp_minus_spin_end_us\": %.1f, \"overlapped\": %s}\n", flag,
                (h[1] - h[0]) * 1e3 / clk_khz, (h[2] - h[1]) * 1e3 / clk_khz, h[2] < h[1] ? "true" : "false");
  }

  // ---- 2. chains ----
  const Shape shapes[] = {
      {"to_out 2048x1280x1280 +res", 2048, 1280, 1280, DA_TILE_K2_128x80, DA_STAGE_PINGPONG, 120, 1},
      {"ff_down 2048x1280x5120 +res", 2048, 1280, 5120, DA_TILE_K2_128x80, DA_STAGE_PINGPONG3, 40, 1},
      {"to_out 8192x640x640 +res", 8192, 640, 640, DA_TILE_K2_128x160, DA_STAGE_PINGPONG, 400, 1},
  };
  for (const Shape& sh : shapes) {
    uint16_t *x, *r, *b;
    std::vector<uint16_t*> w(sh.nw), y(8);
    HIP_OK(hipMalloc(&x, (size_t)sh.M * sh.K * 2));
    HIP_OK(hipMalloc(&r, (size_t)sh.M * sh.N * 2));
    HIP_OK(hipMalloc(&b, (size_t)sh.N * 2));
    fill_kernel<<<1024, 256>>>(x, (size_t)sh.M * sh.K, 1);
    fill_kernel<<<1024, 256>>>(r, (size_t)sh.M * sh.N, 2);
    fill_kernel<<<16, 256>>>(b, (size_t)sh.N, 3);
    for (int i = 0; i < sh.nw; ++i) {
      HIP_OK(hipMalloc(&w[i], (size_t)sh.N * sh.K * 2));
      fill_kernel<<<1024, 256>>>(w[i], (size_t)sh.N * sh.K, 10 + i);
    }
    for (auto& p : y) HIP_OK(hipMalloc(&p, (size_t)sh.M * sh.N * 2));
    HIP_OK(hipDeviceSynchronize());
    std::vector<da_gemm_params> ps(sh.nw);
    for (int i = 0; i < sh.nw; ++i) {
      da_gemm_params p;
      std::memset(&p, 0, sizeof p);
      p.A = x;
      p.W = w[i];
      p.C = y[i % 8];
      p.bias = b;
      p.residual = sh.residual ? r : nullptr;
      p.M = sh.M, p.N = sh.N, p.K = sh.K;
      p.lda = sh.K, p.ldw = sh.K, p.ldc = sh.N, p.ldr = sh.N;
      p.rows_per_batch = sh.M;
      p.tile = sh.tile, p.staging = sh.staging;
      p.prefetch = w[(i + 1) % sh.nw];
      p.prefetch_bytes = (long long)sh.N * sh.K * 2;
      ps[i] = p;
    }
    hipEvent_t e0, e1, fork, join[4];
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (auto& j : join) HIP_OK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
    // mode: 0 in order; 1 any-order flag; 2 / 4 = round robin over that many streams
    for (int rep = 0; rep < 2; ++rep) {
      for (int mode : {0, 1, 2, 4, 0}) {
        const int ns = mode >= 2 ? mode : 1;
        const int reps = 3;
        da_set_launch_flags(0);
        int rc = da_gemm_bf16(&ps[0], s[0]);  // warm (code object, tile check)
        if (rc != DA_OK) {
          std::fprintf(stderr, "da_gemm_bf16 rc %d (%s)\n", rc, da_last_error());
          return 3;
        }
        HIP_OK(hipDeviceSynchronize());
        const auto h0 = std::chrono::steady_clock::now();
        HIP_OK(hipEventRecord(e0, s[0]));
        if (ns > 1) {
          HIP_OK(hipEventRecord(fork, s[0]));
          for (int k = 1; k < ns; ++k) HIP_OK(hipStreamWaitEvent(s[k], fork, 0));
        }
        da_set_launch_flags(mode == 1 ? 1u : 0u);
        for (int q = 0; q < reps; ++q)
          for (int i = 0; i < sh.nw; ++i) {
            rc = da_gemm_bf16(&ps[i], s[ns > 1 ? i % ns : 0]);
            if (rc != DA_OK) return 3;
          }
        da_set_launch_flags(0);
        const auto h1 = std::chrono::steady_clock::now();
        for (int k = 1; k < ns; ++k) {
          HIP_OK(hipEventRecord(join[k], s[k]));
          HIP_OK(hipStreamWaitEvent(s[0], join[k], 0));
        }
        HIP_OK(hipEventRecord(e1, s[0]));
        HIP_OK(hipEventSynchronize(e1));
        HIP_OK(hipDeviceSynchronize());
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        const double host_us = std::chrono::duration<double, std::micro>(h1 - h0).count() / (reps * sh.nw);
        std::printf("{\"test\": \"chain\", \"shape\": \"%s\", \"mode\": \"%s\", \"rep\": %d, \"us_per_launch\": %.2f, \"host_issue_us_per_launch\": %.2f}\n",
                    sh.name, mode == 0 ? "in order" : mode == 1 ? "any-order flag" : mode == 2 ? "2 streams" : "4 streams", rep,
                    ms * 1e3 / (reps * sh.nw), host_us);
        std::fflush(stdout);
      }
    }
    for (auto p : w) (void)hipFree(p);
    for (auto p : y) (void)hipFree(p);
    (void)hipFree(x), (void)hipFree(r), (void)hipFree(b);
  }
  return 0;
}
