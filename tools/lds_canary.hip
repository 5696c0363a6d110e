// Round 6 debug: does any kernel of a VAE decode / U-Net step write LDS outside its own allocation?  A workgroup of ANOTHER kernel that
// shares the CU would see it -- which only happens when two streams run at once (the open item of pipelines._exclusive_decode).
//   canary kernel  : every workgroup fills `kb` KiB of dynamic LDS with a pattern, then re-reads it for `ticks` of the 100 MHz wall
//                    clock; mismatches are counted (bad[0]) and the first one is described (bad[1..4] = block, word, got, expected).
//   overflow kernel: POSITIVE CONTROL -- a workgroup with 1 KiB of LDS issues LDS-DMA pieces (buffer_load ... lds, the instruction every
//                    GEMM / attention ring of the library is filled with) and plain ds_write at byte offsets far beyond its allocation:
//                    does the hardware bound them to the workgroup's allocation, or do they land in a neighbour's LDS?
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/lds_canary.hip -o /tmp/liblds_canary.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void lds_canary_kernel(unsigned* bad, long long ticks, int words) {
  extern __shared__ unsigned lds[];
  const unsigned salt = blockIdx.x * 2654435761u;
  for (int i = threadIdx.x; i < words; i += 256) lds[i] = (unsigned)i * 40503u ^ salt;
  __syncthreads();
  const long long t0 = wall_clock64();
  do {
    for (int i = threadIdx.x; i < words; i += 256) {
      const unsigned want = (unsigned)i * 40503u ^ salt, got = lds[i];
      if (got != want) {
        if (atomicAdd(&bad[0], 1u) == 0) {
          bad[1] = blockIdx.x, bad[2] = (unsigned)i, bad[3] = got, bad[4] = want;
        }
        lds[i] = want;
      }
    }
    __builtin_amdgcn_s_sleep(32);
  } while (wall_clock64() - t0 < ticks);
}

__global__ __launch_bounds__(64) void lds_overflow_kernel(const uint32_t* src, int byte_off, int use_dma, long long ticks) {
  extern __shared__ unsigned char small[];   // 1 KiB
  const long long t0 = wall_clock64();
  do {
    if (use_dma) {
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4096, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(small + byte_off), 16, threadIdx.x * 16, 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      const unsigned addr = (unsigned)(size_t)(small + byte_off) + threadIdx.x * 4;
      const unsigned v = 0xdeadbeefu;
      asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(v) : "memory");
    }
    __builtin_amdgcn_s_sleep(8);
  } while (wall_clock64() - t0 < ticks);
}

extern "C" int lds_canary_launch(void* bad, int blocks, int kib, long long ticks, void* stream) {
  static bool set = false;
  if (!set) {
    if (hipFuncSetAttribute((const void*)lds_canary_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
    set = true;
  }
  hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(256), kib * 1024, (hipStream_t)stream, (unsigned*)bad, ticks, kib * 256);
  return (int)hipGetLastError();
}
extern "C" int lds_overflow_launch(const void* src, int blocks, int byte_off, int use_dma, long long ticks, void* stream) {
  hipLaunchKernelGGL(lds_overflow_kernel, dim3(blocks), dim3(64), 1024, (hipStream_t)stream, (const uint32_t*)src, byte_off, use_dma, ticks);
  return (int)hipGetLastError();
}
