// The drop-in boundary without Python or torch: plain HIP memory + the C ABI of include/diffusers_amd.h.
//
//   hipcc --offload-arch=gfx950 -Iinclude examples/abi_demo.cpp -Ldiffusers_amd/_C -ldiffusers_amd \
//         -Wl,-rpath,$PWD/diffusers_amd/_C -o abi_demo && ./abi_demo
//
// 1. y = x W^T + b through da_gemm_bf16 (the F.linear call sites of the reference), checked
//    against a host loop;  2. one fused classifier-free-guidance + EulerDiscrete step through da_euler_step
//    (scheduling_euler_discrete.py:685-800 + pipeline_stable_diffusion_xl.py:1223-1225), checked the same way.
// 3. the same two calls once more as a launch PLAN built in C++ (da_plan_create / da_plan_launch): identical bytes.
// 4. `./abi_demo step.daplan`: a plan recorded by the Python side (diffusers_amd/plan.py, Plan.save: tools/make_plan_demo.py
//    writes one U-Net denoising step + one VAE decode of a small SDXL-shaped model) -- the file carries the ops with their
//    parameter structs, the contents of every device region they touch and the expected outputs; this program allocates the
//    regions with hipMalloc, moves the plan's addresses onto them (da_plan_relocate), launches and compares bit for bit.
//    A whole model step with no Python and no torch in the process.
// Exit code 0 = everything matches.  tests/test_abi_and_host.py compiles and links this file on every CPU run (no GPU needed
// for that); running it needs an MI355X.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "diffusers_amd.h"

static uint16_t f2bf(float f) {  // round to nearest even
  uint32_t u;
  std::memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
#define HIP_OK(x)                                                              \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
      return 2;                                                                \
    }                                                                          \
  } while (0)

// ---- a plan file written by diffusers_amd.plan.Plan.save --------------------------------------------------------------------
static bool read_exact(std::FILE* f, void* dst, size_t n) { return n == 0 || std::fread(dst, 1, n, f) == n; }

static int run_plan_file(const char* path) {
  std::FILE* f = std::fopen(path, "rb");
  if (!f) {
    std::fprintf(stderr, "cannot open %s\n", path);
    return 2;
  }
  char magic[8];
  uint32_t n_regions, n_ops, n_outputs;
  if (!read_exact(f, magic, 8) || std::memcmp(magic, "DAPLAN01", 8) != 0 || !read_exact(f, &n_regions, 4) ||
      !read_exact(f, &n_ops, 4) || !read_exact(f, &n_outputs, 4)) {
    std::fprintf(stderr, "%s: not a plan file\n", path);
    return 2;
  }
  // device regions: the recording process's base address, the size, the contents the replay starts from
  std::vector<const void*> old_base(n_regions);
  std::vector<unsigned long long> bytes(n_regions);
  std::vector<void*> new_base(n_regions);
  std::vector<char> host;
  unsigned long long total = 0;
  for (uint32_t r = 0; r < n_regions; ++r) {
    uint64_t base, n;
    if (!read_exact(f, &base, 8) || !read_exact(f, &n, 8)) return 2;
    host.resize(n);
    if (!read_exact(f, host.data(), n)) return 2;
    old_base[r] = (const void*)(uintptr_t)base, bytes[r] = n, total += n;
    HIP_OK(hipMalloc(&new_base[r], n));
    HIP_OK(hipMemcpy(new_base[r], host.data(), n, hipMemcpyHostToDevice));
  }
  // ops; the parameter structs / host arrays an op points at follow it as blobs, in argument order
  std::vector<da_plan_op> ops(n_ops);
  std::vector<std::unique_ptr<char[]>> blobs;
  for (uint32_t o = 0; o < n_ops; ++o) {
    int32_t fn, nargs;
    if (!read_exact(f, &fn, 4) || !read_exact(f, &nargs, 4) || !read_exact(f, ops[o].arg, sizeof ops[o].arg)) return 2;
    ops[o].fn = fn, ops[o].reserved = 0;
    const char* kinds = da_plan_arg_kinds(fn);
    if (!kinds || (int)std::strlen(kinds) != nargs) {
      std::fprintf(stderr, "op %u: entry point %d is not one this library replays\n", o, fn);
      return 1;
    }
    for (int i = 0; kinds[i]; ++i)
      if (std::strchr("GAIQ", kinds[i])) {
        uint32_t n;
        if (!read_exact(f, &n, 4)) return 2;
        const size_t want = kinds[i] == 'G' ? sizeof(da_gemm_params) : kinds[i] == 'A' ? sizeof(da_attention_params) : n;
        if (n != want) {
          std::fprintf(stderr, "op %u: a %u-byte parameter struct where this header has %zu (ABI mismatch)\n", o, n, want);
          return 1;
        }
        blobs.emplace_back(new char[n ? n : 1]);
        if (!read_exact(f, blobs.back().get(), n)) return 2;
        ops[o].arg[i] = (unsigned long long)(uintptr_t)blobs.back().get();
      }
  }
  da_plan* plan = nullptr;
  int rc = da_plan_create(ops.data(), (int)n_ops, &plan);
  if (rc != DA_OK) {
    std::fprintf(stderr, "da_plan_create: %d\n", rc);
    return 1;
  }
  blobs.clear();  // the plan owns copies
  int unmatched = 0;
  rc = da_plan_relocate(plan, (int)n_regions, old_base.data(), bytes.data(), new_base.data(), &unmatched);
  if (rc != DA_OK || unmatched) {
    std::fprintf(stderr, "da_plan_relocate: %d, %d addresses outside every region\n", rc, unmatched);
    return 1;
  }
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  int failed = -1;
  rc = da_plan_launch(plan, stream, &failed);
  if (rc != DA_OK) {
    std::fprintf(stderr, "da_plan_launch: op %d failed with %d (%s)\n", failed, rc, da_last_error());
    return 1;
  }
  HIP_OK(hipStreamSynchronize(stream));
  std::printf("plan %s: %d launches over %u device regions (%.1f MB)\n", path, da_plan_op_count(plan), n_regions, total / 1e6);
  // expected outputs: (address in the recording process, bytes, contents)
  int bad = 0;
  std::vector<char> got;
  for (uint32_t k = 0; k < n_outputs; ++k) {
    uint64_t ptr, n;
    if (!read_exact(f, &ptr, 8) || !read_exact(f, &n, 8)) return 2;
    host.resize(n), got.resize(n);
    if (!read_exact(f, host.data(), n)) return 2;
    const char* dev = nullptr;
    for (uint32_t r = 0; r < n_regions; ++r) {
      const uintptr_t b = (uintptr_t)old_base[r];
      if (ptr >= b && ptr - b < bytes[r]) dev = (const char*)new_base[r] + (ptr - b);
    }
    if (!dev) {
      std::fprintf(stderr, "output %u lies in no region\n", k);
      return 1;
    }
    HIP_OK(hipMemcpy(got.data(), dev, n, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < n; ++i) diff += got[i] != host[i];
    std::printf("  output %u: %llu bytes, %zu differ from the Python-driven run\n", k, (unsigned long long)n, diff);
    bad += diff != 0;
  }
  std::fclose(f);
  da_plan_destroy(plan);
  for (void* p : new_base) HIP_OK(hipFree(p));
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  std::printf("libdiffusers_amd ABI version %d (header %d)\n", da_version(), DA_ABI_VERSION);
  if (da_version() != DA_ABI_VERSION || da_sizeof_gemm_params() != sizeof(da_gemm_params) ||
      da_sizeof_attention_params() != sizeof(da_attention_params)) {
    std::fprintf(stderr, "ABI mismatch: the library was built from another revision of include/diffusers_amd.h\n");
    return 3;
  }
  if (argc > 1) return run_plan_file(argv[1]);
  const int M = 200, N = 128, K = 256;
  std::vector<uint16_t> x(M * K), w(N * K), b(N), y(M * N);
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : x) v = f2bf(rnd());
  for (auto& v : w) v = f2bf(rnd() / 16.0f);
  for (auto& v : b) v = f2bf(rnd());
  void *dx, *dw, *db, *dy;
  HIP_OK(hipMalloc(&dx, x.size() * 2));
  HIP_OK(hipMalloc(&dw, w.size() * 2));
  HIP_OK(hipMalloc(&db, b.size() * 2));
  HIP_OK(hipMalloc(&dy, y.size() * 2));
  HIP_OK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));

  da_gemm_params p;
  std::memset(&p, 0, sizeof p);
  p.A = dx, p.W = dw, p.C = dy, p.bias = db;
  p.M = M, p.N = N, p.K = K, p.lda = K, p.ldw = K, p.ldc = N;
  p.alpha = 1.0f, p.out_scale = 1.0f, p.act = DA_ACT_NONE;
  p.tile = DA_TILE_AUTO, p.staging = DA_STAGE_LDS_DIRECT;
  int rc = da_gemm_bf16(&p, stream);
  if (rc != DA_OK) {
    std::fprintf(stderr, "da_gemm_bf16: %d (%s)\n", rc, da_last_error());
    return 1;
  }
  HIP_OK(hipStreamSynchronize(stream));
  HIP_OK(hipMemcpy(y.data(), dy, y.size() * 2, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      float acc = bf2f(b[n]);
      for (int k = 0; k < K; ++k) acc += bf2f(x[m * K + k]) * bf2f(w[n * K + k]);
      worst = std::fmax(worst, std::fabs(bf2f(y[m * N + n]) - acc) / (1.0 + std::fabs(acc)));
    }
  std::printf("linear %dx%dx%d: worst relative error %.3e\n", M, N, K, worst);
  if (!(worst < 1e-2)) return 1;

  // fused CFG + Euler step on fp32 latents: table row = {sigma, sigma_next, dt, sqrt(sigma^2 + 1), -, -, -, timestep}
  const int n = 4 * 64 * 64;
  std::vector<float> lat(n), eps2(2 * n), out(n), table(8, 0.f);
  for (auto& v : lat) v = rnd();
  for (auto& v : eps2) v = rnd();
  table[0] = 14.6f, table[1] = 12.9f, table[2] = table[1] - table[0];
  void *dl, *de, *dt;
  int* dstep;
  HIP_OK(hipMalloc(&dl, n * 4));
  HIP_OK(hipMalloc(&de, 2 * n * 4));
  HIP_OK(hipMalloc(&dt, 8 * 4));
  HIP_OK(hipMalloc((void**)&dstep, 4));
  HIP_OK(hipMemcpy(dl, lat.data(), n * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(de, eps2.data(), 2 * n * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dt, table.data(), 8 * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemset(dstep, 0, 4));
  const float g = 5.0f;
  rc = da_euler_step(de, dl, dl, (const float*)dt, dstep, /*cfg=*/1, g, n, DA_DTYPE_F32, DA_PRED_EPSILON, stream);
  if (rc != DA_OK) {
    std::fprintf(stderr, "da_euler_step: %d\n", rc);
    return 1;
  }
  HIP_OK(hipStreamSynchronize(stream));
  HIP_OK(hipMemcpy(out.data(), dl, n * 4, hipMemcpyDeviceToHost));
  worst = 0;
  for (int i = 0; i < n; ++i) {
    const float e = eps2[i] + g * (eps2[n + i] - eps2[i]);              // uncond + g (cond - uncond)
    const float want = lat[i] + e * table[2];                            // x + derivative * dt, derivative = eps
    worst = std::fmax(worst, std::fabs(out[i] - want));
  }
  std::printf("euler CFG step on %d elements: worst absolute error %.3e\n", n, worst);
  if (!(worst < 1e-3)) return 1;

  // the same two launches as a plan: reset the inputs, replay, compare with what the direct calls left behind
  da_plan_op ops[2];
  std::memset(ops, 0, sizeof ops);
  ops[0].fn = DA_FN_GEMM;
  ops[0].arg[0] = (unsigned long long)(uintptr_t)&p;
  ops[1].fn = DA_FN_EULER_STEP;
  {
    unsigned long long* a = ops[1].arg;
    uint32_t gbits;
    std::memcpy(&gbits, &g, 4);
    a[0] = (uintptr_t)de, a[1] = (uintptr_t)dl, a[2] = (uintptr_t)dl, a[3] = (uintptr_t)dt, a[4] = (uintptr_t)dstep;
    a[5] = 1, a[6] = gbits, a[7] = (unsigned long long)n, a[8] = DA_DTYPE_F32, a[9] = DA_PRED_EPSILON;
  }
  da_plan* plan = nullptr;
  rc = da_plan_create(ops, 2, &plan);
  if (rc != DA_OK || da_plan_op_count(plan) != 2) {
    std::fprintf(stderr, "da_plan_create: %d\n", rc);
    return 1;
  }
  std::memset(&p, 0, sizeof p);  // the plan holds its own copy of the struct
  HIP_OK(hipMemset(dy, 0, y.size() * 2));
  HIP_OK(hipMemcpy(dl, lat.data(), n * 4, hipMemcpyHostToDevice));
  int failed = -1;
  rc = da_plan_launch(plan, stream, &failed);
  if (rc != DA_OK) {
    std::fprintf(stderr, "da_plan_launch: op %d failed with %d\n", failed, rc);
    return 1;
  }
  HIP_OK(hipStreamSynchronize(stream));
  std::vector<uint16_t> y2(y.size());
  std::vector<float> out2(n);
  HIP_OK(hipMemcpy(y2.data(), dy, y2.size() * 2, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(out2.data(), dl, n * 4, hipMemcpyDeviceToHost));
  const bool same = std::memcmp(y2.data(), y.data(), y.size() * 2) == 0 && std::memcmp(out2.data(), out.data(), n * 4) == 0;
  std::printf("plan of the two launches: %s\n", same ? "identical bytes" : "DIFFERS");
  da_plan_destroy(plan);
  return same ? 0 : 1;
}
