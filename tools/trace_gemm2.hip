// Stage timeline of one K2 / K1 GEMM launch (csrc/gemm2_kernel.cuh built with -DDA_GEMM2_TRACE): where the "fixed" microseconds
// of a launch go.  Every wave keeps s_memtime at eight marks (entry, set-up done, ring issued, first pair landed, K loop done,
// partial sums exchanged, stores issued, stores acknowledged); this harness launches the kernel on random operands, reads the
// marks back and prints, per stage, the median / p90 over all waves in shader cycles plus the launch's wall picture (first
// entry -> last end).  Measurement tool only -- the library is never built with the trace.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDA_GEMM2_TRACE -Iinclude -Idiffusers_amd/csrc tools/trace_gemm2.hip -o tools/_trace_gemm2
//   tools/_trace_gemm2 M N K tile staging [bias_residual=1] [reps=5]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gemm2_kernel.cuh"

extern "C" void da_set_last_error(int) {}

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
      std::exit(1);                                                            \
    }                                                                          \
  } while (0)

static uint16_t bf(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int main(int argc, char** argv) {
  if (argc < 6) {
    std::fprintf(stderr, "usage: %s M N K tile staging [bias_residual] [reps]\n", argv[0]);
    return 2;
  }
  const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), tile = atoi(argv[4]), staging = atoi(argv[5]);
  const int epi = argc > 6 ? atoi(argv[6]) : 1, reps = argc > 7 ? atoi(argv[7]) : 5;
  std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K), hb(N), hr((size_t)M * N);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
  for (auto& v : ha) v = bf(rnd());
  for (auto& v : hw) v = bf(rnd() * 0.05f);
  for (auto& v : hb) v = bf(rnd());
  for (auto& v : hr) v = bf(rnd());
  uint16_t *A, *W, *B, *R, *C;
  CK(hipMalloc(&A, ha.size() * 2));
  CK(hipMalloc(&W, hw.size() * 2));
  CK(hipMalloc(&B, hb.size() * 2));
  CK(hipMalloc(&R, hr.size() * 2));
  CK(hipMalloc(&C, hr.size() * 2));
  CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(R, hr.data(), hr.size() * 2, hipMemcpyHostToDevice));
  const size_t max_waves = 8 * 8192;
  unsigned long long* tr;
  CK(hipMalloc(&tr, max_waves * 8 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(da_gemm2::g_da2_trace), &tr, sizeof(tr)));
  void* flush;
  const size_t flush_bytes = 320u << 20;
  CK(hipMalloc(&flush, flush_bytes));

  da_gemm_params p = {};
  p.A = A; p.W = W; p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.lda = K; p.ldw = K; p.ldc = N;
  p.alpha = 1.f; p.out_scale = 1.f;
  // epilogue modes: 0 plain, 1 bias + residual, 2 GEGLU (+ bias), 3 GEGLU with the LayerNorm fold, 4 bias + residual + statistics out,
  // 5 plain with the LayerNorm fold (to_q)
  float *lnstats = nullptr, *lns = nullptr, *lnc = nullptr, *stats_out = nullptr;
  if (epi == 1 || epi == 4) { p.bias = B; p.residual = R; p.ldr = N; }
  if (epi == 2 || epi == 3) { p.bias = B; p.act = DA_ACT_GEGLU; p.ldc = N / 2; }
  if (epi == 3 || epi == 5) {
    std::vector<float> hs((size_t)M * 128), hv(N, 0.01f);
    for (size_t i = 0; i < hs.size(); i += 2) { hs[i] = 0.1f * K / 16; hs[i + 1] = 1.0f * K / 16; }
    CK(hipMalloc(&lnstats, hs.size() * 4));
    CK(hipMalloc(&lns, N * 4));
    CK(hipMalloc(&lnc, N * 4));
    CK(hipMemcpy(lnstats, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(lns, hv.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(lnc, hv.data(), N * 4, hipMemcpyHostToDevice));
    p.ln_stats = lnstats; p.ln_stats_ld = 128; p.ln_parts = 16; p.ln_s = lns; p.ln_c = lnc; p.ln_eps = 1e-5f;
  }
  if (epi == 4) {
    CK(hipMalloc(&stats_out, (size_t)M * 128 * 4));
    p.stats_out = stats_out; p.stats_ld = 128;
  }
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  static const char* names[7] = {"entry -> set-up done", "ring issue", "first pair lands (+barrier)", "K loop", "exchange",
                                 "epilogue math + store issue", "stores acknowledged"};
  for (int cold = 0; cold < 2; ++cold) {
    for (int r = 0; r < reps; ++r) {
      CK(hipMemsetAsync(tr, 0, max_waves * 64, s));
      if (cold) CK(hipMemsetAsync(flush, 0, flush_bytes, s));
      CK(hipEventRecord(e0, s));
      const int rc = da_gemm2::dispatch<false>(p, tile, staging, s);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      if (rc != DA_OK) { std::fprintf(stderr, "dispatch rc=%d\n", rc); return 1; }
      if (r + 1 < reps) continue;                      // report the last repetition
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> h(max_waves * 8);
      CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
      std::vector<unsigned long long> stage[7], t0s, t7s, total;
      for (size_t w = 0; w < max_waves; ++w) {
        const unsigned long long* t = &h[w * 8];
        if (!t[0] || !t[7]) continue;
        for (int i = 0; i < 7; ++i) stage[i].push_back(t[i + 1] - t[i]);
        t0s.push_back(t[0]); t7s.push_back(t[7]); total.push_back(t[7] - t[0]);
      }
      if (t0s.empty()) { std::fprintf(stderr, "no trace\n"); return 1; }
      auto pct = [](std::vector<unsigned long long> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
      const unsigned long long first = *std::min_element(t0s.begin(), t0s.end()), last_in = *std::max_element(t0s.begin(), t0s.end());
      const unsigned long long end = *std::max_element(t7s.begin(), t7s.end()), first_out = *std::min_element(t7s.begin(), t7s.end());
      std::printf("\n## M %d N %d K %d tile %d staging %d epilogue %s, %s operands: %zu waves, event time %.2f us\n", M, N, K, tile, staging,
                  (epi == 0 ? "plain" : epi == 1 ? "bias+residual" : epi == 2 ? "GEGLU" : epi == 3 ? "GEGLU + LN fold" : epi == 4 ? "bias+residual+stats" : "LN fold"), cold ? "cold (flushed)" : "warm", t0s.size(), ms * 1e3);
      std::printf("| stage | median cyc | p10 | p90 |\n|---|---:|---:|---:|\n");
      for (int i = 0; i < 7; ++i)
        std::printf("| %s | %llu | %llu | %llu |\n", names[i], pct(stage[i], 0.5), pct(stage[i], 0.1), pct(stage[i], 0.9));
      std::printf("| wave lifetime | %llu | %llu | %llu |\n", pct(total, 0.5), pct(total, 0.1), pct(total, 0.9));
      std::printf("first wave entry -> last wave entry %llu cyc; first entry -> first wave done %llu; first entry -> last wave done %llu "
                  "(s_memtime ticks; event time / that = %.3f ns per tick)\n", last_in - first, first_out - first, end - first,
                  ms * 1e6 / (double)(end - first));
    }
  }
  return 0;
}
