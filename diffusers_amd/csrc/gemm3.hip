// Third GEMM structure ("K3", round 5): the 256 x 256 tile as an EIGHT-PHASE loop with the two wave rows half a phase apart
// (the schedule of /opt/skills/guides/cdna_hip_programming.md "The 256^2 8-phase template"; VERDICT r4 "next round" item 4).
// nn.Linear only: C[M][N] = epilogue(alpha * A[M][K] . W[N][K]^T), the large-M / large-N products of the DiT blocks
// (transformer_flux.py:383-412 proj_mlp / proj_out, :95-130 to_q/k/v; transformer_wan.py:488-502 ffn) and SDXL's GEGLU projection
// (activations.py:113-124).
//
// Why another structure.  The K1 tiles of gemm2_kernel.cuh run all eight waves through the same K slice at the same time: every
// wave issues its LDS-DMA and its ds_reads right behind the rendezvous and then every wave multiplies -- the matrix pipe of a SIMD
// idles while its two waves fetch, and both want it at once afterwards (1000-1050 TFLOP/s steady state, profiles/r04*).  Here the
// 2 x 4 waves of a workgroup are two GROUPS of four (one wave per SIMD each) that run the same phase sequence ONE BARRIER APART:
// while group 0 multiplies a 64 x 32 quadrant of its 128 x 64 wave tile (16 MFMAs), group 1 -- the other wave of
// every SIMD -- reads its next fragments from LDS and issues its share of the LDS-DMA; at the next barrier they swap.  A SIMD's
// matrix pipe then always has exactly one wave feeding it and the other wave's memory work hides under it.
//
// Geometry.  Tile 256 x 256, K slices of 64 (128 B per row), 8 waves = 2 (wr) x 4 (wc).  LDS: two K-slice buffers x four 16 KiB
// HALF-TILES (A0, A1, B0, B1: 128 rows x 128 B each) = 128 KiB.  A half h = tile rows [128 h, 128 h + 128); wave row wr owns rows
// 64 wr .. 64 wr + 63 of each.  B half h' = the columns 64 q + 32 h' + [0, 32) (q = 0..3) of the tile, wave column wc owns q = wc,
// i.e. the 64 CONTIGUOUS output columns [64 wc, 64 wc + 64) -- with GEGLU's packed weight rows ([32 value | 32 gate] per 64) half 0
// is the value and half 1 its gate.  Quadrant (h, h') of a wave = 4 x 2 MFMA tiles of 16 x 16, x 2 k-steps of 32 = 16 MFMAs
// (v_mfma_f32_16x16x32_bf16, weights as the A operand: a lane ends up with 4 consecutive output channels of one row).
// LDS image of a half-tile: 1 KiB pieces of 8 rows x 128 B written lane-linear by LDS-DMA, the 16-byte slots XOR-swizzled with
// (row >> 1) & 7 on the SOURCE side (as gemm2_kernel.cuh: conflict-free for the ds_read_b128 fragment reads).
//
// Phases of K slice kt (buffer kt & 1); every wave stages 2 pieces (1/8 of one half-tile) per phase; B half 0 of the slice is already
// in registers (BX) when its first phase starts, the roles of the two B register sets alternate from slice to slice:
//   P1 quadrant (0,0): ds_read A0 (8) -> af          stage B0 of slice kt + 2 (this buffer)
//   P2 quadrant (0,1): ds_read B1 (4) -> BY          stage A0 of slice kt + 2
//   P3 quadrant (1,1): ds_read A1 (8) -> af          stage B1 of slice kt + 2
//   P4 quadrant (1,0): ds_read B0 of slice kt + 1 (4, other buffer) -> BY      stage A1 of slice kt + 2
// each phase = [reads, stage, s_waitcnt vmcnt(12), s_waitcnt lgkmcnt(0)] s_barrier [16 MFMA] s_barrier (priority forms: launch()).
// Both waits sit IN FRONT of the phase's first barrier, i.e. inside the segment that runs beside the other group's MFMAs: the
// multiply starts the moment the barrier opens.
// Hazards (group 1 runs one barrier late, so "a phase later" for one group is half a phase later for the other; barriers numbered
// #0, #1, ...: group 0's phase g reads / stages in (#2g-1, #2g) and multiplies in (#2g, #2g+1), group 1's in (#2g, #2g+1) and
// (#2g+1, #2g+2)):
//   WAR  the ds_reads of phase g are RETIRED (lgkmcnt(0)) before their wave arrives at the phase's first barrier, so every read of
//        phase g by either group is complete once barrier #2g+1 has opened, and the staging segment of phase g + 1 starts behind it
//        for both groups: a half-tile may be restaged ONE phase after its last read.  A0 is read in P1 and restaged in P2, B1:
//        P2 -> P3, A1: P3 -> P4, B0 (read in P4 of the previous slice) -> P1: every half-tile at the earliest legal phase.
//   RAW  a wave's counted vmcnt in front of phase w's first barrier retires its own pieces; everybody's are visible once #2w+1 has
//        opened, i.e. to reads from phase w + 1 on.  vmcnt(12) after the phase's own staging leaves the SIX newest half-tiles in
//        flight: the half-tile staged in phase s is retired in phase s + 6 and read in phase s + 7 (B0 of slice kt + 2: staged P1 of
//        kt, read P4 of kt + 1; likewise the other three) -- seven phases, ~3.5 k cycles, between a piece's issue and its use.
// Slices past the end of K are staged as zeros (bit 31 of the per-lane offset: beyond num_records, the range check writes zeros
// without touching memory): every phase issues the same loads, the vmcnt immediates are constants and an odd slice count multiplies
// one slice of zeros.
//
// Variants measured on the way (profiles/r05g_*, same-box ratios against k1:256x256 at 8192^3 / Wan's 32760 x 5120 x 13824):
// the guide's placement (reads 12 / 4 / 8 / 0, lgkmcnt(0) BEHIND the barrier, one vmcnt(6) per slice) 1.12 / 1.09; the reads of a later
// phase issued behind the MFMAs of the current one (nothing but LDS-DMA in the other segment) 1.04 / 0.99 -- the read issue then
// sits between the last MFMA and the barrier that lets the other group start, exposed; this form 1.11 / 1.10 and the best absolute
// rates (1443 TFLOP/s at 8192^3, 1373 at 4096^3, 1434 on the Wan shape; random operands) -- 1484-1509 / 1339-1387 with the 16-byte
// epilogue and the static priority form (launch(), profiles/r05k_*).
//
// Numerics: fp32 accumulation, K slices in order, k-step 0 then 1 -- the order of every KG = 1 tile of gemm2_kernel.cuh, and the
// epilogue is that header's epilogue4: BIT-IDENTICAL to DA_TILE_K1_256x256 (tests/test_gemm_k3_gpu.py asserts equality).
#include "gemm2_shared.cuh"

namespace da_gemm3 {

using da_gemm2::epilogue4;

constexpr int BM = 256, BN = 256;
constexpr int HALF = 128 * 128;      // bytes of one half-tile (128 rows x 64 bf16)
constexpr int KBUF = 4 * HALF;       // one K slice: A0 | A1 | B0 | B1
constexpr int LDS_BYTES = 2 * KBUF;  // 128 KiB ring (+ 1 KiB behind it: the tile's bias)

#define K3_LDS(ptr) ((__attribute__((address_space(3))) void*)(ptr))

template <bool GEGLU, int PRIO = 1>
__global__ __launch_bounds__(512) void gemm3_bf16_kernel(const da_gemm_params p, const int xcd_gx) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r16 = lane & 15, kq = lane >> 4;

  // ---- XCD-aware tile mapping (as gemm2_kernel.cuh: block b runs on XCD b % 8, each XCD owns a rectangle of tiles) ----
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int gyn = 8 / xcd_gx;
  const int tm_per = (tiles_m + gyn - 1) / gyn, tn_per = (tiles_n + xcd_gx - 1) / xcd_gx;
  const int bid = (int)blockIdx.x;
  const int xcd = bid & 7, kblk = bid >> 3;
  const int gy = xcd / xcd_gx, gx = xcd - gy * xcd_gx;
  const int lm = kblk / tn_per, ln = kblk - lm * tn_per;
  const int tm = gy * tm_per + lm, tn = gx * tn_per + ln;
  if (tm >= tiles_m || tn >= tiles_n) return;             // ragged rectangle: the whole block leaves before any barrier
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = p.K >> 6;

  const uint16_t* __restrict__ A = (const uint16_t*)p.A;
  const uint16_t* __restrict__ Wt = (const uint16_t*)p.W;
  __amdgpu_buffer_rsrc_t rs_a = da_gemm2::uniform_rsrc(A + (size_t)m0 * p.lda, 0x7fffffff);
  __amdgpu_buffer_rsrc_t rs_w = da_gemm2::uniform_rsrc(Wt + (size_t)n0 * p.ldw, 0x7fffffff);

  // ---- staging: wave w sends pieces w and w + 8 of every half-tile; lane -> row 8 * piece + (lane >> 3), 16-byte slot lane & 7 ----
  // rows past M / N are clamped to the last valid row (their products land in rows / columns the epilogue never stores)
  int vo_a[2][2], vo_b[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rho = 8 * (wave + 8 * i) + (lane >> 3);
      const int sc = (lane & 7) ^ ((rho >> 1) & 7);
      const int ra = min(h * 128 + rho, p.M - 1 - m0);
      const int rb = min(64 * (rho >> 5) + 32 * h + (rho & 31), p.N - 1 - n0);
      vo_a[h][i] = (ra * p.lda + sc * 8) * 2;
      vo_b[h][i] = (rb * p.ldw + sc * 8) * 2;
    }
  // WHICH: 0 = A0, 1 = A1, 2 = B0, 3 = B1 of slice `ks` into buffer BUF
  auto stage = [&](auto buf_c, auto which_c, int ks) {
    constexpr int BUF = decltype(buf_c)::value, WHICH = decltype(which_c)::value;
    const int z = (ks < nk) ? 0 : (int)0x80000000;
    const int so = min(ks, nk - 1) * 128;
    unsigned char* dst = smem + BUF * KBUF + WHICH * HALF + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (WHICH < 2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, K3_LDS(dst + i * 8192), 16, vo_a[WHICH][i] | z, so, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, K3_LDS(dst + i * 8192), 16, vo_b[WHICH - 2][i] | z, so, 0, 0);
    }
  };
  using std::integral_constant;
  constexpr integral_constant<int, 0> I0{};
  constexpr integral_constant<int, 1> I1{};
  constexpr integral_constant<int, 2> I2{};
  constexpr integral_constant<int, 3> I3{};

  // bias of the tile's 256 columns: ONE LDS-DMA piece (wave 0, lanes 0 .. 31 x 16 bytes; everything past N or without a bias is out of
  // the descriptor's range and arrives as zeros) into the KiB behind the ring -- issued first, so it is older than every staged
  // half-tile and retired by the first counted wait; the epilogue reads it back.  (Held in registers through the loop it cost the
  // eight VGPRs that made the wide epilogue spill.)
  if (wave == 0) {
    const int ncols = min(BN, p.N - n0);
    __amdgpu_buffer_rsrc_t rs_bias = da_gemm2::uniform_rsrc(p.bias ? (const void*)((const uint16_t*)p.bias + n0) : (const void*)Wt,
                                                            p.bias ? (size_t)ncols * 2 : 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_bias, K3_LDS(smem + LDS_BYTES), 16, lane * 16, 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- prologue: slices 0 and 1 whole, in the loop's order (B0, A0, B1, A1) ----
  stage(I0, I2, 0);
  stage(I0, I0, 0);
  stage(I0, I3, 0);
  stage(I0, I1, 0);
  stage(I1, I2, 1);
  stage(I1, I0, 1);
  stage(I1, I3, 1);
  stage(I1, I1, 1);

  f32x4_t acc[2][2][4][2];   // [A half][B half][row tile][column tile]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int hp = 0; hp < 2; ++hp)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[h][hp][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: row r16 of a 16-row tile, 16-byte chunk (kq + 4 ks) ^ ((r16 >> 1) & 7); k-step 1 = chunk ^ 4
  const int fsw = (r16 >> 1) & 7;
  const int foff0 = r16 * 128 + ((kq ^ fsw) << 4), foff1 = r16 * 128 + (((kq ^ fsw) ^ 4) << 4);
  const unsigned char* fa[2] = {smem + wr * 8192 + foff0, smem + wr * 8192 + foff1};
  const unsigned char* fb[2] = {smem + 2 * HALF + wc * 4096 + foff0, smem + 2 * HALF + wc * 4096 + foff1};
  bf16x8_t af[4][2], bq0[2][2], bq1[2][2];   // [tile][k-step]: the current A half and two B halves (roles alternate per slice)

#define K3_READ_A(BUF, H)                                                                                         \
  do {                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                              \
      af[i][ks] = *(const bf16x8_t*)(fa[ks] + (BUF) * KBUF + (H) * HALF + i * 2048);                              \
  } while (0)
#define K3_READ_B(DST, BUF, HP)                                                                                   \
  do {                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                              \
      DST[j][ks] = *(const bf16x8_t*)(fb[ks] + (BUF) * KBUF + (HP) * HALF + j * 2048);                            \
  } while (0)
#define K3_MFMA(H, HP, BQ)                                                                                        \
  do {                                                                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
      acc[H][HP][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BQ[j][ks], af[i][ks], acc[H][HP][i][j], 0, 0, 0); \
  } while (0)
#define K3_FENCE() __builtin_amdgcn_sched_barrier(0)
// one phase: [ds_reads of THIS phase's new fragments, LDS-DMA of one half-tile, counted waits] barrier [16 MFMAs] barrier
#define K3_PHASE(READ_STMT, STAGE_STMT, H, HP, BQ)                                                                \
  do {                                                                                                            \
    K3_FENCE();                                                                                                   \
    READ_STMT;                                                                                                    \
    K3_FENCE();                                                                                                   \
    STAGE_STMT;                                                                                                   \
    K3_FENCE();                                                                                                   \
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   /* all but the six newest half-tiles have landed */       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  /* this phase's fragments are in registers */             \
    K3_FENCE();                                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                 \
    K3_FENCE();                                                                                                   \
    if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);                                                       \
    K3_MFMA(H, HP, BQ);                                                                                           \
    if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);                                                       \
    K3_FENCE();                                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                 \
    K3_FENCE();                                                                                                   \
  } while (0)
// the four phases of slice KT held in buffer BUF (the other buffer: OTH); BX holds B half 0 of the slice on entry, BY is free
#define K3_SLICE(BUF, OTH, KT, BX, BY)                                                                            \
  do {                                                                                                            \
    K3_PHASE(K3_READ_A(BUF, 0), stage(integral_constant<int, BUF>{}, I2, (KT) + 2), 0, 0, BX);                    \
    K3_PHASE(K3_READ_B(BY, BUF, 1), stage(integral_constant<int, BUF>{}, I0, (KT) + 2), 0, 1, BY);                \
    K3_PHASE(K3_READ_A(BUF, 1), stage(integral_constant<int, BUF>{}, I3, (KT) + 2), 1, 1, BY);                    \
    K3_PHASE(K3_READ_B(BY, OTH, 0), stage(integral_constant<int, BUF>{}, I1, (KT) + 2), 1, 0, BX);                \
  } while (0)

  // B0 / A0 of slice 0 landed (every wave's own pieces; the barrier makes them everybody's), six half-tiles in flight; B half 0 of
  // slice 0 is read here and is IN REGISTERS before P1 restages its half-tile
  K3_FENCE();
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  K3_FENCE();
  K3_READ_B(bq0, 0, 0);
  K3_FENCE();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  K3_FENCE();
  if (wr == 1) __builtin_amdgcn_s_barrier();              // group 1 runs one barrier behind group 0 from here on
  if constexpr (PRIO == 2) {                              // static priority for the second-dispatched half, no flips (see launch())
    if (wr == 1) __builtin_amdgcn_s_setprio(1);
  }
  K3_FENCE();
  for (int kt = 0; kt < nk; kt += 2) {
    K3_SLICE(0, 1, kt, bq0, bq1);
    K3_SLICE(1, 0, kt + 1, bq1, bq0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the zero-filled stagings of slices past K)
  K3_FENCE();
  if (wr == 0) __builtin_amdgcn_s_barrier();              // group 0 meets group 1's last barrier
  K3_FENCE();
  __builtin_amdgcn_s_barrier();                           // every wave's LDS-DMA and ds_reads are done: LDS is free for the epilogue
  K3_FENCE();

  // ---- epilogue: lane holds, for output row r16 of a 16-row tile, channels 4 kq .. 4 kq + 3 of a 16-column tile ----
  uint2 bias_v[2][2];                                       // the wave's four 16-column tiles, this lane's 4 channels of each
#pragma unroll
  for (int hp = 0; hp < 2; ++hp)
#pragma unroll
    for (int j = 0; j < 2; ++j) bias_v[hp][j] = *(const uint2*)(smem + LDS_BYTES + (64 * wc + 32 * hp + 16 * j + 4 * kq) * 2);
  K3_FENCE();                                               // (read before the wide epilogue's scratch traffic is issued)
  auto row_of = [&](int h, int i) { return m0 + h * 128 + wr * 64 + i * 16 + r16; };
  auto col_of = [&](int hp, int j) { return n0 + 64 * wc + 32 * hp + 16 * j + 4 * kq; };
  if constexpr (GEGLU) {
    // packed weight rows: per 64 = [32 value | 32 gate]: B half 0 is the value, half 1 its gate (activations.py:113-124)
    auto body = [&](auto tanh_c) __attribute__((always_inline)) {
      constexpr bool TANH = decltype(tanh_c)::value;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = row_of(h, i);
          if (m >= p.M) continue;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (col_of(0, j) >= p.N) continue;
            const int no = ((n0 + 64 * wc) >> 1) + 16 * j + 4 * kq;
            const uint2 bh = bias_v[0][j], bg = bias_v[1][j];      // zeros without a bias
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float hv = acc[h][0][i][j][e] * p.alpha, gv = acc[h][1][i][j][e] * p.alpha;
              hv += (e == 0) ? bf_lo(bh.x) : (e == 1) ? bf_hi(bh.x) : (e == 2) ? bf_lo(bh.y) : bf_hi(bh.y);
              gv += (e == 0) ? bf_lo(bg.x) : (e == 1) ? bf_hi(bg.x) : (e == 2) ? bf_lo(bg.y) : bf_hi(bg.y);
              hv = bf2f(f2bf(hv));   // the reference rounds the projection to bf16 before chunk / gelu / mul
              gv = bf2f(f2bf(gv));
              o[e] = hv * bf2f(f2bf(TANH ? gelu_tanh_f(gv) : gelu_erf_f(gv)));
            }
            uint2 pk;
            pk.x = pack_bf2(o[0], o[1]);
            pk.y = pack_bf2(o[2], o[3]);
            *(uint2*)((uint16_t*)p.C + (size_t)m * p.ldc + no) = pk;
          }
        }
    };
    if (p.act == DA_ACT_GEGLU) body(std::false_type{});
    else body(std::true_type{});
  } else {
    const uint16_t* __restrict__ bias_rows = (const uint16_t*)p.bias_rows;
    const bool has_rowvec = p.rowvec != nullptr, has_res = p.residual != nullptr;
    auto body = [&](auto act_c, auto gate_c) __attribute__((always_inline)) {
      constexpr int ACT = decltype(act_c)::value, GATE = decltype(gate_c)::value;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = row_of(h, i);
          if (m >= p.M) continue;
          const int bidx = (GATE != 0 || has_rowvec) ? (m / p.rows_per_batch) : 0;
          const float brow = bias_rows ? bf2f(bias_rows[m]) : 0.f;
#pragma unroll
          for (int hp = 0; hp < 2; ++hp)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int n = col_of(hp, j);
              if (n >= p.N) continue;
              float o[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = acc[h][hp][i][j][e] * p.alpha;
              uint2 rvv = make_uint2(0, 0), rsv = make_uint2(0, 0);
              if (has_rowvec) rvv = *(const uint2*)((const uint16_t*)p.rowvec + (size_t)bidx * p.ld_rowvec + n);
              if (has_res) rsv = *(const uint2*)((const uint16_t*)p.residual + (size_t)m * p.ldr + n);
              epilogue4<ACT, GATE, true>(p, o, n, bidx, brow, bias_v[hp][j], rvv, rsv);
              if (p.out_f32) {
                *(float4*)((float*)p.C + (size_t)m * p.ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
              } else {
                uint2 pk;
                pk.x = pack_bf2(o[0], o[1]);
                pk.y = pack_bf2(o[2], o[3]);
                *(uint2*)((uint16_t*)p.C + (size_t)m * p.ldc + n) = pk;
              }
            }
        }
    };
    // Row-contiguous output: a wave's 16 x 64 band of finished values goes through LDS once (wave-private scratch: the ring is free, and
    // a wave's LDS operations execute in order, so no barrier) so that every lane stores -- and, for the residual, loads -- 16
    // contiguous bytes of ONE row: 128 contiguous bytes per row and instruction, a whole line, instead of four separate 32-byte
    // segments by four instructions.  The 8-byte path's store tail (32 global_store_dwordx2 per lane, 16 partial lines each) is
    // issue-bound (MI355X_MICROARCH.md, "epilogue store tail") and cost the short-K Wan shapes more than the loop gained.  Values
    // cross LDS in fp32 and meet the residual / output scale behind it (the same fp32 operations in the same order as the 8-byte
    // path: bit-identical), or already packed to bf16 when there is neither.  Needs 16-byte aligned rows.
    const bool wide = !p.out_f32 && !(p.ldc & 7) && !((size_t)p.C & 15) && !(p.N & 7) &&
                      (!p.residual || (!(p.ldr & 7) && !((size_t)p.residual & 15)));
    constexpr int ROWB = 4 * 64 + 32;                       // bytes per staged fp32 row (ROWB / 2: the packed rows)
    unsigned char* stg = smem + wave * 16384;               // two alternating bands of 16 rows
    auto wide_body = [&](auto act_c, auto gate_c, auto packed_c) __attribute__((always_inline)) {
      constexpr int ACT = decltype(act_c)::value, GATE = decltype(gate_c)::value;
      constexpr bool PACKED = decltype(packed_c)::value;
      const uint16_t* __restrict__ resid = (const uint16_t*)p.residual;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned char* sb = stg + (i & 1) * (16 * ROWB);
          const int mc = min(row_of(h, i), p.M - 1);
          const int bidx = (GATE != 0 || has_rowvec) ? (mc / p.rows_per_batch) : 0;
          const float brow = bias_rows ? bf2f(bias_rows[mc]) : 0.f;
#pragma unroll
          for (int hp = 0; hp < 2; ++hp)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int c = 32 * hp + 16 * j + 4 * kq;      // column inside the wave's 64-column band
              const int n = min(n0 + 64 * wc + c, p.N - 4);
              float o[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = acc[h][hp][i][j][e] * p.alpha;
              uint2 rvv = make_uint2(0, 0);
              if (has_rowvec) rvv = *(const uint2*)((const uint16_t*)p.rowvec + (size_t)bidx * p.ld_rowvec + n);
              epilogue4<ACT, GATE, false>(p, o, n, bidx, brow, bias_v[hp][j], rvv, make_uint2(0, 0));
              if constexpr (PACKED) {
                uint2 pk;
                pk.x = pack_bf2(o[0], o[1]);
                pk.y = pack_bf2(o[2], o[3]);
                *(uint2*)(sb + r16 * (ROWB / 2) + c * 2) = pk;
              } else {
                *(float4*)(sb + r16 * ROWB + c * 4) = make_float4(o[0], o[1], o[2], o[3]);
              }
            }
          // ... and back: piece q = lane + 64 t of the band -> row q / 8, 8-channel slot q % 8
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            const int q = lane + 64 * t2;
            const int row = q >> 3, c8 = q & 7;
            const int mo = m0 + h * 128 + wr * 64 + i * 16 + row, no = n0 + 64 * wc + c8 * 8;
            const bool inside = mo < p.M && no < p.N;
            uint4 pk;
            if constexpr (PACKED) {
              pk = *(const uint4*)(sb + row * (ROWB / 2) + c8 * 16);
            } else {
              const float4 lo = *(const float4*)(sb + row * ROWB + c8 * 32), hi = *(const float4*)(sb + row * ROWB + c8 * 32 + 16);
              float o[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
              uint4 rw = make_uint4(0, 0, 0, 0);
              if (has_res && inside) rw = *(const uint4*)(resid + (size_t)mo * p.ldr + no);
              o[0] += bf_lo(rw.x); o[1] += bf_hi(rw.x); o[2] += bf_lo(rw.y); o[3] += bf_hi(rw.y);
              o[4] += bf_lo(rw.z); o[5] += bf_hi(rw.z); o[6] += bf_lo(rw.w); o[7] += bf_hi(rw.w);
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] *= p.out_scale;
              pk.x = pack_bf2(o[0], o[1]); pk.y = pack_bf2(o[2], o[3]); pk.z = pack_bf2(o[4], o[5]); pk.w = pack_bf2(o[6], o[7]);
            }
            if (inside) *(uint4*)((uint16_t*)p.C + (size_t)mo * p.ldc + no) = pk;
          }
        }
    };
    constexpr integral_constant<int, -1> RT{};
    constexpr integral_constant<int, 0> G0{};
    if (wide) {
      const bool packed = !has_res && p.out_scale == 1.0f;
      if (p.gate && p.gate_f32) wide_body(RT, integral_constant<int, 2>{}, std::false_type{});
      else if (p.gate) wide_body(RT, integral_constant<int, 1>{}, std::false_type{});
      else if (packed) {
        if (p.act == DA_ACT_NONE) wide_body(integral_constant<int, DA_ACT_NONE>{}, G0, std::true_type{});
        else if (p.act == DA_ACT_GELU_TANH) wide_body(integral_constant<int, DA_ACT_GELU_TANH>{}, G0, std::true_type{});
        else wide_body(RT, G0, std::true_type{});
      } else {
        if (p.act == DA_ACT_NONE) wide_body(integral_constant<int, DA_ACT_NONE>{}, G0, std::false_type{});
        else wide_body(RT, G0, std::false_type{});
      }
    } else {   // fp32 output, rows that are not 16-byte aligned: the 8-byte path, one run-time-switched copy per gate form
      if (p.gate && p.gate_f32) body(RT, integral_constant<int, 2>{});
      else if (p.gate) body(RT, integral_constant<int, 1>{});
      else body(RT, G0);
    }
  }
#undef K3_READ_A
#undef K3_READ_B
#undef K3_MFMA
#undef K3_FENCE
#undef K3_PHASE
#undef K3_SLICE
#endif  // __HIP_DEVICE_COMPILE__
}

template <bool GEGLU, int PRIO>
int launch_prio(const da_gemm_params& p, hipStream_t s) {
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int gx = da_gemm2::choose_xcd_gx2(tiles_m, tiles_n, BM, BN), gy = 8 / gx;
  const int grid = 8 * ((tiles_m + gy - 1) / gy) * ((tiles_n + gx - 1) / gx);
  auto kern = gemm3_bf16_kernel<GEGLU, PRIO>;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 1024) != hipSuccess)
      return DA_ERR_LAUNCH;
    attr_set = true;
  }
  DA_LAUNCH(kern, dim3(grid), dim3(512), LDS_BYTES + 1024, s, p, gx);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// The GEGLU projection's own eight-phase tile: 256 x 320 (DA_TILE_K3_256x320).  SDXL's ff.net.0.proj (activations.py:113-124) is
// M 2048 x N 10240 (packed [32 value | 32 gate] rows) x K 1280 at the 32 x 32 level and 8192 x 5120 x 640 at the 64 x 64 level: 320
// and 640 tiles of 256 x 256 (1.25 / 2.5 rounds of the 256 CUs), but exactly 256 and 512 tiles of 256 x 320 -- ONE and TWO rounds.  It
// is the largest single launch population of an SDXL step (70 launches, 4.1 ms of 20.9).
//
// Same structure as the kernel above (two wave groups one barrier apart, phases of [reads, one half-tile of LDS-DMA, waits] barrier
// [MFMAs] barrier, every half-tile restaged one phase after its last read) with the roles of the operands exchanged, because the wave
// tile is 64 x 160:
//   8 waves = 4 rows (wr) x 2 columns (wc = the group); A half h = tile rows 64 wr' + 32 h + [0, 32): a wave owns 2 row tiles of each
//   half; B half 0 = the ten VALUE tiles of the 320 packed columns, half 1 their GATE tiles (LDS row 16 T + r of half h' = packed row
//   64 (T / 2) + 32 h' + 16 (T % 2) + r): a wave owns value tiles 5 wc .. 5 wc + 4 and their gates, i.e. 80 contiguous OUTPUT columns.
//   Quadrant (h, h') = 2 x 5 tiles x 2 k-steps = 20 MFMAs; quadrant order (0,0) (1,0) (1,1) (0,1) with ONE B fragment set (40
//   registers: B0 for P1 / P2, B1 for P3 / P4) and ONE A set (16: A0, A1, A1, A0 again -- A half 0 is read twice per slice) next to 160
//   accumulator registers: 240 VGPRs, no scratch.  (Two A sets -- A0 kept from P1 to P4, the next slice's A0 read in P4, six half-tiles
//   in flight as in the 256 x 256 kernel -- compiled to 256 registers plus spills INSIDE the loop, whose scratch traffic also counts in
//   vmcnt: discarded at the compiler output.)
//   P1: read A0 (4) + B0 (10)   stage A0 of slice kt + 1 (other buffer)     P2: read A1 (4)    stage B0 of slice kt + 2
//   P3: read B1 (10)            stage A1 of slice kt + 2                    P4: read A0 (4)    stage B1 of slice kt + 2, vmcnt(8)
//   ONE counted wait per slice: P4's vmcnt(8) leaves the three newest half-tiles (B0, A1, B1 of slice kt + 2: 3 + 2 + 3 loads) in
//   flight and retires everything older, i.e. all of slice kt + 1 -- read from P1 of kt + 1 on, one phase after the wait (RAW rule of
//   the kernel above).  Latency budgets in phases of ~680 cycles: B0 6, A1 5, B1 4 (the weight, met cold) and 3 for A0 (the
//   activation, written by the kernel in front: warm).  WAR: A0 is last read in P4 and restaged in P1 of the next slice, B0: P1 -> P2,
//   A1: P2 -> P3, B1: P3 -> P4 -- each one phase after its last read, whose lgkmcnt(0) sits in front of that phase's barrier.
//   LDS: 2 x (16 + 16 + 20 + 20) KiB + 1 KiB dump (the four waves without a third B piece issue theirs out of range into it; the weight
//   prefetch lands there too) + 1 KiB bias = 146 KiB.  The fragment bases toggle between the two buffers by an opaque add (as
//   compile-time offsets the second buffer's 72 KiB origin does not fit ds_read's 16-bit offset field and the compiler keeps a
//   second set of bases).
// Restrictions (tile_ok): GEGLU epilogue only, M % 256 == 0, N % 320 == 0 (no clamped rows: pieces and halves ride in the scalar
// offset, ONE per-lane offset per operand serves all of them), 16-byte aligned output rows.
// Numerics: K slices in order, k-step 0 then 1, the GEGLU arithmetic of the other tiles: bit-identical to k1:128x320.
// Measured (profiles/r05n_*): chains 58.2 -> 51.6 us at 2048 x 10240 x 1280 and 72 -> 59.5 us at 8192 x 5120 x 640, in situ 58.7 -> 48.5 us,
// same-box SDXL image + 3.9 %.
constexpr int G_BM = 256, G_BN = 320;
constexpr int G_AH = 128 * 128, G_BH = 160 * 128;          // bytes of an A / B half-tile
constexpr int G_KBUF = 2 * G_AH + 2 * G_BH;                // 72 KiB per slice
constexpr int G_DUMP = 2 * G_KBUF, G_BIAS = G_DUMP + 1024, G_LDS = G_BIAS + 1024;
// LayerNorm fold, consumer side (round 6: norm3 -> ff.net.0.proj on this tile; attention.py:1056, da_gemm_params.ln_*): the tile's 320
// s[n] and c[n] (fp32, packed column order like the bias) and the (mean, rstd) of its 256 rows, behind the bias KiB
constexpr int G_LNS = G_LDS, G_LNC = G_LNS + 2048, G_LNROW = G_LNC + 2048, G_LDS_LNF = G_LNROW + G_BM * 8;   // (2 KiB each: an LDS-DMA piece writes all 64 lanes)

template <int PRIO, bool LNF = false>
__global__ __launch_bounds__(512) void gemm3_geglu_kernel(const da_gemm_params p, const int xcd_gx) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wc = wave >> 2, wr = wave & 3;                 // wc = the group (one wave per SIMD each)
  const int r16 = lane & 15, kq = lane >> 4;

  const int tiles_n = p.N / G_BN, tiles_m = p.M / G_BM;
  const int gyn = 8 / xcd_gx;
  const int tm_per = (tiles_m + gyn - 1) / gyn, tn_per = (tiles_n + xcd_gx - 1) / xcd_gx;
  const int bid = (int)blockIdx.x;
  const int xcd = bid & 7, kblk = bid >> 3;
  const int gy = xcd / xcd_gx, gx = xcd - gy * xcd_gx;
  const int lm = kblk / tn_per, ln = kblk - lm * tn_per;
  const int tm = gy * tm_per + lm, tn = gx * tn_per + ln;
  if (tm >= tiles_m || tn >= tiles_n) return;
  const int m0 = tm * G_BM, n0 = tn * G_BN;
  const int nk = p.K >> 6;

  const uint16_t* __restrict__ A = (const uint16_t*)p.A;
  const uint16_t* __restrict__ Wt = (const uint16_t*)p.W;
  __amdgpu_buffer_rsrc_t rs_a = da_gemm2::uniform_rsrc(A + (size_t)m0 * p.lda, 0x7fffffff);
  __amdgpu_buffer_rsrc_t rs_w = da_gemm2::uniform_rsrc(Wt + (size_t)n0 * p.ldw, 0x7fffffff);

  // ---- staging: wave w sends pieces w, w + 8 (A) / w, w + 8, w + 16 (B; the third exists for w < 4) of every half-tile.  Offsets are
  // those of half 0; half 1 adds 32 rows through the scalar offset ----
  // (piece w + 8 i is 64 LDS rows = 128 tile rows further on for both operands: uniform, so ONE per-lane offset per operand serves every
  // piece and half -- the rest rides in the scalar offset)
  int vo_a, vo_b;
  {
    const int rho = 8 * wave + (lane >> 3);               // piece `wave`: LDS rows 0 .. 63
    const int sc = (lane & 7) ^ ((rho >> 1) & 7);
    const int T = rho >> 4;
    vo_a = ((64 * (rho >> 5) + (rho & 31)) * p.lda + sc * 8) * 2;
    vo_b = ((64 * (T >> 1) + 16 * (T & 1) + (rho & 15)) * p.ldw + sc * 8) * 2;
  }
  const int half_a = 32 * p.lda * 2, half_b = 32 * p.ldw * 2, piece_a = 128 * p.lda * 2, piece_b = 128 * p.ldw * 2;
  const int dead3 = wave < 4 ? 0 : (int)0x80000000;       // the third B piece exists for waves 0 .. 3 (LDS rows 128 .. 159)
  // WHICH: 0 = A0, 1 = A1, 2 = B0, 3 = B1 of slice `ks` into buffer BUF
  auto stage = [&](auto buf_c, auto which_c, int ks) {
    constexpr int BUF = decltype(buf_c)::value, WHICH = decltype(which_c)::value;
    const int z = (ks < nk) ? 0 : (int)0x80000000;
    const int so = min(ks, nk - 1) * 128;
    if constexpr (WHICH < 2) {
      unsigned char* dst = smem + BUF * G_KBUF + WHICH * G_AH + wave * 1024;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, K3_LDS(dst + i * 8192), 16, vo_a | z, so + WHICH * half_a + i * piece_a, 0, 0);
    } else {
      unsigned char* dst = smem + BUF * G_KBUF + 2 * G_AH + (WHICH - 2) * G_BH + wave * 1024;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, K3_LDS(dst + i * 8192), 16, vo_b | z, so + (WHICH - 2) * half_b + i * piece_b, 0, 0);
      // third piece (waves 0 .. 3); the others issue theirs out of range (zeros, no memory traffic) into the dump KiB
      unsigned char* dst3 = wave < 4 ? dst + 16384 : smem + G_DUMP;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, K3_LDS(dst3), 16, vo_b | z | dead3, so + (WHICH - 2) * half_b + 2 * piece_b, 0, 0);
    }
  };
  using std::integral_constant;
  constexpr integral_constant<int, 0> I0{};
  constexpr integral_constant<int, 1> I1{};
  constexpr integral_constant<int, 2> I2{};
  constexpr integral_constant<int, 3> I3{};

  // bias of the tile's 320 packed columns: one LDS-DMA piece (wave 0, lanes 0 .. 39 x 16 bytes; zeros without a bias), issued first
  if (wave == 0) {
    __amdgpu_buffer_rsrc_t rs_bias = da_gemm2::uniform_rsrc(p.bias ? (const void*)((const uint16_t*)p.bias + n0) : (const void*)Wt,
                                                            p.bias ? (size_t)G_BN * 2 : 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_bias, K3_LDS(smem + G_BIAS), 16, lane * 16, 0, 0, 0);
  }
  // LayerNorm fold: the statistics partials of the tile's 256 rows -- four lanes per row, six (sum, sum of squares) slots = three
  // 16-byte loads each, rows t / 4 and 128 + t / 4, exactly the slots / order the other families' consumers read (gemm2_kernel.cuh
  // ln_load) -- and the tile's s / c by LDS-DMA (waves 1 and 2: 1280 bytes = two pieces, range-checked).  Issued FIRST: older than every
  // staged piece, so the "slice 0 landed" wait covers them and the loop's counted waits never see them.
  float4 lq[LNF ? 2 : 1][3];
  if constexpr (LNF) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const float* sp = p.ln_stats + (size_t)(m0 + 128 * rr + (t >> 2)) * p.ln_stats_ld;
#pragma unroll
      for (int u = 0; u < 3; ++u) lq[rr][u] = *(const float4*)(sp + ((6 * (t & 3) + 2 * u < p.ln_parts) ? 12 * (t & 3) + 4 * u : 0));
    }
    if (wave == 1 || wave == 2) {
      __amdgpu_buffer_rsrc_t rs_sc = da_gemm2::uniform_rsrc((wave == 1 ? p.ln_s : p.ln_c) + n0, (size_t)G_BN * 4);
      unsigned char* dst = smem + (wave == 1 ? G_LNS : G_LNC);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_sc, K3_LDS(dst), 16, lane * 16, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_sc, K3_LDS(dst + 1024), 16, lane * 16, 1024, 0, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- prologue: slice 0 whole, then B0 / A1 / B1 of slice 1 (its A0 leaves in P1 of slice 0: the loop's order) ----
  stage(I0, I0, 0);
  stage(I0, I2, 0);
  stage(I0, I1, 0);
  stage(I0, I3, 0);
  stage(I1, I2, 1);
  stage(I1, I1, 1);
  stage(I1, I3, 1);

  f32x4_t acc[2][2][2][5];   // [A half][B half: value / gate][row tile][column tile]
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int hp = 0; hp < 2; ++hp)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 5; ++j) acc[h][hp][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  if constexpr (!LNF) zero_acc();    // (LNF: behind the statistics reduce, whose 24 registers die there)

  const int fsw = (r16 >> 1) & 7;
  const int foff0 = r16 * 128 + ((kq ^ fsw) << 4), foff1 = r16 * 128 + (((kq ^ fsw) ^ 4) << 4);
  // Fragment base addresses of the CURRENT buffer (32-bit LDS addresses), toggled between the two buffers by an opaque add: written as
  // compile-time buffer offsets the second buffer's 72 KiB origin does not fit the 16-bit ds_read offset and the compiler keeps a
  // second set of bases -- the registers that made this kernel spill.  pa moves in front of P4 (whose read is the NEXT slice's A0),
  // pb behind it.
  unsigned pa0 = (unsigned)(size_t)(smem + wr * 4096 + foff0), pa1 = (unsigned)(size_t)(smem + wr * 4096 + foff1);
  unsigned pb0 = (unsigned)(size_t)(smem + 2 * G_AH + wc * 10240 + foff0), pb1 = (unsigned)(size_t)(smem + 2 * G_AH + wc * 10240 + foff1);
  int dbuf = G_KBUF;                                       // + 72 KiB, then - 72 KiB, ...
  bf16x8_t af[2][2], bq[5][2];   // [tile][k-step]: the current A half, the current B half
#define G3_LDSP(addr) ((const __attribute__((address_space(3))) bf16x8_t*)(size_t)(addr))
#define G3_TOGGLE(P0, P1)                                                                                         \
  do {                                                                                                            \
    asm volatile("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2" : "+v"(P0), "+v"(P1) : "s"(dbuf));               \
  } while (0)
#define G3_READ_A(H)                                                                                              \
  do {                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                               \
      af[i][0] = *G3_LDSP(pa0 + (H) * G_AH + i * 2048);                                                           \
      af[i][1] = *G3_LDSP(pa1 + (H) * G_AH + i * 2048);                                                           \
    }                                                                                                             \
  } while (0)
#define G3_READ_B(HP)                                                                                             \
  do {                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                                               \
      bq[j][0] = *G3_LDSP(pb0 + (HP) * G_BH + j * 2048);                                                          \
      bq[j][1] = *G3_LDSP(pb1 + (HP) * G_BH + j * 2048);                                                          \
    }                                                                                                             \
  } while (0)
#define G3_MFMA(H, HP)                                                                                            \
  do {                                                                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                 \
    _Pragma("unroll") for (int j = 0; j < 5; ++j)                                                                 \
      acc[H][HP][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j][ks], af[i][ks], acc[H][HP][i][j], 0, 0, 0); \
  } while (0)
#define G3_FENCE() __builtin_amdgcn_sched_barrier(0)
// WAIT_STMT: the slice's ONE counted vmcnt (P4), nothing elsewhere
#define G3_PHASE(READ_STMT, STAGE_STMT, WAIT_STMT, H, HP)                                                         \
  do {                                                                                                            \
    G3_FENCE();                                                                                                   \
    READ_STMT;                                                                                                    \
    G3_FENCE();                                                                                                   \
    STAGE_STMT;                                                                                                   \
    G3_FENCE();                                                                                                   \
    WAIT_STMT;                                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
    G3_FENCE();                                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                 \
    G3_FENCE();                                                                                                   \
    if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);                                                       \
    G3_MFMA(H, HP);                                                                                               \
    if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);                                                       \
    G3_FENCE();                                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                 \
    G3_FENCE();                                                                                                   \
  } while (0)
#define G3_READ_A0B0() do { G3_READ_A(0); G3_FENCE(); G3_READ_B(0); } while (0)
#define G3_VMCNT8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")   /* all but the three newest half-tiles (B0, A1, B1: 3 + 2 + 3 loads) */
// the four phases of slice KT in buffer BUF (the other buffer: OTH)
#define G3_SLICE(BUF, OTH, KT)                                                                                    \
  do {                                                                                                            \
    G3_PHASE(G3_READ_A0B0(), stage(integral_constant<int, OTH>{}, I0, (KT) + 1), (void)0, 0, 0);                  \
    G3_PHASE(G3_READ_A(1), stage(integral_constant<int, BUF>{}, I2, (KT) + 2), (void)0, 1, 0);                    \
    G3_PHASE(G3_READ_B(1), stage(integral_constant<int, BUF>{}, I1, (KT) + 2), (void)0, 1, 1);                    \
    G3_PHASE(G3_READ_A(0), stage(integral_constant<int, BUF>{}, I3, (KT) + 2), G3_VMCNT8(), 0, 1);                \
    G3_TOGGLE(pa0, pa1);                                                                                          \
    G3_TOGGLE(pb0, pb1);                                                                                          \
    dbuf = -dbuf;                                                                                                 \
  } while (0)

  // slice 0 landed (every wave's own pieces; the barrier makes them everybody's), B0 / A1 / B1 of slice 1 in flight
  G3_FENCE();
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  G3_FENCE();
  if constexpr (LNF) {
    // (mean, rstd) of rows t / 4 and 128 + t / 4 -- the partials landed with slice 0 -- in the summation order of the other families'
    // consumers (six slots per lane in slot order, then (a + b) + (c + d) across the row's four lanes); lane 0 of a row parks the pair
    // in LDS for the epilogue, so nothing rides through the loop
    const float inv_c = 1.0f / (float)p.K;
    const int q4 = t & 3;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int q = 6 * q4 + 2 * u;
        s1 += (q < p.ln_parts ? lq[rr][u].x : 0.f) + (q + 1 < p.ln_parts ? lq[rr][u].z : 0.f);
        s2 += (q < p.ln_parts ? lq[rr][u].y : 0.f) + (q + 1 < p.ln_parts ? lq[rr][u].w : 0.f);
      }
      s1 += __shfl_xor(s1, 1, 64);
      s2 += __shfl_xor(s2, 1, 64);
      s1 += __shfl_xor(s1, 2, 64);
      s2 += __shfl_xor(s2, 2, 64);
      const float mean = s1 * inv_c;
      const float rs = rsqrtf(fmaxf(s2 * inv_c - mean * mean, 0.f) + p.ln_eps);
      // (written through inline asm: behind a C++ store to LDS the compiler waits for EVERY outstanding LDS-DMA piece -- vmcnt(0), i.e.
      // slice 1's round trip in front of the loop -- because it cannot see that no piece lands here; the phases' lgkmcnt(0) retire it)
      if (q4 == 0) {
        const unsigned dst = (unsigned)(size_t)(smem + G_LNROW + (128 * rr + (t >> 2)) * 8);
        const float2 mr = make_float2(mean, rs);
        asm volatile("ds_write_b64 %0, %1" : : "v"(dst), "v"(mr));
      }
    }
    G3_FENCE();
    zero_acc();
    G3_FENCE();
  }
  if (wc == 1) __builtin_amdgcn_s_barrier();              // group 1 runs one barrier behind group 0 from here on
  if constexpr (PRIO == 2) {
    if (wc == 1) __builtin_amdgcn_s_setprio(1);
  }
  G3_FENCE();
  for (int kt = 0; kt < nk; kt += 2) {
    G3_SLICE(0, 1, kt);
    G3_SLICE(1, 0, kt + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G3_FENCE();
  if (wc == 0) __builtin_amdgcn_s_barrier();              // group 0 meets group 1's last barrier
  G3_FENCE();
  __builtin_amdgcn_s_barrier();                           // LDS is free for the epilogue
  G3_FENCE();

  // optional: pull a later launch's weight towards the memory-side cache (da_gemm_params.prefetch, as gemm2_kernel.cuh: up to 16 KiB per
  // wave of this workgroup's share into the dump KiB; nothing waits for it)
  if (p.prefetch) {
    const int nchunk = (int)min((long long)0x7fffffff >> 10, p.prefetch_bytes >> 10);
    __amdgpu_buffer_rsrc_t rs_pf = da_gemm2::uniform_rsrc(p.prefetch, (size_t)nchunk << 10);
    const int stride = (int)gridDim.x * 8;
    int c = bid * 8 + wave;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (c < nchunk) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_pf, K3_LDS(smem + G_DUMP), 16, lane * 16, c << 10, 0, 0);
      c += stride;
    }
  }

  // ---- epilogue: value * gelu(gate) of the wave's 64 rows x 80 output columns, whole rows from wave-private LDS (16 bytes per lane) ----
  // (its lane-dependent values are formed from an OPAQUE copy of the thread index: computed from `lane` the compiler hoists them above
  // the loop and they cost the loop registers it does not have)
  int tl = t;
  asm volatile("" : "+v"(tl));
  const int lane_e = tl & 63, r16_e = tl & 15, kq_e = (tl >> 4) & 3;
  if constexpr (LNF) {
    // LayerNorm fold: acc <- rstd (alpha acc - mu s[n]) + c[n], in place, in front of the bias -- the arithmetic (and statement structure:
    // -ffp-contract=on fuses inside one expression only) of gemm2_kernel.cuh ln_apply4.  Column tile outermost: the lane's s / c quads of a
    // value tile and its gate tile are read once and serve its four rows.
    float2 mr[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) mr[h][i] = *(const float2*)(smem + G_LNROW + (64 * wr + 32 * h + 16 * i + r16_e) * 8);
    const float* lns = (const float*)(smem + G_LNS);
    const float* lnc = (const float*)(smem + G_LNC);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int T = 5 * wc + j;
      const int cv0 = 64 * (T >> 1) + 16 * (T & 1) + 4 * kq_e;   // packed column of the value quad; its gate quad: + 32
      const float4 sv = *(const float4*)(lns + cv0), cv = *(const float4*)(lnc + cv0);
      const float4 sg = *(const float4*)(lns + cv0 + 32), cg = *(const float4*)(lnc + cv0 + 32);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float ln_mu = mr[h][i].x, ln_rs = mr[h][i].y;
          float hq[4], gq[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) hq[e] = acc[h][0][i][j][e] * p.alpha, gq[e] = acc[h][1][i][j][e] * p.alpha;
          hq[0] = ln_rs * (hq[0] - ln_mu * sv.x) + cv.x;
          hq[1] = ln_rs * (hq[1] - ln_mu * sv.y) + cv.y;
          hq[2] = ln_rs * (hq[2] - ln_mu * sv.z) + cv.z;
          hq[3] = ln_rs * (hq[3] - ln_mu * sv.w) + cv.w;
          gq[0] = ln_rs * (gq[0] - ln_mu * sg.x) + cg.x;
          gq[1] = ln_rs * (gq[1] - ln_mu * sg.y) + cg.y;
          gq[2] = ln_rs * (gq[2] - ln_mu * sg.z) + cg.z;
          gq[3] = ln_rs * (gq[3] - ln_mu * sg.w) + cg.w;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[h][0][i][j][e] = hq[e], acc[h][1][i][j][e] = gq[e];
        }
    }
    G3_FENCE();
  }
  uint2 bias_v[2][5];
#pragma unroll
  for (int hp = 0; hp < 2; ++hp)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int T = 5 * wc + j;
      bias_v[hp][j] = *(const uint2*)(smem + G_BIAS + (64 * (T >> 1) + 32 * hp + 16 * (T & 1) + 4 * kq_e) * 2);
    }
  G3_FENCE();
  constexpr int OROW = 160 + 16;                            // bytes per staged output row (80 bf16 + pad)
  unsigned char* stg = smem + wave * 8192;                  // two alternating bands of 16 rows
  auto body = [&](auto tanh_c) __attribute__((always_inline)) {
    constexpr bool TANH = decltype(tanh_c)::value;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        unsigned char* sb = stg + (i & 1) * (16 * OROW);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const uint2 bh = bias_v[0][j], bg = bias_v[1][j];      // zeros without a bias
          float o[4], hq[4], gq[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            hq[e] = LNF ? acc[h][0][i][j][e] : acc[h][0][i][j][e] * p.alpha;    // (LNF: alpha and the fold were applied above)
            gq[e] = LNF ? acc[h][1][i][j][e] : acc[h][1][i][j][e] * p.alpha;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float hv = hq[e], gv = gq[e];
            hv += (e == 0) ? bf_lo(bh.x) : (e == 1) ? bf_hi(bh.x) : (e == 2) ? bf_lo(bh.y) : bf_hi(bh.y);
            gv += (e == 0) ? bf_lo(bg.x) : (e == 1) ? bf_hi(bg.x) : (e == 2) ? bf_lo(bg.y) : bf_hi(bg.y);
            hv = bf2f(f2bf(hv));   // the reference rounds the projection to bf16 before chunk / gelu / mul
            gv = bf2f(f2bf(gv));
            o[e] = hv * bf2f(f2bf(TANH ? gelu_tanh_f(gv) : gelu_erf_f(gv)));
          }
          uint2 pk;
          pk.x = pack_bf2(o[0], o[1]);
          pk.y = pack_bf2(o[2], o[3]);
          *(uint2*)(sb + r16_e * OROW + (16 * j + 4 * kq_e) * 2) = pk;
        }
#pragma unroll
        for (int t2 = 0; t2 < 3; ++t2) {
          const int q = lane_e + 64 * t2;
          if (q >= 160) continue;
          const int row = q / 10, c8 = q - row * 10;
          const int mo = m0 + 64 * wr + 32 * h + 16 * i + row, no = (n0 >> 1) + 80 * wc + c8 * 8;
          *(uint4*)((uint16_t*)p.C + (size_t)mo * p.ldc + no) = *(const uint4*)(sb + row * OROW + c8 * 16);
        }
      }
  };
  if (p.act == DA_ACT_GEGLU) body(std::false_type{});
  else body(std::true_type{});
#undef G3_READ_A
#undef G3_LDSP
#undef G3_TOGGLE
#undef G3_READ_B
#undef G3_MFMA
#undef G3_FENCE
#undef G3_PHASE
#undef G3_READ_A0B0
#undef G3_VMCNT8
#undef G3_SLICE
#endif  // __HIP_DEVICE_COMPILE__
}

template <int PRIO, bool LNF = false>
int launch_geglu_prio(const da_gemm_params& p, hipStream_t s) {
  const int tiles_m = p.M / G_BM, tiles_n = p.N / G_BN;
  const int gx = da_gemm2::choose_xcd_gx2(tiles_m, tiles_n, G_BM, G_BN), gy = 8 / gx;
  const int grid = 8 * ((tiles_m + gy - 1) / gy) * ((tiles_n + gx - 1) / gx);
  auto kern = gemm3_geglu_kernel<PRIO, LNF>;
  constexpr int LDSB = LNF ? G_LDS_LNF : G_LDS;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB) != hipSuccess) return DA_ERR_LAUNCH;
    attr_set = true;
  }
  DA_LAUNCH(kern, dim3(grid), dim3(512), LDSB, s, p, gx);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// Priority form (speed only; DA_K3_PRIO = 0 / 1 / 2 for A/B runs): 1 = the guide's per-phase pair (s_setprio 1 around the 16 MFMAs),
// 0 = none, 2 = ONE s_setprio 1 for the second-dispatched wave row before the loop and no flips (MI355X_MICROARCH.md, "static priority
// for the younger half": waves 4-7 otherwise lose the arbitration at the head of every phase).  Measured, chained launches on random
// operands, 1 / 0 / 2 / 1 / 0 / 2 in one call (profiles/r05k_k3_priority_forms.txt): 8192^3 1484 / 1509 / 1507 / 1486 / 1480 / 1505 TFLOP/s,
// 4096^3 1355 / 1339 / 1387 / 1355 / 1350 / 1350: the static form is the only one that is never behind; default.
template <bool GEGLU>
int launch(const da_gemm_params& p, hipStream_t s) {
  static const int prio = [] { const char* v = getenv("DA_K3_PRIO"); return v ? atoi(v) : 2; }();
  if (prio == 0) return launch_prio<GEGLU, 0>(p, s);
  if (prio == 1) return launch_prio<GEGLU, 1>(p, s);
  return launch_prio<GEGLU, 2>(p, s);
}

// nn.Linear, one ring form (DA_STAGE_LDS_DIRECT), no split-K / LayerNorm fold / transposed block / cross-attention epilogue
int dispatch_lin(const da_gemm_params& p, int tile, int staging, hipStream_t s) {
  if ((tile != DA_TILE_K3_256x256 && tile != DA_TILE_K3_256x320) || staging != DA_STAGE_LDS_DIRECT) return DA_ERR_UNSUPPORTED;
  if (p.conv || p.split_k > 1 || p.stats_out || (p.ln_stats && tile != DA_TILE_K3_256x320) || p.vt || p.xa_k || !da_gemm2::staging_fits(p))
    return DA_ERR_UNSUPPORTED;
  const bool geglu = (p.act == DA_ACT_GEGLU || p.act == DA_ACT_GEGLU_TANH);
  if (tile == DA_TILE_K3_256x320) {   // the GEGLU projection's tile: whole tiles only, 16-byte aligned output rows
    if (!geglu || (p.M % G_BM) || (p.N % G_BN) || (p.ldc & 7) || ((size_t)p.C & 15)) return DA_ERR_UNSUPPORTED;
    static const int prio = [] { const char* v = getenv("DA_K3_PRIO"); return v ? atoi(v) : 2; }();
    if (p.ln_stats) {   // LayerNorm fold, consumer side: 16-byte aligned s / c and partials (da_gemm.hip checked the rest)
      if (((size_t)p.ln_s & 15) || ((size_t)p.ln_c & 15) || ((size_t)p.ln_stats & 15)) return DA_ERR_UNSUPPORTED;
      if (prio == 0) return launch_geglu_prio<0, true>(p, s);
      if (prio == 1) return launch_geglu_prio<1, true>(p, s);
      return launch_geglu_prio<2, true>(p, s);
    }
    if (prio == 0) return launch_geglu_prio<0>(p, s);
    if (prio == 1) return launch_geglu_prio<1>(p, s);
    return launch_geglu_prio<2>(p, s);
  }
  return geglu ? launch<true>(p, s) : launch<false>(p, s);
}

}  // namespace da_gemm3
