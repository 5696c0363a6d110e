/*
 * diffusers_amd -- C ABI of the MI355X (gfx950) denoising hot path.
 *
 * The reference (huggingface/diffusers, 100 % Python) has no FFI: its "operator API" is the set of torch calls its
 * nn.Modules make.  Every entry point below replaces one such call site (cited per function as file:line under the
 * reference's src/diffusers/) with a hand-written HIP kernel.  The boundary is plain C: device pointers, sizes and a
 * hipStream_t passed as void*.  The library never allocates, never synchronises and keeps no mutable global state, so
 * every call can be captured into a HIP graph on the caller's stream.
 *
 * Conventions
 *   - all tensors are bf16 (uint16_t bit patterns) unless a parameter says otherwise; accumulation is fp32
 *   - activations are channels-last: an image tensor is [B][H][W][C], a token tensor [B*S][C]
 *   - return value: 0 = DA_OK, otherwise a DA_ERR_* code (the Python binding raises RuntimeError)
 */
#ifndef DIFFUSERS_AMD_H
#define DIFFUSERS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DA_OK 0
#define DA_ERR_INVALID 1
#define DA_ERR_LAUNCH 2
#define DA_ERR_UNSUPPORTED 3

/* epilogue activations of da_gemm_bf16 */
#define DA_ACT_NONE 0
#define DA_ACT_GEGLU 1     /* out[m][j] = h * gelu_erf(g), weight rows packed [32 h | 32 g] per 64 (activations.py:93-124) */
#define DA_ACT_GELU_TANH 2 /* activations.py:60-90 (approximate="tanh"), Flux / Wan feed-forward */
#define DA_ACT_SILU 3
#define DA_ACT_GELU_ERF 4
#define DA_ACT_QUICK_GELU 5  /* x * sigmoid(1.702 x): transformers ACT2FN["quick_gelu"], the CLIP-L text encoder's MLP */
#define DA_ACT_GEGLU_TANH 6  /* DA_ACT_GEGLU with the tanh GELU on the gate rows: T5 / UMT5 "gated-gelu" feed-forward
                                (gelu_new(wi_0 x) * (wi_1 x), modeling_t5.py T5DenseGatedActDense); rows packed [32 wi_1 | 32 wi_0] */

#define DA_TILE_AUTO 0
#define DA_TILE_128x128 1
#define DA_TILE_64x128 2
#define DA_TILE_128x64 3
#define DA_TILE_64x64 4
#define DA_TILE_256x128 5 /* 8 waves (4x2); LDS-direct staging only */
#define DA_TILE_128x256 6 /* 8 waves (2x4); LDS-direct staging only */
#define DA_TILE_256x256 7 /* 8 waves (2x4), 128x64 per wave; LDS-direct, 2-stage only */
#define DA_TILE_128x128_W8 8 /* 128x128 computed by 8 waves (2x4, 64x32 per wave): twice the waves per staged byte of
                                DA_TILE_128x128, for problems with fewer tiles than CUs where ONE block per CU has to hide the
                                fill latency on its own; LDS-direct, 2 / 3 / 4 ring slots; not for GEGLU epilogues */

/* Second kernel family "K2" (csrc/gemm2_kernel.cuh): 8 waves = 2 K-groups x 4 waves on alternate K slices of one tile,
 * v_mfma_f32_16x16x32_bf16, slice pairs with one mid-pair rendezvous, built for ONE workgroup per CU; tile shapes in
 * 16-column steps.  staging: DA_STAGE_LDS_DIRECT = ring of 2 slice pairs, DA_STAGE_LDS_DIRECT3 = 3 pairs (128x80, 80x128,
 * 128x64).  Not for split_k / LayerNorm fold / paired launches; GEGLU epilogues on DA_TILE_K2_128x128 only.  All K2
 * tiles give bit-identical results; versus the first family the fp32 summation order differs ((even K slices) + (odd)). */
#define DA_TILE_K2_128x128 9  /* wave tile 64 x 64 */
#define DA_TILE_K2_128x80 10  /* wave tile 32 x 80: M 2048 x N 1280 = 256 tiles, one per CU */
#define DA_TILE_K2_128x160 11 /* wave tile 64 x 80: M 8192 x N 640 = 256 tiles */
#define DA_TILE_K2_80x128 12  /* wave tile 80 x 32 (nn.Linear only): the swapped V^T = W_v X^T products */
#define DA_TILE_K2_128x64 13  /* wave tile 64 x 32 */
/* the same loop with all eight waves on every K slice (no K-groups): large tiles for problems with several tiles per CU,
 * where the L2 -> LDS traffic per flop of a 128 x 128 tile is what binds */
#define DA_TILE_K1_128x320 14 /* 4 x 2 waves of 32 x 160: M 2048 x N 10240 = 512 tiles, two per CU */
#define DA_TILE_K1_256x128 15 /* 4 x 2 waves of 64 x 64; 2 or 3 ring slots */
#define DA_TILE_K1_128x256 16 /* 2 x 4 waves of 64 x 64; 2 or 3 ring slots */
#define DA_TILE_K1_256x160 17 /* 4 x 2 waves of 64 x 80 */
#define DA_TILE_K1_256x256 18 /* 2 x 4 waves of 128 x 64 */
#define DA_TILE_K1_256x320 19 /* 4 x 2 waves of 64 x 160, W fragments streamed (nn.Linear; GEGLU with interleaved tile ownership) */
/* Third structure (K3, gemm3.hip; round 5): the 256 x 256 tile as an eight-phase loop, the two wave rows of the workgroup half a
 * phase apart (one multiplies while the other reads LDS and issues LDS-DMA).  nn.Linear only, staging DA_STAGE_LDS_DIRECT only; no
 * split_k / transposed block / cross-attention epilogue; LayerNorm fold: DA_TILE_K3_256x320 as a CONSUMER only (round 6: ln_stats / ln_s /
 * ln_c, 16-byte aligned), neither tile as a producer.  Bit-identical to the K1 tiles above (same K order, same epilogue). */
#define DA_TILE_K3_256x256 20 /* 2 x 4 waves of 128 x 64 (64 contiguous columns per wave: GEGLU pairs inside the wave) */
#define DA_TILE_K3_256x320 21 /* 4 x 2 waves of 64 x 160, GEGLU epilogue ONLY (a wave owns five value tiles and their gate tiles), M % 256
                                 == 0, N % 320 == 0, 16-byte aligned output rows: M 2048 x N 10240 = exactly 256 tiles (SDXL's ff.net.0.proj) */
#define DA_TILE_COUNT 22
/* da_gemm_tune only: which variants compete, given in da_gemm_params.tile (DA_TILE_AUTO = all of them).  Within one family
 * every variant is bit-identical to every other; the two families differ in the fp32 summation order. */
#define DA_TILE_FAMILY_1 (-1)
#define DA_TILE_FAMILY_K2 (-2)

#define DA_STAGE_REGISTER 0   /* global_load_dwordx4 -> ds_write_b128 */
/* LDS-DMA variants: buffer-addressed (buffer_load_dwordx4 ... offen lds: descriptor base at the tile's first operand row,
 * loop-invariant per-lane offsets, scalar K advance, range-check zero fill for conv padding) whenever the tile's operand
 * panels fit 31-bit offsets -- always, for this engine's shapes; per-lane 64-bit pointers (global_load_lds_dwordx4)
 * otherwise, or when DA_GEMM_FLAT_STAGING=1 is set in the environment (A/B measurements).  Both give identical results. */
#define DA_STAGE_LDS_DIRECT 1 /* LDS-DMA, 2-slot ring: prefetch distance 1 */
#define DA_STAGE_LDS_DIRECT3 2 /* LDS-DMA, 3-slot ring: prefetch distance 2, counted vmcnt across the barrier */
#define DA_STAGE_LDS_DIRECT4 3 /* LDS-DMA, 4-slot ring (tiles up to 128x128) */
#define DA_STAGE_LDS_DIRECT6 4 /* LDS-DMA, 6-slot ring (64x128, 128x64, 64x64) */
#define DA_STAGE_LDS_DIRECT8 5 /* LDS-DMA, 8-slot ring (64x64): 112 KiB in flight per CU */
#define DA_STAGE_PINGPONG 6    /* K2 tiles: 2-pair ring, the two K-groups half an iteration apart (each stages its own slices) */
#define DA_STAGE_PINGPONG3 7   /* K2 tiles: 3-pair ring, ditto */

/* ABI version: bumped on EVERY change of a parameter struct's layout or of an entry point's signature.  A host checks it once after
 * loading the library -- `da_version() == DA_ABI_VERSION` of the header it was compiled against, and, for hosts that mirror the
 * structs by hand (ctypes, JNI, cgo), `da_sizeof_*()` against their own sizeof -- because the structs carry no size field: a host
 * built against an older header would pass shorter structs and the library would read garbage for the new members
 * (da_gemm_params.vt is a STORE address).  History: 1 = rounds 1-3; 4 = round 4 (da_gemm_params.vt / vt_col0 / ld_vt,
 * da_attention_params.algo); 5 = round 5; 6 = round 6 (this header: da_attention_params.split_ws / split_ws_bytes / kv_split; da_groupnorm_nhwc_bf16 takes `sync`). */
#define DA_ABI_VERSION 6
int da_version(void);
size_t da_sizeof_gemm_params(void);
size_t da_sizeof_attention_params(void);
/* Timing hook (no reference counterpart; bench.py's roofline legs): arm two HIP events (hipEvent_t, created by the caller with
 * timing enabled) for the calling thread.  The NEXT kernel this library launches on that thread records `start_event` at the begin
 * of its dispatch, and every launch until the pair is cleared with (NULL, NULL) records `stop_event` at the end of its dispatch
 * (hipExtLaunchKernelGGL): after a synchronisation their elapsed time is the execution time of the entry point's kernel(s) as the
 * profiler reports it -- an event pair recorded around the call from the host also counts the marker packets' own processing.
 * Both or neither must be NULL.  Launch results do not depend on it. */
int da_set_launch_events(void* start_event, void* stop_event);
/* Experiment hook (no reference counterpart; tools/probe_anyorder.py): the `flags` word hipExtLaunchKernelGGL receives for every
 * later launch of the calling thread (1 = hipExtAnyOrderLaunch: the dispatch packet is queued WITHOUT its barrier bit, so it may
 * start before earlier launches of its stream have finished -- only for launches the caller knows to be independent).  0 = the
 * ordinary in-order launch (default).  Not recorded by launch plans. */
int da_set_launch_flags(unsigned flags);
/* Box normaliser (no reference counterpart; bench.py `config.box_mfma_tflops`): a register-only loop of
 * v_mfma_f32_32x32x16_bf16 -- `blocks` workgroups of four waves, `iters` x 16 MFMAs per wave, no memory traffic.  Its rate is the
 * matrix pipe's issue rate times the clock THIS box sustains under matrix load; the pool's boxes differ by +- 8 % on one build, so
 * cross-box lines are compared after dividing by it.  *flop (may be NULL): the launch's work.  `sink`: any device address of >= 4
 * bytes (never written). */
int da_mfma_probe(int blocks, int iters, void* sink, double* flop, void* stream);
/* name of the HIP runtime error behind this thread's most recent DA_ERR_LAUNCH (diagnostics only) */
const char* da_last_error(void);

/* ------------------------------------------------------------------------------------------------------------------
 * da_gemm_bf16: C[M][N] = epilogue( alpha * A[M][K] . W[N][K]^T )
 *   conv == 0  nn.Linear: A is [M][lda].  Replaces F.linear at attention_processor.py:2743-2777 (to_q/k/v/out),
 *              activations.py:113-124 (GEGLU proj), attention.py:1717-1742 (FeedForward), transformer_2d.py:466-512
 *              (proj_in / proj_out), embeddings.py:1262-1308 (TimestepEmbedding, large-M case), resnet.py:345-349.
 *   conv == 3  nn.Conv2d(k=3, pad, stride 1|2) on channels-last input, as an implicit GEMM with K = 9*(C1+C2)
 *              (conv == 1: the 1x1 shortcut conv of resnet.py:373 over the same two-source gather):
 *              A  = source 1 [B][Hin][Win][C1], A2 = optional source 2 [B][Hin][Win][C2] (fused torch.cat on channels,
 *              unet_2d_blocks.py:2444/:2561), W = [N][3][3][C1+C2]; up == 1 fuses F.interpolate(scale 2, nearest)
 *              (upsampling.py:177-190); stride 2 = Downsample2D (downsampling.py:130-147).  M = B*Hout*Wout.
 *              Replaces resnet.py:340/:365/:373, unet_2d_condition.py:1108/:1230, vae.py:286/:309.
 *   epilogue   + bias[N] + bias_rows[M] + rowvec[m / rows_per_batch][N] (time-embedding injection, resnet.py:345-349),
 *              activation, * gate[m / rows_per_batch][N] (adaLN-Zero gates, transformer_flux.py:400-405,:470-494;
 *              transformer_wan.py:488-502), + residual[M][ldr] (resnet.py:375, attention.py:1034-1080,
 *              transformer_2d.py:499-512), * out_scale.  bias_rows is the bias of a SWAPPED product (V^T = W_v . X^T:
 *              the Linear's bias runs along M).  With a gate the linear output and the gated value are each rounded to
 *              bf16 first, as the reference's separate torch ops round them.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct da_gemm_params {
  const void* A;
  const void* A2;
  const void* W;
  void* C;
  const void* bias;     /* [N] or NULL */
  const void* rowvec;   /* [M / rows_per_batch][ld_rowvec] or NULL */
  const void* residual; /* [M][ldr] or NULL; may be C itself (ldr == ldc): every element is read and written by one lane,
                           so the launch accumulates in place (temporal taps of WanCausalConv3d, autoencoder_kl_wan.py:131-173) */
  const void* bias_rows; /* [M] or NULL */
  const void* gate;      /* [M / rows_per_batch][ld_gate] or NULL (bf16, or fp32 when gate_f32) */
  int M, N, K;
  int lda, ldw, ldc, ldr, ld_rowvec, ld_gate;
  int rows_per_batch;
  float alpha;     /* 0 -> 1 */
  float out_scale; /* 0 -> 1 */
  int act;         /* DA_ACT_* */
  int out_f32;     /* 1: C is float [M][ldc] */
  int conv;        /* 0 linear; 1 or 3 = implicit-GEMM conv with that kernel size (pad given by `pad`) */
  int Hin, Win, C1, C2, Hout, Wout, stride, up, pad;
  int tile;    /* DA_TILE_* */
  int staging; /* DA_STAGE_* */
  int gate_f32; /* 1: gate is float and out = residual + bf16(xW+b) * gate in fp32, rounded once
                   (WanTransformerBlock, transformer_wan.py:491,:502); 0: Flux rounding (gate product rounded to bf16) */
  int split_k;  /* 0 / 1: every tile is computed by one block.  2..DA_SPLITK_MAX (24): the K range of each tile is dealt to
                   split_k co-resident blocks that hand fp32 partial tiles over through `workspace` (in-launch reduction,
                   fixed summation order: deterministic, but the last fp32 bit of a sum differs from split_k = 1).  For
                   problems with fewer tiles than the 256 CUs (SDXL: M = 2048, N = 1280 Linear and the 1280-channel convs). */
  void* workspace;          /* split_k > 1: caller-owned device buffer, >= tiles * (split_k - 1) * tile_rows * tile_cols * 4 B */
  void* sync_flags;         /* split_k > 1: DA_SPLITK_FLAGS ints, zeroed ONCE by the caller; the kernel re-arms what it uses */
  long long workspace_bytes;
  /* ---- LayerNorm folded into the GEMMs either side of it (nn.Linear, bf16 output; attention.py:1030,:1056: norm2 ->
   * attn2.to_q, norm3 -> ff.net.0.proj).  LN(x) W^T = rstd (x (gamma o W)^T - mu s^T) + c^T with s[n] = sum_k (gamma o W)[n,k],
   * c[n] = sum_k beta[k] W[n,k]: the normalised tensor is never written or read.
   *   PRODUCER (the GEMM that writes x): stats_out != NULL -> besides C, every column tile writes the partial (sum, sum of
   *     squares) of ITS columns of each row of the bf16-rounded output: float2 stats_out[m * stats_ld + 2 * column_tile]
   *     (da_gemm_stats_parts() = number of column tiles for *p; combined across a block's waves in LDS in fixed order, no
   *     atomics: deterministic).
   *   CONSUMER (the GEMM that reads LN(x)): ln_stats != NULL -> each row's mean / rstd are formed from its ln_parts partials
   *     (fixed order) while the first K slices are in flight, and the epilogue applies the identity above to alpha * acc
   *     before bias / activation (both GEGLU halves included).  W must be the pre-scaled (gamma o W) in bf16. */
  float* stats_out;
  int stats_ld;             /* floats per row of stats_out (>= 2 * parts) */
  const float* ln_stats;    /* partials written by the producer of this GEMM's A operand; every row must hold
                               DA_LN_MAX_PARTS (sum, sum of squares) slots (ln_stats_ld >= 2 * DA_LN_MAX_PARTS): the
                               kernel loads whole 16-byte pairs of slots and masks the ones past ln_parts */
  int ln_stats_ld, ln_parts;
  const float* ln_s;        /* [N] fp32 */
  const float* ln_c;        /* [N] fp32 */
  float ln_eps;
  int k_valid; /* 0: every K element counts.  > 0: only the first k_valid elements of every K PERIOD are non-zero -- a period is
                  one kernel tap's channels (C1, conv with C2 == 0) or the whole K (Linear) -- because the channel count was
                  zero-padded up to the 64-wide K granule (AutoencoderKLWan's 96-channel stage: k_valid = 96 in periods of
                  128).  The operand values past k_valid MUST be zeros in both operands' layouts (activation channels and
                  weight columns); the kernel then runs half the MFMA steps on a slice that is half padding and none on a
                  slice that is all padding.  Results equal k_valid = 0 (adding products of zeros changes nothing). */
  const void* prefetch;     /* optional (second kernel family): a tensor a LATER launch will stream from HBM -- the next layer's
                               weight.  Every workgroup reads its share of the first prefetch_bytes behind its K loop (LDS-DMA into a
                               scratch KiB, nothing waits for it), so the lines sit in the memory-side cache when that launch starts:
                               inside a denoising step every weight is otherwise met cold (5 GB of weights per step against a 256 MB
                               cache).  Speed only; the bytes are never interpreted. */
  long long prefetch_bytes; /* multiple of 16 */
  /* Transposed column block (round 4, second kernel family, nn.Linear): output columns n >= vt_col0 are written TRANSPOSED,
   * vt[(n - vt_col0) * ld_vt + m] (bf16), instead of into C -- the fused Q | K | V projection of a self-attention layer
   * (attention_processor.py:2743-2751: to_q / to_k / to_v on the same input) in ONE GEMM whose V block leaves in the
   * [channel][token] layout the flash kernel consumes, replacing the paired Q|K + swapped V^T launch.  vt_col0 must be a multiple
   * of the tile's column count (80 / 160) and of 16; C needs only vt_col0 columns.  bias / alpha / the LayerNorm fold apply to
   * those columns as to the others; residual, gate, activation and statistics do not combine with it. */
  void* vt;
  int vt_col0;
  long long ld_vt;
  /* Cross-attention in the epilogue (round 5, second kernel family, tile DA_TILE_K2_128x128, nn.Linear): xa_k != NULL makes this
   * launch attn2.to_q AND the attention it feeds (attention_processor.py:2743-2777 with encoder_hidden_states): C receives
   * softmax(scale * q k^T) v, [M][N] with heads of 64 channels side by side, where q = the launch's ordinary result (alpha, LayerNorm
   * fold, bias; rounded to bf16 as the reference's to_q output is).  xa_k = K [batches * xa_skv_alloc][xa_k_ld] (row = key, the
   * heads' channels side by side as in C), xa_vt = V^T [N][xa_vt_ld] (row = channel, column = batch * xa_skv_alloc + key): the
   * step-invariant projections of the text embeddings (layers.CrossKV).  Keys >= xa_skv are masked; rows past them up to
   * xa_skv_alloc must exist (zero rows of K, zero columns of V^T).  Batch of a row = m / rows_per_batch (a multiple of 128 rows).
   * xa_skv_alloc <= 80 (the 77 CLIP tokens of the SD / SDXL U-Nets), a multiple of 8.  The softmax over so few keys is one exact
   * pass (no running maximum); probabilities are rounded to bf16 for the P . V product like the flash kernels' (rel. rms vs fp32
   * within the same tolerance, not bit-identical to da_attention_bf16: another summation order).  Does not combine with residual,
   * gate, rowvec, activation, statistics, vt. */
  const void* xa_k;
  const void* xa_vt;
  int xa_skv, xa_skv_alloc, xa_k_ld;
  long long xa_vt_ld;
  float xa_scale;
} da_gemm_params;

/* number of stats partials per row the launch *p (tile resolved as da_gemm_bf16 resolves it) writes to stats_out */
int da_gemm_stats_parts(const da_gemm_params* p);

#define DA_LN_MAX_PARTS 64   /* slots per row of a statistics buffer */
#define DA_LN_PAIR_LOADS 6   /* 16-byte loads per lane half in the consumer: rows with up to 24 partials (N <= 24 column tiles) */
#define DA_SPLITK_FLAGS 4096
#define DA_SPLITK_MAX 24   /* largest split_k (round 6: 8 -> 24 for the M <= 128 deep-K convs of the SD1.5 / DDPM U-Nets) */
#define DA_SPLITK_ERR_SLOT (DA_SPLITK_FLAGS - 1) /* set to 1 by a reducer whose producer never arrived (bounded spin) */

int da_gemm_bf16(const da_gemm_params* p, void* stream);

/* Two independent nn.Linear problems in ONE launch (same tile / staging variant, taken from *a; split_k must be 0 / 1):
 * blocks [0, grid_a) work on *a, the rest on *b, each exactly as in its own launch -- bit-identical results, better
 * fill when neither problem has 256 tiles.  The engine pairs the fused Q|K projection with the swapped V^T projection of
 * a self-attention layer (attention_processor.py:2743-2751: three F.linear calls on the same input). */
int da_gemm_pair_bf16(const da_gemm_params* a, const da_gemm_params* b, void* stream);

/* Times every (tile, staging) variant able to run *p on `stream` (HIP events; one warm launch + min of `iters` timed
 * launches each) and returns the fastest in *best_tile / *best_staging (and its time in *best_us, may be NULL).
 * `scratch` (may be NULL) is a caller-owned device buffer, ideally larger than the 256 MiB Infinity Cache, that is
 * memset before every timed launch so the operands are fetched from HBM as they are inside the denoising loop.
 * Every variant walks K in the same order with the same MFMA, so all of them write bit-identical C: the choice
 * changes speed only.  The library keeps no tuning state: the caller owns the table (diffusers_amd/tuning.py keeps
 * it per problem shape, the role torch's cublasLt/hipblasLt heuristic cache plays for F.linear / F.conv2d in the
 * reference).  Synchronises the stream; must not be called while the stream is being captured into a graph. */
/* `pair` (may be NULL): time the two problems as ONE da_gemm_pair_bf16 launch.  `best_split` (may be NULL): when given and
 * p->workspace / p->sync_flags are set, split_k = 2, 3, 4, 6, 8, 12, 16, 24 variants of every admissible tile are timed as well and the
 * winner's split factor is returned (1 = unsplit); otherwise only split_k = 1 is considered. */
int da_gemm_tune(const da_gemm_params* p, const da_gemm_params* pair, void* stream, int iters, void* scratch,
                 size_t scratch_bytes, int* best_tile, int* best_staging, int* best_split, float* best_us);

/* ------------------------------------------------------------------------------------------------------------------
 * da_attention_bf16: out = softmax(scale * Q K^T) V, flash-style (no S x S tensor), no mask / dropout / causal.
 *   Replaces F.scaled_dot_product_attention at attention_processor.py:2767 (AttnProcessor2_0) and
 *   attention_dispatch.py:3709 (_native_attention: Flux / Wan).
 *   q   element (b, s, h, d) at q  + b*q_batch_stride + s*q_row_stride + h*D + d
 *   k   element (b, s, h, d) at k  + b*k_batch_stride + s*k_row_stride + h*D + d
 *   vt  element (b, s, h, d) at vt + (h*D + d)*vt_ld + b*vt_batch_stride + s        (V transposed: keys contiguous)
 *   out element (b, s, h, d) at out + b*o_batch_stride + s*o_row_stride + h*D + d
 *   Skv = number of keys attended; Skv_alloc (multiple of 8, >= Skv) = keys present in memory per batch.
 *   D in {64, 96, 128, 160} (SD1.5's head dims 40 / 80 are zero-padded to 64 / 96 by the host-side weight packing).
 *   All strides in elements, multiples of 8 (o: 4).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct da_attention_params {
  const void* q;
  const void* k;
  const void* vt;
  void* out;
  int B, H, Sq, Skv, Skv_alloc, D;
  long long q_batch_stride, k_batch_stride, vt_batch_stride, o_batch_stride;
  int q_row_stride, k_row_stride, vt_ld, o_row_stride;
  float scale;
  int ring_slots; /* K / V^T LDS ring depth: 0 = default for the head size (deepest that fits), 2..4 = pinned (speed only:
                     every depth computes the same tiles in the same order -> bit-identical outputs) */
  /* ---- masked variant (D = 64 only; the text encoders either side of the hot path) ----
   * causal != 0: key j is visible to query i iff j <= i (CLIP text transformer, modeling_clip.py create_causal_mask).
   * bias != NULL: scores = scale * q.k^T + bias[b][h][i][j] before the softmax: T5 / UMT5 relative position bias with
   *   scale = 1 (modeling_t5.py T5Attention: no 1/sqrt(d)), optionally with a key-padding mask folded in (entries <= -1e29
   *   are treated as -inf).  Element type bf16 (bias_f32 == 0) or fp32; rows of bias_row_stride >= ceil64(Skv) elements,
   *   or bias_row_stride == 0: ONE row shared by every query (a key-padding mask such as UNet2DConditionModel's
   *   encoder_attention_mask, unet_2d_condition.py:1071-1073: 0 for kept keys, -10000 for discarded ones). */
  const void* bias;
  long long bias_batch_stride, bias_head_stride; /* elements; 0 = shared across batches / heads */
  int bias_row_stride, bias_f32, causal;
  int q_block; /* queries per workgroup: 0 = default, 128 = four waves, 64 = two waves (first generation: D = 64, unmasked,
                  ring depth 2 only; other cases run 128), 256 = eight waves (second generation, see algo).  Speed only: each wave owns 32 queries and walks the same K / V^T tiles in
                  the same order either way -> bit-identical outputs */
  int pv_delay; /* 0 = default for the head size, 1 = on, -1 = off.  On (D = 64 / 128, unmasked, 3 ring slots): the P.V
                   product of K / V tile j - 1 is issued after the Q.K^T product of tile j so that it runs under tile j's
                   softmax arithmetic.  2 (4 ring slots): tile j + 1's Q.K^T product rides in those softmax slices as well
                   (measured: no faster -- the loop is bound by its instruction issue, not by latency; kept as a variant).
                   Speed only: the same operations in the same order per accumulator. */
  int algo; /* kernel generation.  0 = default: the second-generation kernel (attention2.hip: permuted-key Q.K^T so that V^T
               fragments are single 16-byte reads, buffer-addressed staging, deferred running maximum, whole-row output stores,
               128- or 256-query workgroups) wherever it exists -- unmasked, D = 64 / 128, no first-generation variant pinned
               through ring_slots = 2 / q_block = 64 / pv_delay != 0 -- else the first-generation kernel;
               1 = first generation (attention.hip) always; 2 = second generation or DA_ERR_UNSUPPORTED;
               3 = second generation with the softmax shift folded into the Q.K^T product ("AUG": Q pre-multiplied by
               scale * log2 e and rounded to bf16 once more, -m as one extra k-step);
               4 = second generation with the row sum taken by the matrix pipe ("RSM": a ones row next to V^T, l = the sum of the
               bf16-rounded probabilities); 5 = both.
               The generations agree to bf16 rounding of the output, not bit for bit (different shift, different order of
               the row sum); within the second generation q_block 128 / 256 and ring_slots 3 / 4 are bit-identical. */
  /* ---- key-split tail (round 6; second generation only; speed only up to fp32 summation order) ----
   * A launch has nb = B * H * ceil(Sq / q_block) query blocks for a chip of 256 CUs; nb = 320 (SDXL, S = 1024) leaves 64 CUs with two
   * blocks and 192 with one, nb = 432 (Flux) runs two rounds of which the second is 69 % full.  With a workspace the library runs the
   * first nb - nb % CUs blocks whole and splits the keys of each of the remaining nb % CUs blocks over `s` workgroups (s chosen so
   * that the split units cover the chip once more with short units); the units of a block publish (O, m, l) partials in fp32 with
   * write-through stores, draw a ticket from a per-block counter, and the LAST arriver combines the partials of all s units in
   * unit order (a fixed order: the result does not depend on which unit arrives last) and writes the block's rows.
   * split_ws: device workspace, or NULL = never split.  Its first DA_ATTN_SPLIT_COUNTER_BYTES hold the ticket counters: the
   *   caller zeroes them ONCE (the kernel re-arms them); the rest is scratch.  One workspace per stream (launches on a stream are
   *   ordered).  A workspace that is too small for a launch's plan just disables the split for that launch.
   * kv_split: 0 = the library's choice, 1 = off, 2..8 = that many units per tail block (tests). */
  void* split_ws;
  long long split_ws_bytes;
  int kv_split;
} da_attention_params;
#define DA_ATTN_SPLIT_COUNTER_BYTES 16384
/* bytes of split_ws this launch would use (0: it would not split), and the split it would choose; for tests / sizing */
long long da_attention_split_plan(const da_attention_params* p, int* full_blocks, int* tail_blocks, int* units_per_tail_block);

int da_attention_bf16(const da_attention_params* p, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Normalisation (norm.hip)
 *   da_groupnorm_nhwc_bf16   nn.GroupNorm(G, C, eps) [+ SiLU when act == DA_ACT_SILU] on channels-last [B][HW][C].
 *                            Replaces resnet.py:326-327,:350,:362, transformer_2d.py:466, unet_2d_condition.py:1228-1229,
 *                            vae.py:305-308, attention_processor.py:2740.  workspace: da_groupnorm_workspace_bytes().
 *                            x2 != NULL: channels [0,C1) come from x and [C1,C) from x2 (fused skip concat,
 *                            unet_2d_blocks.py:2444).
 *   da_layernorm_bf16        nn.LayerNorm(C, eps) over rows of [M][ldx] (gamma/beta may be NULL = no affine), optional
 *                            AdaLN modulation y = LN(x) * (1 + mod_scale[b]) + mod_shift[b], b = row / rows_per_batch.
 *                            mod_f32 == 0: bf16 vectors, LN(x) rounded to bf16 before the modulation (the reference's
 *                            separate bf16 ops); mod_f32 == 1: fp32 vectors and fp32 math throughout (FP32LayerNorm +
 *                            fp32 scale_shift_table, transformer_wan.py:488-502,:723).
 *                            Replaces attention.py:986,:1030,:1056 and normalization.py:157-170,:194-202,:346-351.
 *   da_rmsnorm_rope_bf16     per-head RMSNorm (torch.nn.RMSNorm(head_dim), transformer_flux.py:316-317,:101-102 ;
 *                            "rms_norm_across_heads" when heads == 1, transformer_wan.py:83-84) followed by rotary
 *                            position embedding (apply_rotary_emb, embeddings.py:1187-1230, interleaved pairs) IN PLACE
 *                            on `parts` column blocks (query, key) of a token-major buffer x[rows][ld]:
 *                            block j starts at column col_off[j] and holds heads*D channels, weight[j] is its [D]
 *                            scale (NULL = none).  cos/sin are fp32 [>= rope_row0 + rows][D] tables (NULL = no
 *                            rotation); row r of x uses table row rope_row0 + (r % rows_per_batch).
 *   da_softmax_rows_f32_bf16 row softmax of fp32 scores -> bf16 (single-head D=512 VAE mid-block attention,
 *                            attention_processor.py:2767 via vae.py / unet_2d_blocks.py:736-748).
 * ------------------------------------------------------------------------------------------------------------------ */
size_t da_groupnorm_workspace_bytes(int B, int HW, int C, int G);
/* sync (round 6; may be NULL): a device buffer of da_groupnorm_sync_bytes() bytes, ZEROED ONCE by the caller and then owned by the
 * library's kernels (monotonic arrival counters + fp64 partial sums), one per stream.  With it, tensors whose per-(batch, group set)
 * slab exceeds one CU's LDS but that fit the chip's (<= 256 LDS-resident workgroups) are normalised in ONE launch that reads the tensor
 * once: the parts of a slab exchange their partial statistics through `sync` and wait for each other (all of them are resident).
 * Without it those tensors take the two-kernel form (statistics pass + apply pass = one more read).  Same formula and rounding points
 * either way; the statistics are summed in another order (outputs agree to a bf16 ulp). */
size_t da_groupnorm_sync_bytes(void);
int da_groupnorm_nhwc_bf16(const void* x, const void* x2, int C1, const void* gamma, const void* beta, void* y,
                           void* workspace, int B, int HW, int C, int G, float eps, int act, void* sync, void* stream);
/* T5LayerNorm (modeling_t5.py: no mean subtraction, no bias): y = bf16(bf16(x * rsqrt(mean(x^2) + eps)) * gamma), the two
 * roundings of the reference's fp32 -> weight-dtype cast followed by the bf16 multiply. */
int da_rmsnorm_bf16(const void* x, const void* gamma, void* y, int M, int C, int ldx, int ldy, float eps, void* stream);
int da_layernorm_bf16(const void* x, const void* gamma, const void* beta, void* y, const void* mod_scale,
                      const void* mod_shift, int mod_ld, int mod_f32, int rows_per_batch, int M, int C, int ldx, int ldy,
                      float eps, void* stream);
int da_rmsnorm_rope_bf16(void* x, int ld, int rows, int rows_per_batch, int heads, int D, int parts, const int* col_off,
                         const void* const* weight, float eps, const float* cos, const float* sin, int rope_row0,
                         int do_norm, void* stream);
int da_softmax_rows_f32_bf16(const void* scores, void* probs, int M, int N, long long ld, long long ldo, void* stream);
/* WanRMS_norm over the channels of channels-last rows x[rows][C] (autoencoder_kl_wan.py:198-206, bias-free):
 * y = bf16(bf16(bf16(x / max(||x||_2, 1e-12)) * scale) * gamma) [+ SiLU, :352,:367,:899]; scale = sqrt(real channel
 * count) (zero-padded channels with gamma 0 stay 0).  C a multiple of 8, <= 1024. */
int da_rmsnorm_channels_bf16(const void* x, const void* gamma, void* y, long long rows, int C, float scale, int act,
                             void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Sampler (sampler.hip): fused classifier-free-guidance combine + scheduler.step.
 *   `table` is a device array of 8 floats per step, `step_idx` a device int (graph-replay safe).  When cfg != 0 the
 *   model output holds [uncond | cond] halves of n elements each and noise_pred = u + guidance * (c - u)
 *   (pipeline_stable_diffusion.py:1054-1055).  dtype selects bf16 or fp32 tensors.
 *   da_euler_scale_model_input  scheduling_euler_discrete.py:326-348, output replicated `rep` times (torch.cat([x]*2))
 *                               row = [sigma, sigma_next, dt, sqrt(sigma^2+1), c_out, sigma^2+1, -, timestep]
 *   da_euler_step               scheduling_euler_discrete.py:685-800 (gamma = 0); pred_type = DA_PRED_* selects the
 *                               pred_original_sample form of :760-775 (c_out = -sigma / sqrt(sigma^2+1) for v_prediction)
 *   da_x0_linear_step           scheduling_ddim.py:384-514 and scheduling_ddpm.py:461-567; pred_type = DA_PRED_* selects
 *                               the x0 / pred_epsilon forms of scheduling_ddim.py:455-468
 *                               row = [sqrt(beta_t), sqrt(alpha_t), k0, ke, kx, kn, clip_range, timestep]
 *                               noise (may be NULL) + step * noise_step_stride elements = this step's variance noise
 *                               (stride 0: one buffer refilled by the host per step; > 0: all steps pre-drawn)
 *   da_flowmatch_step           scheduling_flow_match_euler_discrete.py:423-523 ; row = [sigma, sigma_next, dt, ...]
 *   da_advance_step             the reference's `self._step_index += 1`
 * ------------------------------------------------------------------------------------------------------------------ */
#define DA_DTYPE_BF16 0
#define DA_DTYPE_F32 1
#define DA_PRED_EPSILON 0
#define DA_PRED_V 1
#define DA_PRED_SAMPLE 2
int da_euler_scale_model_input(const void* x, void* out, const float* table, const int* step_idx, int rep, long long n,
                               int dtype, void* stream);
int da_euler_step(const void* eps, const void* x, void* out, const float* table, const int* step_idx, int cfg,
                  float guidance, long long n, int dtype, int pred_type, void* stream);
int da_x0_linear_step(const void* eps, const void* x, const void* noise, long long noise_step_stride, void* out,
                      const float* table, const int* step_idx, int cfg, float guidance, long long n, int dtype,
                      int pred_type, void* stream);
/* dtype = dtype of the model output v AND of out; x_dtype = dtype of the sample (fp32 sample + bf16 model output is the
 * reference's Wan hand-over: the update is formed in fp32 and stored in the model output's dtype, :484,:517) */
int da_flowmatch_step(const void* v, const void* x, void* out, const float* table, const int* step_idx, int cfg,
                      float guidance, long long n, int dtype, int x_dtype, void* stream);
/* da_unipc_flow_step: schedulers/scheduling_unipc_multistep.py:760-1300 (flow_prediction, predict_x0, B(h),
 * solver_order <= 2): x0 conversion + corrector + predictor + history roll in ONE pass, in place on x / last / m1 / m2;
 * coef = 16 floats per step [sigma, use_corr, order_c, c1c, c2c, c3c, rk_c, rho0_c, rho_last_c, order_p, c1p, c2p, c3p,
 * rk_p, -, -]; x_dtype (sample + history) / v_dtype (model output): f32/f32, f32/bf16 (Wan pipeline) or bf16/bf16. */
int da_unipc_flow_step(const void* v, void* x, void* last, void* m1, void* m2, const float* coef, const int* step_idx,
                       int cfg, float guidance, long long n, int x_dtype, int v_dtype, void* stream);
int da_advance_step(int* step_idx, void* stream);
/* rescale_noise_cfg (pipelines/stable_diffusion/pipeline_stable_diffusion.py:69-92; SDXL :1227-1229, SD :1057-1059): eps
 * [2][B][n_per] = (uncond, cond) -> out [B][n_per] = guidance_rescale * (cfg * std(text) / std(cfg)) + (1 - guidance_rescale) * cfg
 * with cfg = uncond + guidance * (cond - uncond), every torch op rounded in the tensor dtype; ratio_ws: B floats of scratch.
 * The step kernels then run with cfg = 0 on `out`. */
int da_cfg_rescale(const void* eps, void* out, float* ratio_ws, int B, long long n_per, float guidance, float guidance_rescale,
                   int dtype, void* stream);
/* out[r][:] = bf16(x) for r < rep: latents.to(transformer_dtype) (pipeline_wan.py:600) + CFG batch doubling */
int da_cast_f32_bf16(const float* x, void* out, int rep, long long n, void* stream);
/* out[r][:] = x * s in the tensor dtype for r < rep: latents * scheduler.init_noise_sigma
 * (pipeline_stable_diffusion.py:713); rep = 2, s = 1 is the CFG batch doubling torch.cat([latents] * 2) (:1037) */
int da_mul_scalar(const void* x, void* out, float s, int rep, long long n, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Misc (misc.hip)
 *   da_timestep_embedding   get_timestep_embedding, embeddings.py:27-78.  t[B] (device floats) or table/step_idx
 *                           (column 7 of the sampler table).  Output [B][dim] bf16 or fp32.
 *   da_linear_small_m_bf16  M <= 8 Linear with optional SiLU on the input and SiLU/GELU on the output and a residual:
 *                           TimestepEmbedding (embeddings.py:1262-1308), add_embedding (unet_2d_condition.py:906-922),
 *                           time_emb_proj(SiLU(temb)) (resnet.py:345-349), AdaLN linears (normalization.py:157-202).
 *   da_conv_thin_in_bf16    Conv2d with Cin <= 16, k in {1,3}: conv_in (unet_2d_condition.py:1108, vae.py:286),
 *                           post_quant_conv (autoencoder_kl.py:204).  Input NCHW or NHWC, output NHWC.
 *                           in_div != 1: input is first divided by it in bf16 (latents / scaling_factor,
 *                           pipeline_stable_diffusion_xl.py:1283), then in_add is added in bf16 (+ shift_factor,
 *                           pipeline_flux.py:960).
 *   da_conv_thin_out_bf16   Conv2d 3x3 with Cout in {3,4,8,16}: conv_out (unet_2d_condition.py:1230, vae.py:309).
 *                           Input NHWC, output NCHW (bf16 or fp32).
 * ------------------------------------------------------------------------------------------------------------------ */
/* out[b][i] = (float)a[i] + (float)m[b][i]: scale_shift_table + temb.float() (transformer_wan.py:483-485,:719) */
int da_bcast_add_f32(const float* a, const void* m_bf16, float* out, int B, int n, void* stream);
/* Conv3d patch embedding with kernel == stride as a gather + GEMM (transformer_wan.py:598,:663-664): tokens[b*f*h*w + ...]
 * [c][dt][dh][dw] <- x[b][c][F][H][W]; and its inverse for the output (:727-731): x[b][c][F][H][W] <- tokens[..][dt][dh][dw][c] */
int da_patchify3d_bf16(const void* x, void* tokens, int B, int C, int F, int H, int W, int pt, int ph, int pw, void* stream);
int da_unpatchify3d_bf16(const void* tokens, void* x, int B, int C, int F, int H, int W, int pt, int ph, int pw,
                         void* stream);
/* out[c][r] = in[r][c]: turns a token-major value tensor into the V^T operand of da_attention_bf16 (attention backends
 * that receive (B, S, H, D) tensors, attention_dispatch.py:494-515) */
int da_transpose_bf16(const void* in, void* out, int R, int C, long long ldi, long long ldo, void* stream);
/* out[b][c][p] = in[b][p][c], c < cout <= 8, rows of `cpad` channels (cpad % 8 == 0): the NCHW planes of a thin-output conv
 * (conv_out 320 -> 4 of the U-Net, unet_2d_condition.py:1230; 128 -> 3 of the VAE decoder, vae.py:309) that ran on the
 * implicit-GEMM kernel with its output channels zero-padded to 16 */
int da_nhwc_take_nchw_bf16(const void* in, void* out, long long B, long long HW, int cpad, int cout, void* stream);
/* The same pass with VaeImageProcessor.postprocess (image_processor.py:738-786) as its epilogue: the decoder's conv_out result
 * (bf16, padded rows) is denormalised ((x * 0.5 + 0.5).clamp(0, 1)) and written ONCE in the caller's layout -- mode 0: NCHW
 * fp32 ("pt"), 1: NHWC fp32 ("np"), 2: NHWC uint8 (the bytes numpy_to_pil hands to PIL).  cout <= 4.  Same values as
 * da_nhwc_take_nchw_bf16 followed by da_image_postprocess. */
int da_nhwc_take_postprocess(const void* in, void* out, long long B, long long HW, int cpad, int cout, int mode, void* stream);
/* dst[D0][D2][D1][D3] = src[D0][D1][D2][D3] (bf16, D3 % 8 == 0): the channel-halves -> frame pairs interleave of
 * WanResample 'upsample3d' (autoencoder_kl_wan.py:297-299) on channels-last frames. */
int da_permute_0213_bf16(const void* src, void* dst, long long D0, int D1, int D2, int D3, void* stream);
/* Channels-last frames src[B*T][HW][Cs] (channels [0, C), C <= 4 <= Cs) -> video dst[B][C][T][HW] (bf16, or fp32 when
 * out_f32), clamped to [lo, hi]: the output layout + torch.clamp of AutoencoderKLWan._decode (autoencoder_kl_wan.py:1210). */
/* VaeImageProcessor.postprocess (image_processor.py:738-786) on the decoded image / video img[B][C][HW] (bf16, or fp32
 * when in_f32): v = clamp(x * 0.5 + 0.5, 0, 1) (denormalize, :222-234); mode 0 "pt": out[B][C][HW] fp32; mode 1 "np":
 * out[B][HW][C] fp32 (pt_to_numpy, :191-204); mode 2 "pil" bytes: out[B][HW][C] uint8 = round-half-even(255 v)
 * (numpy_to_pil, :128-149).  C <= 4. */
int da_image_postprocess(const void* img, void* out, int B, int C, long long HW, int in_f32, int mode, void* stream);
int da_frames_to_ncthw_bf16(const void* src, void* dst, int B, int T, long long HW, int Cs, int C, float lo, float hi,
                            int out_f32, void* stream);
int da_timestep_embedding(const float* t, const float* table, const int* step_idx, void* out, int B, int dim,
                          int flip_sin_to_cos, float shift, float scale, float max_period, int out_f32, void* stream);
int da_linear_small_m_bf16(const void* x, const void* W, const void* bias, const void* res, void* out, int M, int N,
                           int K, int ldx, int ldo, int ldr, int act_in, int act_out, void* stream);
int da_conv_thin_in_bf16(const void* x, const void* w, const void* bias, void* y, int B, int H, int W, int Cin, int Cout,
                         int ksize, int in_nchw, float in_div, float in_add, void* stream);
int da_conv_thin_out_bf16(const void* x, const void* w, const void* bias, void* y, int B, int H, int W, int Cin,
                          int Cout, int out_f32, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Launch plans: a recorded sequence of the launch entry points above, owned by the library and replayed by ONE call --
 * the C-ABI form of the denoising step the Python pipelines capture into a HIP graph (the reference's per-step work,
 * pipeline_stable_diffusion_xl.py:1186-1250: scale_model_input, the U-Net forward, the CFG combine, scheduler.step; and the
 * decode, :1283-1299), so that a host with no Python (examples/abi_demo.cpp) runs a whole model step.
 *
 * An op names an entry point (DA_FN_*) and carries its arguments in declaration order, the trailing stream excluded, one
 * 64-bit slot each: device and host pointers as addresses, int / long long sign-extended, float as its 32 bits (upper half
 * ignored).  da_plan_create() copies what the call passes by host address -- the da_gemm_params / da_attention_params
 * structs, the two small host arrays of da_rmsnorm_rope_bf16 -- so the caller's copies may go away; DEVICE addresses are kept
 * as given and must stay valid for as long as the plan is launched (the buffers of a step: weights, activations,
 * workspaces).  da_plan_launch() issues the ops in order on `stream` (no synchronisation; results are those of the same calls
 * made one by one: bit-identical) and stops at the first failing op (its index in *failed_op, its code returned).
 * da_plan_relocate() rewrites every device address that falls into [old_base[r], old_base[r] + bytes[r]) to the same offset
 * in new_base[r] -- for a plan recorded by one process and replayed by another; *unmatched counts the non-NULL addresses no
 * region covered (left as they were).  A plan is immutable while launched; launches of one plan from several threads are
 * safe, relocate / destroy are not concurrent with them.
 * ------------------------------------------------------------------------------------------------------------------ */
#define DA_FN_GEMM 1                      /* da_gemm_bf16 */
#define DA_FN_GEMM_PAIR 2                 /* da_gemm_pair_bf16 */
#define DA_FN_ATTENTION 3                 /* da_attention_bf16 */
#define DA_FN_GROUPNORM_NHWC 4            /* da_groupnorm_nhwc_bf16 */
#define DA_FN_RMSNORM 5                   /* da_rmsnorm_bf16 */
#define DA_FN_LAYERNORM 6                 /* da_layernorm_bf16 */
#define DA_FN_RMSNORM_ROPE 7              /* da_rmsnorm_rope_bf16 */
#define DA_FN_SOFTMAX_ROWS 8              /* da_softmax_rows_f32_bf16 */
#define DA_FN_RMSNORM_CHANNELS 9          /* da_rmsnorm_channels_bf16 */
#define DA_FN_EULER_SCALE_MODEL_INPUT 10  /* da_euler_scale_model_input */
#define DA_FN_EULER_STEP 11               /* da_euler_step */
#define DA_FN_X0_LINEAR_STEP 12           /* da_x0_linear_step */
#define DA_FN_FLOWMATCH_STEP 13           /* da_flowmatch_step */
#define DA_FN_UNIPC_FLOW_STEP 14          /* da_unipc_flow_step */
#define DA_FN_ADVANCE_STEP 15             /* da_advance_step */
#define DA_FN_CFG_RESCALE 16              /* da_cfg_rescale */
#define DA_FN_CAST_F32_BF16 17            /* da_cast_f32_bf16 */
#define DA_FN_MUL_SCALAR 18               /* da_mul_scalar */
#define DA_FN_BCAST_ADD_F32 19            /* da_bcast_add_f32 */
#define DA_FN_PATCHIFY3D 20               /* da_patchify3d_bf16 */
#define DA_FN_UNPATCHIFY3D 21             /* da_unpatchify3d_bf16 */
#define DA_FN_TRANSPOSE 22                /* da_transpose_bf16 */
#define DA_FN_NHWC_TAKE_NCHW 23           /* da_nhwc_take_nchw_bf16 */
#define DA_FN_NHWC_TAKE_POSTPROCESS 24    /* da_nhwc_take_postprocess */
#define DA_FN_PERMUTE_0213 25             /* da_permute_0213_bf16 */
#define DA_FN_IMAGE_POSTPROCESS 26        /* da_image_postprocess */
#define DA_FN_FRAMES_TO_NCTHW 27          /* da_frames_to_ncthw_bf16 */
#define DA_FN_TIMESTEP_EMBEDDING 28       /* da_timestep_embedding */
#define DA_FN_LINEAR_SMALL_M 29           /* da_linear_small_m_bf16 */
#define DA_FN_CONV_THIN_IN 30             /* da_conv_thin_in_bf16 */
#define DA_FN_CONV_THIN_OUT 31            /* da_conv_thin_out_bf16 */
#define DA_FN_COUNT 32
#define DA_PLAN_MAX_ARGS 16

typedef struct da_plan_op {
  int fn;       /* DA_FN_* */
  int reserved; /* 0 */
  unsigned long long arg[DA_PLAN_MAX_ARGS];
} da_plan_op;
typedef struct da_plan da_plan;

int da_plan_create(const da_plan_op* ops, int n_ops, da_plan** out);
int da_plan_launch(const da_plan* plan, void* stream, int* failed_op);
int da_plan_relocate(da_plan* plan, int n_regions, const void* const* old_base, const unsigned long long* bytes,
                     void* const* new_base, int* unmatched);
int da_plan_op_count(const da_plan* plan);
void da_plan_destroy(da_plan* plan);
/* Argument layout of entry point `fn` as da_plan_op.arg[] holds it: the number of slots, and one character per slot --
 * p device pointer, i / n int, l long long, f float, G da_gemm_params*, A da_attention_params*, I host int array and Q host
 * array of device pointers (both of length n, the call's `parts`).  -1 / NULL for an unknown fn. */
int da_plan_arg_count(int fn);
const char* da_plan_arg_kinds(int fn);

#ifdef __cplusplus
}
#endif
#endif
