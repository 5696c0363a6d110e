// Launch plans (include/diffusers_amd.h, "launch plans"): a recorded sequence of this library's launch entry points, owned by
// the library and replayed by ONE call from any host language -- the C-ABI counterpart of the HIP graph the Python pipelines
// capture.  Host code only: nothing here touches the device except through the entry points it replays.
//
// A plan deep-copies what the recorded calls passed by address on the host side (the parameter structs, the small host arrays
// of da_rmsnorm_rope_bf16) and keeps device addresses as they were given; da_plan_relocate() moves them when the buffers of the
// replaying process live elsewhere (a plan written to a file by one process and run by another: examples/abi_demo.cpp).
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "common.cuh"

namespace {

// One character per argument of the entry point, in declaration order, the trailing stream excluded:
//   p device pointer   i int   l long long   f float   G da_gemm_params*   A da_attention_params*
//   I host int[parts]   Q host (device pointer)[parts]      -- `parts` = the 'n' argument of the same call
//   n int that also gives the length of the call's host arrays
const char* const kSig[DA_FN_COUNT] = {
    nullptr,
    /* DA_FN_GEMM                     */ "G",
    /* DA_FN_GEMM_PAIR                */ "GG",
    /* DA_FN_ATTENTION                */ "A",
    /* DA_FN_GROUPNORM_NHWC           */ "ppippppiiiifip",
    /* DA_FN_RMSNORM                  */ "pppiiiif",
    /* DA_FN_LAYERNORM                */ "ppppppiiiiiiif",
    /* DA_FN_RMSNORM_ROPE             */ "piiiiinIQfppii",
    /* DA_FN_SOFTMAX_ROWS             */ "ppiill",
    /* DA_FN_RMSNORM_CHANNELS         */ "ppplifi",
    /* DA_FN_EULER_SCALE_MODEL_INPUT  */ "ppppili",
    /* DA_FN_EULER_STEP               */ "pppppiflii",
    /* DA_FN_X0_LINEAR_STEP           */ "ppplpppiflii",
    /* DA_FN_FLOWMATCH_STEP           */ "pppppiflii",
    /* DA_FN_UNIPC_FLOW_STEP          */ "pppppppiflii",
    /* DA_FN_ADVANCE_STEP             */ "p",
    /* DA_FN_CFG_RESCALE              */ "pppilffi",
    /* DA_FN_CAST_F32_BF16            */ "ppil",
    /* DA_FN_MUL_SCALAR               */ "ppfili",
    /* DA_FN_BCAST_ADD_F32            */ "pppii",
    /* DA_FN_PATCHIFY3D               */ "ppiiiiiiii",
    /* DA_FN_UNPATCHIFY3D             */ "ppiiiiiiii",
    /* DA_FN_TRANSPOSE                */ "ppiill",
    /* DA_FN_NHWC_TAKE_NCHW           */ "ppllii",
    /* DA_FN_NHWC_TAKE_POSTPROCESS    */ "ppllii" "i",
    /* DA_FN_PERMUTE_0213             */ "ppliii",
    /* DA_FN_IMAGE_POSTPROCESS        */ "ppiilii",
    /* DA_FN_FRAMES_TO_NCTHW          */ "ppiiliiffi",
    /* DA_FN_TIMESTEP_EMBEDDING       */ "ppppiiifffi",
    /* DA_FN_LINEAR_SMALL_M           */ "pppppiiiiiiii",
    /* DA_FN_CONV_THIN_IN             */ "ppppiiiiiiiff",
    /* DA_FN_CONV_THIN_OUT            */ "ppppiiiiii",
};

struct Op {
  int fn;
  unsigned long long a[DA_PLAN_MAX_ARGS];
};

}  // namespace

struct da_plan {
  std::vector<Op> ops;
  // deep copies; Op::a[] of a G / A / I / Q argument holds the address of its copy (stable: one allocation each)
  std::vector<std::unique_ptr<da_gemm_params>> gemm;
  std::vector<std::unique_ptr<da_attention_params>> attn;
  std::vector<std::unique_ptr<std::vector<int>>> ints;
  std::vector<std::unique_ptr<std::vector<const void*>>> ptrs;
};

namespace {

inline float as_float(unsigned long long v) {
  uint32_t u = (uint32_t)v;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// every device address the plan holds, as a list of slots to visit
template <class F>
void for_each_device_pointer(da_plan* plan, F&& f) {
  for (Op& op : plan->ops) {
    const char* sig = kSig[op.fn];
    for (int i = 0; sig[i]; ++i)
      if (sig[i] == 'p') {
        void* p = (void*)(uintptr_t)op.a[i];
        f(p);
        op.a[i] = (unsigned long long)(uintptr_t)p;
      }
  }
  auto cv = [&](const void*& p) { void* q = const_cast<void*>(p); f(q); p = q; };
  auto cf = [&](const float*& p) { void* q = (void*)p; f(q); p = (const float*)q; };
  for (auto& g : plan->gemm) {
    cv(g->A), cv(g->A2), cv(g->W), f(g->C), cv(g->bias), cv(g->rowvec), cv(g->residual), cv(g->bias_rows), cv(g->gate);
    f(g->workspace), f(g->sync_flags);
    { void* q = g->stats_out; f(q); g->stats_out = (float*)q; }
    cf(g->ln_stats), cf(g->ln_s), cf(g->ln_c), cv(g->prefetch), f(g->vt), cv(g->xa_k), cv(g->xa_vt);
  }
  for (auto& t : plan->attn) cv(t->q), cv(t->k), cv(t->vt), f(t->out), cv(t->bias), f(t->split_ws);
  for (auto& v : plan->ptrs)
    for (const void*& p : *v) cv(p);
}

int run_op(const Op& op, void* s) {
  const unsigned long long* a = op.a;
#define P(i) ((void*)(uintptr_t)a[i])
#define CF(i) ((const float*)(uintptr_t)a[i])
#define CI(i) ((const int*)(uintptr_t)a[i])
#define I(i) ((int)(long long)a[i])
#define LL(i) ((long long)a[i])
#define F(i) (as_float(a[i]))
  switch (op.fn) {
    case DA_FN_GEMM: return da_gemm_bf16((const da_gemm_params*)P(0), s);
    case DA_FN_GEMM_PAIR: return da_gemm_pair_bf16((const da_gemm_params*)P(0), (const da_gemm_params*)P(1), s);
    case DA_FN_ATTENTION: return da_attention_bf16((const da_attention_params*)P(0), s);
    case DA_FN_GROUPNORM_NHWC:
      return da_groupnorm_nhwc_bf16(P(0), P(1), I(2), P(3), P(4), P(5), P(6), I(7), I(8), I(9), I(10), F(11), I(12), P(13), s);
    case DA_FN_RMSNORM: return da_rmsnorm_bf16(P(0), P(1), P(2), I(3), I(4), I(5), I(6), F(7), s);
    case DA_FN_LAYERNORM:
      return da_layernorm_bf16(P(0), P(1), P(2), P(3), P(4), P(5), I(6), I(7), I(8), I(9), I(10), I(11), I(12), F(13), s);
    case DA_FN_RMSNORM_ROPE:
      return da_rmsnorm_rope_bf16(P(0), I(1), I(2), I(3), I(4), I(5), I(6), CI(7), (const void* const*)P(8), F(9), CF(10), CF(11),
                                  I(12), I(13), s);
    case DA_FN_SOFTMAX_ROWS: return da_softmax_rows_f32_bf16(P(0), P(1), I(2), I(3), LL(4), LL(5), s);
    case DA_FN_RMSNORM_CHANNELS: return da_rmsnorm_channels_bf16(P(0), P(1), P(2), LL(3), I(4), F(5), I(6), s);
    case DA_FN_EULER_SCALE_MODEL_INPUT: return da_euler_scale_model_input(P(0), P(1), CF(2), CI(3), I(4), LL(5), I(6), s);
    case DA_FN_EULER_STEP: return da_euler_step(P(0), P(1), P(2), CF(3), CI(4), I(5), F(6), LL(7), I(8), I(9), s);
    case DA_FN_X0_LINEAR_STEP:
      return da_x0_linear_step(P(0), P(1), P(2), LL(3), P(4), CF(5), CI(6), I(7), F(8), LL(9), I(10), I(11), s);
    case DA_FN_FLOWMATCH_STEP: return da_flowmatch_step(P(0), P(1), P(2), CF(3), CI(4), I(5), F(6), LL(7), I(8), I(9), s);
    case DA_FN_UNIPC_FLOW_STEP:
      return da_unipc_flow_step(P(0), P(1), P(2), P(3), P(4), CF(5), CI(6), I(7), F(8), LL(9), I(10), I(11), s);
    case DA_FN_ADVANCE_STEP: return da_advance_step((int*)P(0), s);
    case DA_FN_CFG_RESCALE: return da_cfg_rescale(P(0), P(1), (float*)P(2), I(3), LL(4), F(5), F(6), I(7), s);
    case DA_FN_CAST_F32_BF16: return da_cast_f32_bf16(CF(0), P(1), I(2), LL(3), s);
    case DA_FN_MUL_SCALAR: return da_mul_scalar(P(0), P(1), F(2), I(3), LL(4), I(5), s);
    case DA_FN_BCAST_ADD_F32: return da_bcast_add_f32(CF(0), P(1), (float*)P(2), I(3), I(4), s);
    case DA_FN_PATCHIFY3D: return da_patchify3d_bf16(P(0), P(1), I(2), I(3), I(4), I(5), I(6), I(7), I(8), I(9), s);
    case DA_FN_UNPATCHIFY3D: return da_unpatchify3d_bf16(P(0), P(1), I(2), I(3), I(4), I(5), I(6), I(7), I(8), I(9), s);
    case DA_FN_TRANSPOSE: return da_transpose_bf16(P(0), P(1), I(2), I(3), LL(4), LL(5), s);
    case DA_FN_NHWC_TAKE_NCHW: return da_nhwc_take_nchw_bf16(P(0), P(1), LL(2), LL(3), I(4), I(5), s);
    case DA_FN_NHWC_TAKE_POSTPROCESS: return da_nhwc_take_postprocess(P(0), P(1), LL(2), LL(3), I(4), I(5), I(6), s);
    case DA_FN_PERMUTE_0213: return da_permute_0213_bf16(P(0), P(1), LL(2), I(3), I(4), I(5), s);
    case DA_FN_IMAGE_POSTPROCESS: return da_image_postprocess(P(0), P(1), I(2), I(3), LL(4), I(5), I(6), s);
    case DA_FN_FRAMES_TO_NCTHW:
      return da_frames_to_ncthw_bf16(P(0), P(1), I(2), I(3), LL(4), I(5), I(6), F(7), F(8), I(9), s);
    case DA_FN_TIMESTEP_EMBEDDING:
      return da_timestep_embedding(CF(0), CF(1), CI(2), P(3), I(4), I(5), I(6), F(7), F(8), F(9), I(10), s);
    case DA_FN_LINEAR_SMALL_M:
      return da_linear_small_m_bf16(P(0), P(1), P(2), P(3), P(4), I(5), I(6), I(7), I(8), I(9), I(10), I(11), I(12), s);
    case DA_FN_CONV_THIN_IN:
      return da_conv_thin_in_bf16(P(0), P(1), P(2), P(3), I(4), I(5), I(6), I(7), I(8), I(9), I(10), F(11), F(12), s);
    case DA_FN_CONV_THIN_OUT: return da_conv_thin_out_bf16(P(0), P(1), P(2), P(3), I(4), I(5), I(6), I(7), I(8), I(9), s);
  }
#undef P
#undef CF
#undef CI
#undef I
#undef LL
#undef F
  return DA_ERR_INVALID;
}

}  // namespace

extern "C" int da_plan_arg_count(int fn) {
  if (fn <= 0 || fn >= DA_FN_COUNT) return -1;
  return (int)std::strlen(kSig[fn]);
}

extern "C" const char* da_plan_arg_kinds(int fn) { return (fn <= 0 || fn >= DA_FN_COUNT) ? nullptr : kSig[fn]; }

extern "C" int da_plan_create(const da_plan_op* ops, int n_ops, da_plan** out) {
  if (!out) return DA_ERR_INVALID;
  *out = nullptr;
  if (n_ops < 0 || (n_ops > 0 && !ops)) return DA_ERR_INVALID;
  std::unique_ptr<da_plan> plan(new da_plan);
  plan->ops.reserve(n_ops);
  for (int o = 0; o < n_ops; ++o) {
    const da_plan_op& src = ops[o];
    if (src.fn <= 0 || src.fn >= DA_FN_COUNT) return DA_ERR_INVALID;
    const char* sig = kSig[src.fn];
    Op op;
    op.fn = src.fn;
    std::memset(op.a, 0, sizeof(op.a));
    int parts = 0;
    for (int i = 0; sig[i]; ++i) {
      const unsigned long long v = src.arg[i];
      switch (sig[i]) {
        case 'G': {
          if (!v) return DA_ERR_INVALID;
          plan->gemm.emplace_back(new da_gemm_params(*(const da_gemm_params*)(uintptr_t)v));
          op.a[i] = (unsigned long long)(uintptr_t)plan->gemm.back().get();
          break;
        }
        case 'A': {
          if (!v) return DA_ERR_INVALID;
          plan->attn.emplace_back(new da_attention_params(*(const da_attention_params*)(uintptr_t)v));
          op.a[i] = (unsigned long long)(uintptr_t)plan->attn.back().get();
          break;
        }
        case 'n':
          parts = (int)(long long)v;
          if (parts < 0 || parts > 64) return DA_ERR_INVALID;
          op.a[i] = v;
          break;
        case 'I': {
          if (!v && parts) return DA_ERR_INVALID;
          const int* h = (const int*)(uintptr_t)v;
          plan->ints.emplace_back(new std::vector<int>(h, h + parts));
          op.a[i] = (unsigned long long)(uintptr_t)plan->ints.back()->data();
          break;
        }
        case 'Q': {
          if (!v && parts) return DA_ERR_INVALID;
          const void* const* h = (const void* const*)(uintptr_t)v;
          plan->ptrs.emplace_back(new std::vector<const void*>(h, h + parts));
          op.a[i] = (unsigned long long)(uintptr_t)plan->ptrs.back()->data();
          break;
        }
        default: op.a[i] = v;
      }
    }
    plan->ops.push_back(op);
  }
  *out = plan.release();
  return DA_OK;
}

extern "C" int da_plan_op_count(const da_plan* plan) { return plan ? (int)plan->ops.size() : -1; }

extern "C" int da_plan_launch(const da_plan* plan, void* stream, int* failed_op) {
  if (failed_op) *failed_op = -1;
  if (!plan) return DA_ERR_INVALID;
  const int n = (int)plan->ops.size();
  for (int o = 0; o < n; ++o) {
    const int rc = run_op(plan->ops[o], stream);
    if (rc != DA_OK) {
      if (failed_op) *failed_op = o;
      return rc;
    }
  }
  return DA_OK;
}

extern "C" int da_plan_relocate(da_plan* plan, int n_regions, const void* const* old_base, const unsigned long long* bytes,
                                void* const* new_base, int* unmatched) {
  if (unmatched) *unmatched = 0;
  if (!plan || n_regions < 0 || (n_regions > 0 && (!old_base || !bytes || !new_base))) return DA_ERR_INVALID;
  int miss = 0;
  for_each_device_pointer(plan, [&](void*& p) {
    if (!p) return;
    const uintptr_t v = (uintptr_t)p;
    for (int r = 0; r < n_regions; ++r) {
      const uintptr_t b = (uintptr_t)old_base[r];
      if (v >= b && v - b < bytes[r]) {
        p = (void*)((uintptr_t)new_base[r] + (v - b));
        return;
      }
    }
    ++miss;
  });
  if (unmatched) *unmatched = miss;
  return DA_OK;
}

extern "C" void da_plan_destroy(da_plan* plan) { delete plan; }
