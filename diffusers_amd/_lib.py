"""ctypes binding of the C ABI declared in ``include/diffusers_amd.h``.

There is NO fallback: if ``libdiffusers_amd.so`` is missing or a symbol is absent the import of any compute path raises.
"""
from __future__ import annotations

import ctypes as C
import threading
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "_C" / "libdiffusers_amd.so"

# ---- constants mirrored from include/diffusers_amd.h ----
DA_OK = 0
ERRORS = {1: "DA_ERR_INVALID", 2: "DA_ERR_LAUNCH", 3: "DA_ERR_UNSUPPORTED"}
ACT_NONE, ACT_GEGLU, ACT_GELU_TANH, ACT_SILU, ACT_GELU_ERF, ACT_QUICK_GELU, ACT_GEGLU_TANH = 0, 1, 2, 3, 4, 5, 6
(TILE_AUTO, TILE_128x128, TILE_64x128, TILE_128x64, TILE_64x64, TILE_256x128, TILE_128x256, TILE_256x256,
 TILE_128x128_W8, TILE_K2_128x128, TILE_K2_128x80, TILE_K2_128x160, TILE_K2_80x128, TILE_K2_128x64,
 TILE_K1_128x320, TILE_K1_256x128, TILE_K1_128x256, TILE_K1_256x160, TILE_K1_256x256, TILE_K1_256x320,
 TILE_K3_256x256, TILE_K3_256x320) = range(22)
TILE_NAMES = ("auto", "128x128", "64x128", "128x64", "64x64", "256x128", "128x256", "256x256", "128x128w8",
              "k2:128x128", "k2:128x80", "k2:128x160", "k2:80x128", "k2:128x64",
              "k1:128x320", "k1:256x128", "k1:128x256", "k1:256x160", "k1:256x256", "k1:256x320",
              "k3:256x256", "k3:256x320")   # k3 = csrc/gemm3.hip (eight-phase loop; sums K in the k1 order: bit-identical to the k1 tiles)
FIRST_K2_TILE = TILE_K2_128x128   # tiles >= this run csrc/gemm2_kernel.cuh (2 K-groups x 4 waves, 16x16x32 MFMA)
STAGE_REGISTER, STAGE_LDS_DIRECT, STAGE_LDS_DIRECT3, STAGE_LDS_DIRECT4, STAGE_LDS_DIRECT6, STAGE_LDS_DIRECT8 = range(6)
STAGE_PINGPONG, STAGE_PINGPONG3 = 6, 7   # K2 tiles: 2- / 3-pair ring with the two K-groups half an iteration apart
RING_SLOTS = (2, 2, 3, 4, 6, 8)  # LDS ring depth per staging code
DTYPE_BF16, DTYPE_F32 = 0, 1
SPLITK_FLAGS = 4096          # DA_SPLITK_FLAGS
SPLITK_ERR_SLOT = SPLITK_FLAGS - 1
SPLITK_MAX = 24              # DA_SPLITK_MAX
PRED_EPSILON, PRED_V, PRED_SAMPLE = 0, 1, 2
PRED_TYPES = {"epsilon": PRED_EPSILON, "v_prediction": PRED_V, "sample": PRED_SAMPLE}


class GemmParams(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("A2", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p),
        ("bias", C.c_void_p), ("rowvec", C.c_void_p), ("residual", C.c_void_p),
        ("bias_rows", C.c_void_p), ("gate", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", C.c_int), ("ldw", C.c_int), ("ldc", C.c_int), ("ldr", C.c_int), ("ld_rowvec", C.c_int),
        ("ld_gate", C.c_int),
        ("rows_per_batch", C.c_int),
        ("alpha", C.c_float), ("out_scale", C.c_float),
        ("act", C.c_int), ("out_f32", C.c_int), ("conv", C.c_int),
        ("Hin", C.c_int), ("Win", C.c_int), ("C1", C.c_int), ("C2", C.c_int),
        ("Hout", C.c_int), ("Wout", C.c_int), ("stride", C.c_int), ("up", C.c_int), ("pad", C.c_int),
        ("tile", C.c_int), ("staging", C.c_int), ("gate_f32", C.c_int),
        ("split_k", C.c_int), ("workspace", C.c_void_p), ("sync_flags", C.c_void_p), ("workspace_bytes", C.c_longlong),
        ("stats_out", C.c_void_p), ("stats_ld", C.c_int), ("ln_stats", C.c_void_p), ("ln_stats_ld", C.c_int),
        ("ln_parts", C.c_int), ("ln_s", C.c_void_p), ("ln_c", C.c_void_p), ("ln_eps", C.c_float), ("k_valid", C.c_int),
        ("prefetch", C.c_void_p), ("prefetch_bytes", C.c_longlong),
        ("vt", C.c_void_p), ("vt_col0", C.c_int), ("ld_vt", C.c_longlong),
        ("xa_k", C.c_void_p), ("xa_vt", C.c_void_p), ("xa_skv", C.c_int), ("xa_skv_alloc", C.c_int), ("xa_k_ld", C.c_int),
        ("xa_vt_ld", C.c_longlong), ("xa_scale", C.c_float),
    ]


class AttentionParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("vt", C.c_void_p), ("out", C.c_void_p),
        ("B", C.c_int), ("H", C.c_int), ("Sq", C.c_int), ("Skv", C.c_int), ("Skv_alloc", C.c_int), ("D", C.c_int),
        ("q_batch_stride", C.c_longlong), ("k_batch_stride", C.c_longlong),
        ("vt_batch_stride", C.c_longlong), ("o_batch_stride", C.c_longlong),
        ("q_row_stride", C.c_int), ("k_row_stride", C.c_int), ("vt_ld", C.c_int), ("o_row_stride", C.c_int),
        ("scale", C.c_float), ("ring_slots", C.c_int),
        ("bias", C.c_void_p), ("bias_batch_stride", C.c_longlong), ("bias_head_stride", C.c_longlong),
        ("bias_row_stride", C.c_int), ("bias_f32", C.c_int), ("causal", C.c_int), ("q_block", C.c_int), ("pv_delay", C.c_int),
        ("algo", C.c_int),
        ("split_ws", C.c_void_p), ("split_ws_bytes", C.c_longlong), ("kv_split", C.c_int),
    ]


ATTN_SPLIT_COUNTER_BYTES = 16384      # DA_ATTN_SPLIT_COUNTER_BYTES


# ---- launch plans (include/diffusers_amd.h "launch plans"; csrc/plan.hip) ----
PLAN_MAX_ARGS = 16
FN_IDS = {name: i + 1 for i, name in enumerate((
    "da_gemm_bf16", "da_gemm_pair_bf16", "da_attention_bf16", "da_groupnorm_nhwc_bf16", "da_rmsnorm_bf16", "da_layernorm_bf16",
    "da_rmsnorm_rope_bf16", "da_softmax_rows_f32_bf16", "da_rmsnorm_channels_bf16", "da_euler_scale_model_input", "da_euler_step",
    "da_x0_linear_step", "da_flowmatch_step", "da_unipc_flow_step", "da_advance_step", "da_cfg_rescale", "da_cast_f32_bf16",
    "da_mul_scalar", "da_bcast_add_f32", "da_patchify3d_bf16", "da_unpatchify3d_bf16", "da_transpose_bf16",
    "da_nhwc_take_nchw_bf16", "da_nhwc_take_postprocess", "da_permute_0213_bf16", "da_image_postprocess",
    "da_frames_to_ncthw_bf16", "da_timestep_embedding", "da_linear_small_m_bf16", "da_conv_thin_in_bf16",
    "da_conv_thin_out_bf16"))}
FN_COUNT = len(FN_IDS) + 1


class PlanOp(C.Structure):
    _fields_ = [("fn", C.c_int), ("reserved", C.c_int), ("arg", C.c_ulonglong * PLAN_MAX_ARGS)]


_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "da_version": (_i, []),
    "da_sizeof_gemm_params": (C.c_size_t, []),
    "da_sizeof_attention_params": (C.c_size_t, []),
    "da_last_error": (C.c_char_p, []),
    "da_set_launch_events": (_i, [_vp, _vp]),
    "da_set_launch_flags": (_i, [C.c_uint]),
    "da_mfma_probe": (_i, [_i, _i, _vp, C.POINTER(C.c_double), _vp]),
    "da_gemm_bf16": (_i, [C.POINTER(GemmParams), _vp]),
    "da_gemm_pair_bf16": (_i, [C.POINTER(GemmParams), C.POINTER(GemmParams), _vp]),
    "da_gemm_stats_parts": (_i, [C.POINTER(GemmParams)]),
    "da_gemm_tune": (_i, [C.POINTER(GemmParams), C.POINTER(GemmParams), _vp, _i, _vp, C.c_size_t, C.POINTER(C.c_int),
                          C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "da_attention_bf16": (_i, [C.POINTER(AttentionParams), _vp]),
    "da_attention_split_plan": (_ll, [C.POINTER(AttentionParams), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "da_groupnorm_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "da_groupnorm_sync_bytes": (C.c_size_t, []),
    "da_groupnorm_nhwc_bf16": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "da_rmsnorm_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "da_layernorm_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "da_rmsnorm_rope_bf16": (_i, [_vp, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_void_p), _f, _vp, _vp,
                                  _i, _i, _vp]),
    "da_softmax_rows_f32_bf16": (_i, [_vp, _vp, _i, _i, _ll, _ll, _vp]),
    "da_euler_scale_model_input": (_i, [_vp, _vp, _vp, _vp, _i, _ll, _i, _vp]),
    "da_euler_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _ll, _i, _i, _vp]),
    "da_x0_linear_step": (_i, [_vp, _vp, _vp, _ll, _vp, _vp, _vp, _i, _f, _ll, _i, _i, _vp]),
    "da_flowmatch_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _ll, _i, _i, _vp]),
    "da_unipc_flow_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _ll, _i, _i, _vp]),
    "da_advance_step": (_i, [_vp, _vp]),
    "da_cfg_rescale": (_i, [_vp, _vp, _vp, _i, _ll, _f, _f, _i, _vp]),
    "da_cast_f32_bf16": (_i, [_vp, _vp, _i, _ll, _vp]),
    "da_mul_scalar": (_i, [_vp, _vp, _f, _i, _ll, _i, _vp]),
    "da_bcast_add_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "da_patchify3d_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "da_unpatchify3d_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "da_transpose_bf16": (_i, [_vp, _vp, _i, _i, _ll, _ll, _vp]),
    "da_nhwc_take_nchw_bf16": (_i, [_vp, _vp, _ll, _ll, _i, _i, _vp]),
    "da_nhwc_take_postprocess": (_i, [_vp, _vp, _ll, _ll, _i, _i, _i, _vp]),
    "da_image_postprocess": (_i, [_vp, _vp, _i, _i, _ll, _i, _i, _vp]),
    "da_permute_0213_bf16": (_i, [_vp, _vp, _ll, _i, _i, _i, _vp]),
    "da_frames_to_ncthw_bf16": (_i, [_vp, _vp, _i, _i, _ll, _i, _i, _f, _f, _i, _vp]),
    "da_rmsnorm_channels_bf16": (_i, [_vp, _vp, _vp, _ll, _i, _f, _i, _vp]),
    "da_timestep_embedding": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _i, _vp]),
    "da_linear_small_m_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "da_conv_thin_in_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _vp]),
    "da_conv_thin_out_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "da_plan_create": (_i, [C.POINTER(PlanOp), _i, C.POINTER(C.c_void_p)]),
    "da_plan_launch": (_i, [_vp, _vp, C.POINTER(C.c_int)]),
    "da_plan_relocate": (_i, [_vp, _i, C.POINTER(C.c_void_p), C.POINTER(C.c_ulonglong), C.POINTER(C.c_void_p),
                              C.POINTER(C.c_int)]),
    "da_plan_op_count": (_i, [_vp]),
    "da_plan_destroy": (None, [_vp]),
    "da_plan_arg_count": (_i, [_i]),
    "da_plan_arg_kinds": (C.c_char_p, [_i]),
}

ABI_VERSION = 6              # include/diffusers_amd.h DA_ABI_VERSION
_lib = None
_tls = threading.local()     # .recorder: the plan recorder of this thread (diffusers_amd/plan.py), if one is active


def load() -> C.CDLL:
    """Load libdiffusers_amd.so (building is a separate, explicit step: ``python -m diffusers_amd.build``)."""
    global _lib
    rec = getattr(_tls, "recorder", None)
    if _lib is not None:
        return _lib if rec is None else rec.proxy
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m diffusers_amd.build` "
            "(or __graft_entry__.build()). diffusers_amd has no CPU / PyTorch fallback path."
        )
    # PyTorch-ROCm ships its own libamdhip64.so (SONAME libamdhip64.so.7) and must be the HIP runtime of the process:
    # device pointers and streams handed to the C ABI are its objects.  Importing torch first makes the dynamic loader
    # resolve this library's libamdhip64.so.7 dependency to the runtime torch already mapped (loading ours first would
    # map /opt/rocm's copy as a second, device-less runtime -> hipErrorNoDevice on every launch).
    import torch  # noqa: F401
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud failure
        fn.restype = res
        fn.argtypes = args
    # ABI handshake (include/diffusers_amd.h DA_ABI_VERSION): the parameter structs carry no size field, so a library built from
    # another header revision than these ctypes mirrors must be refused before the first launch reads a struct
    got = (lib.da_version(), lib.da_sizeof_gemm_params(), lib.da_sizeof_attention_params())
    want = (ABI_VERSION, C.sizeof(GemmParams), C.sizeof(AttentionParams))
    if got != want:
        raise RuntimeError(f"{LIB_PATH} has ABI (version, sizeof da_gemm_params, sizeof da_attention_params) = {got}, this package "
                           f"expects {want}: rebuild the extension (`python -m diffusers_amd.build`)")
    _lib = lib
    return lib if rec is None else rec.proxy


def check(status: int, what: str) -> None:
    if status != DA_OK:
        detail = ""
        if status == 2 and _lib is not None:
            detail = f" ({_lib.da_last_error().decode()})"
        raise RuntimeError(f"{what} failed: {ERRORS.get(status, status)}{detail}")
