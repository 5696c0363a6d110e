"""Thin torch-tensor front end of the C ABI: pointer extraction, shape checks, output allocation.

torch is used here only for device memory and the current HIP stream; every arithmetic op below is a hand-written
gfx950 kernel in ``csrc/``.  Nothing in this module has a CPU or PyTorch fallback: tensors must live on a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref
from typing import Optional

import torch

from . import _lib as L
from . import tuning

bf16 = torch.bfloat16

# Global knobs (tests / bench flip these to A/B the staging paths).
DEFAULT_STAGING = L.STAGE_LDS_DIRECT
DEFAULT_TILE = L.TILE_AUTO
TUNING = True  # per-shape (tile, staging) from diffusers_amd.tuning when the caller does not pin them

# ---- weight prefetch hints (da_gemm_params.prefetch) -------------------------------------------------------------------------
# Inside a denoising step every weight is met cold: a step streams 5 GB of weights through a 256 MB memory-side cache.  A GEMM
# launch can read the NEXT launch's weight behind its own K loop (the second kernel family does; measured 7-16 % per launch on
# SDXL's projections, tools/bench_prefetch.py).  The ops layer does not know what comes next, so the pipelines run ONE step
# under `weight_prefetch(pf, "record")` -- every implicit-GEMM launch notes its weight tensor, in issue order -- and every later
# step (the one captured into the HIP graph included) under `weight_prefetch(pf, "apply")`: launch i is handed the weight of
# launch i + 1, the last one that of launch 0 (the next step).  The weight of a launch is whichever operand the MODEL owns.  A sequence that does not reproduce the recorded one switches the
# hints off for the rest of that step (they are a speed hint: results never depend on them).
PREFETCH = os.environ.get("DIFFUSERS_AMD_PREFETCH", "1") != "0"
_prefetch_tls = threading.local()     # the active trace of THIS thread (two pipelines may step on two threads)


class WeightPrefetch:
    def __init__(self, models=(), owner=None, slots=()):
        """`models`: whose tensors count as weights -- either given directly, or looked up as attributes `slots` of `owner` at
        every refresh() (a pipeline whose `unet` is re-assigned later must neither keep the old model alive nor hint at it)."""
        self._models = tuple(m for m in models if m is not None)
        self._owner = weakref.ref(owner) if owner is not None else None
        self._slots = tuple(slots)
        self.ptrs = set()      # never an activation address: a hint captured into a HIP graph must stay valid for ever
        self.seq = []          # per implicit-GEMM launch of one step, in issue order: (ptr, bytes) of its weight, or None
        self.mode = None
        self.idx = 0
        self.ok = True
        self.applied = 0       # hints handed out by the last "apply" pass (diagnostics)

    @property
    def models(self):
        owner = self._owner() if self._owner is not None else None
        found = tuple(m for m in (getattr(owner, n, None) for n in self._slots) if m is not None) if owner is not None else ()
        return self._models + found

    def refresh(self):
        from .packed_cache import packed_tensors
        self.ptrs = {t.data_ptr() for m in self.models for t in packed_tensors(m).values() if t.is_cuda}


class weight_prefetch:
    """Context manager around ONE step: `mode` = "record" or "apply" (see above)."""

    def __init__(self, pf: Optional[WeightPrefetch], mode: str):
        self.pf, self.mode = pf, mode

    def __enter__(self):
        self.prev = getattr(_prefetch_tls, "pf", None)
        if self.pf is not None and PREFETCH:
            pf = self.pf
            pf.mode, pf.idx, pf.applied = self.mode, 0, 0
            pf.ok = True                                           # a mismatch switches the hints off for ONE step, not for ever
            if self.mode == "record":
                pf.seq = []
                pf.refresh()
            _prefetch_tls.pf = pf
        return self.pf

    def __exit__(self, *exc):
        pf = self.pf
        if pf is not None and PREFETCH and getattr(_prefetch_tls, "pf", None) is pf and pf.mode == "apply" and pf.idx != len(pf.seq):
            pf.ok = False                                          # not the step that was recorded
        _prefetch_tls.pf = self.prev
        return False


def _nbytes16(t: torch.Tensor) -> int:
    return (t.numel() * t.element_size()) & ~15


# A/B knob: bytes of the next weight a launch may pull (0 = all the kernel's per-wave budget covers; 16 MiB = the round-3 reach)
PREFETCH_CAP_BYTES = int(float(os.environ.get("DIFFUSERS_AMD_PREFETCH_CAP_MB", "0")) * (1 << 20)) & ~1023


def _prefetch_hook(p: "L.GemmParams", x: torch.Tensor, w: torch.Tensor) -> None:
    pf = getattr(_prefetch_tls, "pf", None)
    if pf is None:
        return
    # the weight is whichever operand the model owns (the swapped V^T projections pass it as `x`)
    mine = (w.data_ptr(), _nbytes16(w)) if w.data_ptr() in pf.ptrs else (x.data_ptr(), _nbytes16(x)) if x.data_ptr() in pf.ptrs else None
    if pf.mode == "record":
        pf.seq.append(mine)
        return
    i = pf.idx
    pf.idx += 1
    if not pf.ok or i >= len(pf.seq) or pf.seq[i] != mine:
        if pf.ok and os.environ.get("DIFFUSERS_AMD_PREFETCH_DEBUG"):
            print(f"[prefetch] launch {i}: operands {tuple(x.shape)} / {tuple(w.shape)} are not the recorded step's", flush=True)
        pf.ok = False                                              # not the step that was recorded: no more hints this step
        return
    nxt = pf.seq[(i + 1) % len(pf.seq)]
    if nxt is not None:
        p.prefetch, p.prefetch_bytes = nxt[0], (min(nxt[1], PREFETCH_CAP_BYTES) if PREFETCH_CAP_BYTES else nxt[1])
        pf.applied += 1


SPLITK_WS_BYTES = 64 << 20    # split-K workspace per (device, stream): 256 fp32 partial tiles of 256x256
_splitk_ws = {}


def splitk_workspace(device: torch.device, stream: int):
    """(workspace, flags) of the in-launch split-K reduction (da_gemm_params.workspace / .sync_flags): one pair per
    (device, stream) -- launches on one stream are ordered, so they can share it; the flags are zeroed ONCE here and
    re-armed by the kernel.  Allocated on first use and kept (a HIP graph may have captured the addresses)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream)
    ent = _splitk_ws.get(key)
    if ent is None:
        ent = (torch.empty(SPLITK_WS_BYTES, dtype=torch.uint8, device=device),
               torch.zeros(L.SPLITK_FLAGS, dtype=torch.int32, device=device))
        _splitk_ws[key] = ent
    return ent


# Key-split tail of the flash kernel (include/diffusers_amd.h da_attention_params.split_ws, csrc/attention2.hip): a launch whose
# query blocks do not fill the chip's CUs in whole rounds (SDXL S = 1024: 320 blocks on 256 CUs) splits the keys of its LAST
# partial round over several workgroups.  DIFFUSERS_AMD_ATTN_SPLIT=0 turns it off (every block whole: the round-5 launches).
ATTN_SPLIT = os.environ.get("DIFFUSERS_AMD_ATTN_SPLIT", "1") == "1"
ATTN_SPLIT_WS_BYTES = 128 << 20
_attn_split_ws = {}


def attn_split_workspace(device: torch.device, stream: int) -> torch.Tensor:
    """The flash kernel's key-split workspace, one per (device, stream) like the split-K one: ticket counters (zeroed ONCE here,
    re-armed by the kernel) followed by scratch for the (O, m, l) partials.  Allocated on first use and kept (a HIP graph may have
    captured its address); pipelines allocate the capture stream's before they capture."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream)
    ws = _attn_split_ws.get(key)
    if ws is None:
        ws = torch.empty(ATTN_SPLIT_WS_BYTES, dtype=torch.uint8, device=device)
        ws[:L.ATTN_SPLIT_COUNTER_BYTES].zero_()
        _attn_split_ws[key] = ws
    return ws


def splitk_error(device=None) -> bool:
    """True if any split-K reducer ever gave up waiting for a producer on this device (diagnostics / tests)."""
    return any(bool(f[L.SPLITK_ERR_SLOT].item()) for (d, _), (_, f) in _splitk_ws.items()
               if device is None or d == torch.device(device).index)


# BasicTransformerBlock can fold norm2 / norm3 into the GEMMs either side of them (linear(stats_out=) / linear(ln=)).  OFF by
# default: measured on an MI355X (profiles/r02d_layernorm_fold.md) the folded consumers cost +4.5 us (to_q) and +15 us (GEGLU
# projection) against the 8.6 us LayerNorm launch + ~1.5 us boundary they remove -- a wash for norm2, a loss for norm3.
# Round 4: the second kernel family carries the fold.  Modes: 0 = off; 2 = norm2 only (attn1.to_out writes the statistics, attn2.to_q
# applies them; measured: to_out > LayerNorm > to_q 42.7 -> 32.5 us per chain at the 1280 level, 40.7 -> 32.7 at 640,
# profiles/r04b_layernorm_fold_k2.jsonl); 1 (or True) = norm2 and norm3 (the GEGLU projection as the second consumer).
# Round 6: 3 (default) = norm2 everywhere and norm3 WHERE THE FOLDED GEGLU LAUNCH RUNS ON ITS OWN EIGHT-PHASE TILE (k3:256x320 carries the
# consumer side since this round: geglu_fold_on_k3 below).  Same box, SDXL image: mode 2 0.9704 / 0.9716, mode 3 0.9836 / 0.9850
# (+ 1.4 %: 60 of the 70 LayerNorm launches left in a step disappear), mode 1 with the folded launch on k1:128x320 0.9332; parity
# unchanged (46.0 dB vs the reference in fp32); SD1.5, whose 32 x 32 / 16 x 16 levels run other tiles, loses 1.2 % under mode 1 and is
# what the per-shape rule is for (profiles/r06d_layernorm_fold_k3.txt).
LN_FOLD = int(os.environ.get("DIFFUSERS_AMD_LN_FOLD", "3"))
LN_FOLD_NORM1 = os.environ.get("DIFFUSERS_AMD_LN_FOLD_NORM1", "1") == "1"   # with LN_FOLD: norm1 too, through the fused Q | K | V projection
LN_FOLD_K2 = os.environ.get("DIFFUSERS_AMD_LN_FOLD_K2", "1") == "1"   # folded launches may use the second kernel family (round 4)
LN_FOLD_K3 = os.environ.get("DIFFUSERS_AMD_LN_FOLD_K3", "1") == "1"   # a folded GEGLU projection may use its eight-phase tile (round 6)
STATS_MAX_PARTS = 64          # DA_LN_MAX_PARTS: slots per row of a statistics buffer
STATS_MAX_CONSUMED = 24       # 4 * DA_LN_PAIR_LOADS: partials per row a consumer launch reads


class RowStats:
    """Per-row (sum, sum of squares) partials of a GEMM output, written by the producing launch (``linear(stats_out=)``)
    and consumed by the GEMM that reads LayerNorm of that output (``linear(ln=)``): fp32 [M][STATS_MAX_PARTS][2], of which
    the first ``parts`` pairs of every row are valid."""

    __slots__ = ("buf", "parts")

    def __init__(self, rows: int, device):
        self.buf = torch.empty((rows, STATS_MAX_PARTS, 2), device=device, dtype=torch.float32)
        self.parts = 0


def geglu_fold_on_k3(rows: int, n_packed: int, k: int) -> bool:
    """Does a LayerNorm-folded GEGLU projection of this shape run on the projection's own eight-phase tile?  (The shipped / live
    table sends the UNFOLDED problem to k3:256x320, whole tiles; linear(ln=) then keeps that tile.)"""
    if not (LN_FOLD_K3 and TUNING) or rows % 256 or n_packed % 320:
        return False
    ent = tuning.table().get(f"lin:M{tuning._m_key(int(rows))}:N{int(n_packed)}:K{int(k)}:a{L.ACT_GEGLU}:f0:r0")
    return ent is not None and ent[0] == L.TILE_K3_256x320


class LNFold:
    """What the consumer GEMM of a folded LayerNorm needs besides the pre-scaled weight: s[n] = sum_k (gamma o W)[n, k]
    and c[n] = sum_k beta[k] W[n, k] (fp32), and eps.  Built once at load time by :func:`fold_layernorm`."""

    def __init__(self, s, c, eps):
        self.s, self.c, self.eps = s, c, eps


def fold_layernorm(w: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float):
    """(W', LNFold) for LN(x; gamma, beta) @ W^T == rstd * (x @ W'^T - mu * s) + c   (W' = bf16(gamma o W))."""
    wf = w.float()
    wp = (wf * gamma.float()[None, :]).to(bf16) if gamma is not None else w
    s = wp.float().sum(dim=1).contiguous()
    c = (wf @ beta.float()).contiguous() if beta is not None else torch.zeros_like(s)
    return wp.contiguous(), LNFold(s, c, float(eps))


def _select_variant(p: "L.GemmParams", tile: Optional[int], staging: Optional[int], stream: int,
                    inplace: bool = False, split_k: Optional[int] = None, device=None) -> None:
    p.split_k = 1
    auto = (TUNING and tile is None and staging is None and split_k is None and DEFAULT_TILE == L.TILE_AUTO
            and DEFAULT_STAGING == L.STAGE_LDS_DIRECT)
    ent = tuning.table().get(tuning.key_of(p)) if auto else None
    # the split-K workspace rides along when the caller asks for a split, when the live tuner may try one, and when the shipped
    # table holds one for this shape (the small-M, deep-K convs of the SD1.5 / DDPM U-Nets)
    want_ws = device is not None and ((auto and (tuning.SPLIT_K or (ent is not None and ent[3] > 1))) or (split_k or 1) > 1)
    if want_ws and not inplace:
        ws, flags = splitk_workspace(device, stream)
        p.workspace, p.sync_flags, p.workspace_bytes = ws.data_ptr(), flags.data_ptr(), ws.numel()
    p._auto = auto
    if auto:
        p.tile, p.staging, sk = tuning.lookup(p, stream, inplace=inplace)
        p.split_k = sk if p.workspace else 1
    else:
        p.tile = DEFAULT_TILE if tile is None else tile
        p.staging = DEFAULT_STAGING if staging is None else staging
        p.split_k = split_k or 1


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


DA_ERR_UNSUPPORTED = 3


def _launch_gemm(p: "L.GemmParams", st: int, what: str) -> None:
    """da_gemm_bf16 with the selected variant.  The per-shape table is keyed by shape only, but the second kernel family
    (tiles >= FIRST_K2_TILE) also looks at operand alignment / strides and the 31-bit offset budget of its staging: when a
    TABLE-selected variant is refused for such a property the launch is retried with the library's own choice from the first
    family (TILE_AUTO), which serves every operand layout.  A variant the CALLER pinned is never replaced."""
    rc = L.load().da_gemm_bf16(C.byref(p), st)
    if rc == DA_ERR_UNSUPPORTED and getattr(p, "_auto", False) and p.tile != L.TILE_AUTO:
        p.tile, p.staging, p.split_k = L.TILE_AUTO, L.STAGE_LDS_DIRECT, 1
        rc = L.load().da_gemm_bf16(C.byref(p), st)
    L.check(rc, what)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, name: str, dtype=bf16) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: diffusers_amd ops need a HIP device tensor (got {t.device}); there is no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


def require_hip(t: torch.Tensor, name: str, dtypes=(bf16,)) -> None:
    """Entry check of every model forward: the engine computes on HIP device tensors only."""
    if t.dtype not in dtypes or not t.is_cuda:
        kinds = " / ".join(str(d).replace("torch.", "").replace("bfloat16", "bf16").replace("float32", "fp32") for d in dtypes)
        raise ValueError(f"{name} must be a {kinds} HIP tensor (there is no CPU / fp32 fallback)")


def _rows2d(t: torch.Tensor, name: str) -> int:
    """Leading dimension (elements) of a 2-D row-major view whose last dim is contiguous."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: expected a 2-D tensor with contiguous last dim, got shape {tuple(t.shape)} "
                         f"strides {t.stride()}")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


# ----------------------------------------------------------------------------------------------------------------------
# GEMM / conv
# ----------------------------------------------------------------------------------------------------------------------
QKV_CANDIDATES = ((L.TILE_K2_128x80, L.STAGE_PINGPONG), (L.TILE_K2_128x80, L.STAGE_PINGPONG3), (L.TILE_K2_128x160, L.STAGE_PINGPONG),
                  (L.TILE_K2_128x160, L.STAGE_LDS_DIRECT), (L.TILE_K1_128x256, L.STAGE_LDS_DIRECT), (L.TILE_K1_128x256, L.STAGE_LDS_DIRECT3),
                  (L.TILE_K1_256x128, L.STAGE_LDS_DIRECT), (L.TILE_K1_256x128, L.STAGE_LDS_DIRECT3))
QKV_RETUNE = os.environ.get("DIFFUSERS_AMD_QKV_RETUNE", "0") == "1"   # measure again what the shipped table pins (tools/gpu_r4.sh qkv)
_qkv_retuned = set()
QKV_TILE_COLS = {L.TILE_K2_128x80: 80, L.TILE_K2_128x160: 160, L.TILE_K1_128x256: 256, L.TILE_K1_256x128: 128}


def qkv_variant(p: "L.GemmParams", stream: int):
    """(tile, staging) of a fused Q | K | V projection (da_gemm_params.vt): the per-shape table under the key ``qkv:<shape>``,
    else measured now among the variants that carry the transposed column block (outside graph capture; HIP events, three
    launches each), else the first candidate whose columns divide the transposed block's origin."""
    key = "qkv:" + tuning.key_of(p)
    ent = tuning.table().get(key) if TUNING else None
    if ent is not None and not (QKV_RETUNE and key not in _qkv_retuned):
        return ent[0], ent[1]
    _qkv_retuned.add(key)
    # The one-K-group tiles (k1:*) add the K slices in another order than the two-K-group tiles: a live, timing-dependent choice
    # stays inside the k2 tiles (every variant bit-identical) unless DIFFUSERS_AMD_GEMM_FAMILY=all asked for the whole field
    # (tools/gpu_r4.sh qkv: that is how the shipped table's entries were measured).
    field = QKV_CANDIDATES if tuning.FAMILY == "all" else tuple(c for c in QKV_CANDIDATES if c[0] < L.TILE_K1_128x320)
    ok = [(t, s) for t, s in field if p.vt_col0 % QKV_TILE_COLS[t] == 0]
    if not ok:
        raise ValueError(f"linear(vt_out=): column origin {p.vt_col0} is not a multiple of a tile width {sorted(set(QKV_TILE_COLS.values()))}")
    if not (TUNING and tuning.LIVE) or torch.cuda.is_current_stream_capturing():
        return ok[0]
    lib, best = L.load(), None
    for t, s in ok:
        p.tile, p.staging, p.split_k = t, s, 1
        if lib.da_gemm_bf16(C.byref(p), stream) != 0:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            lib.da_gemm_bf16(C.byref(p), stream)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 3
        if best is None or us < best[2]:
            best = (t, s, us)
    if best is None:
        raise RuntimeError("linear(vt_out=): no kernel variant accepted this problem")
    tuning.table()[key] = (best[0], best[1], best[2], 1)
    tuning.mark_dirty()
    return best[0], best[1]


# cross-attention in the epilogue of attn2.to_q (round 5).  Opt-in: measured + 7 % on the link in a chain of launches at SDXL's 1280
# level (20.7 -> 19.2 us per layer), - 12 % at the 640 level, and nothing on the image (profiles/r05b_*): the 128 x 128 tile it needs
# covers 160 of the 256 CUs where the 128 x 80 tile of the plain to_q covers all of them
XATTN = os.environ.get("DIFFUSERS_AMD_XATTN", "0") == "1"
# Cross-attention: attn2.to_q pulls the layer's step-invariant K / V^T (one buffer, layers.CrossKV.buf) towards the memory-side cache
# behind its K loop instead of the next GEMM's weight -- the 77-key attention launch that follows otherwise meets them cold
# (5 GB of weights went through the 256 MB cache since the last step touched them).  A/B knob, round 5: measured - 0.5 % on the image on
# two boxes (profiles/r05j_kv_prefetch_same_box_ab.txt: the hint displaces the to_out weight's prefetch and buys less than that): off.
KV_PREFETCH = os.environ.get("DIFFUSERS_AMD_KV_PREFETCH", "0") == "1"
XATTN_MAX_KEYS = 80                                              # da_gemm_params.xa_skv_alloc limit of the one instantiation
XATTN_STAGING = L.STAGE_PINGPONG if os.environ.get("DIFFUSERS_AMD_XATTN_STAGE", "pp") == "pp" else L.STAGE_LDS_DIRECT
XATTN_MIN_TILES = int(os.environ.get("DIFFUSERS_AMD_XATTN_MIN_TILES", "0"))


def xattn_eligible(M: int, N: int, seq: int, head_dim: int, skv_alloc: int) -> bool:
    """The shapes the cross-attention epilogue (da_gemm_params.xa_*) covers: heads of 64 channels, two per 128-column tile; a
    128-row tile never straddles two batches; at most 80 (padded) text tokens."""
    return head_dim == 64 and N % 128 == 0 and seq > 0 and seq % 128 == 0 and M % seq == 0 and M > 8 and \
        0 < skv_alloc <= XATTN_MAX_KEYS and skv_alloc % 8 == 0


def linear_qkv(x: torch.Tensor, wqkv: torch.Tensor, col0: int, bias: Optional[torch.Tensor] = None, ln: Optional[tuple] = None):
    """Fused to_q | to_k | to_v (attention_processor.py:2743-2751) in ONE GEMM: returns (qk [M][col0], vt [N - col0][M]) -- the V
    columns leave the epilogue transposed, the layout the flash kernel consumes.  ``ln``: LayerNorm fold (see :func:`linear`)."""
    M = x.shape[0]
    vt = torch.empty((wqkv.shape[0] - col0, M), device=x.device, dtype=bf16)
    qk = linear(x, wqkv, bias, ln=ln, vt_out=(vt, col0))
    return qk, vt


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, act: int = L.ACT_NONE,
           residual: Optional[torch.Tensor] = None, rowvec: Optional[torch.Tensor] = None, rows_per_batch: int = 0,
           alpha: float = 1.0, out_scale: float = 1.0, out: Optional[torch.Tensor] = None, out_f32: bool = False,
           bias_rows: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None,
           tile: Optional[int] = None, staging: Optional[int] = None, split_k: Optional[int] = None,
           stats_out: Optional[RowStats] = None, ln: Optional[tuple] = None, k_valid: int = 0,
           prefetch: Optional[torch.Tensor] = None, vt_out: Optional[tuple] = None,
           xattn: Optional[dict] = None) -> torch.Tensor:
    """out[M][N] = epilogue(alpha * x[M][K] @ w[N][K]^T).  For act == GEGLU, w/bias are in the packed layout of
    :func:`pack_geglu` and the output has N/2 columns.  ``k_valid`` > 0: columns k >= k_valid of BOTH operands are
    zero padding (da_gemm_params.k_valid): the kernel skips the MFMA steps that would multiply them.

    LayerNorm fold: ``stats_out`` (a :class:`RowStats`) makes this launch also write the row statistics of its output;
    ``ln=(RowStats, LNFold)`` makes it compute LN(x) @ w^T from the UN-normalised ``x`` whose statistics another launch
    wrote (``w`` pre-scaled by :func:`fold_layernorm`).

    ``xattn=dict(k=, vt=, skv=, skv_alloc=, seq=, scale=)`` (da_gemm_params.xa_*): this launch is a cross-attention layer's to_q
    AND its attention -- the result is softmax(scale * q k^T) v for K / V^T of the text embeddings (layers.CrossKV), see
    :func:`xattn_eligible`."""
    p, st = _linear_params(x, w, bias, act=act, residual=residual, rowvec=rowvec, rows_per_batch=rows_per_batch,
                           alpha=alpha, out_scale=out_scale, out=out, out_f32=out_f32, bias_rows=bias_rows, gate=gate,
                           tile=tile, staging=staging, split_k=split_k, stats_out=stats_out, ln=ln, k_valid=k_valid,
                           prefetch=prefetch, vt_out=vt_out, xattn=xattn)
    if p is None:
        return st     # the skinny-M path ran
    _launch_gemm(p, st, "da_gemm_bf16(linear)")
    return p._out


def linear_pair(a: dict, b: dict):
    """Two independent nn.Linear problems in ONE launch (da_gemm_pair_bf16): ``a`` / ``b`` are keyword dicts of
    :func:`linear` (x, w, bias, ...).  The paired launch runs the first kernel family (csrc/gemm_kernel.cuh): bit-identical to
    two :func:`linear` calls pinned to a first-family tile, and within one bf16 ulp of the rounded GEMM term of launches the
    table sends to the second family (a different fp32 summation order of K).  Used where neither problem fills the
    256 CUs on its own (the Q|K and V^T projections of a self-attention layer).  Falls back to two launches when the
    paired variant is unknown and cannot be tuned now, or is not faster than the two separate launches."""
    if a["x"].shape[0] <= 8 or b["x"].shape[0] <= 8:
        # a skinny problem (a 1x1 / 2x2-token mid block) belongs to the GEMV-style kernel: two ordinary launches, decided
        # BEFORE any parameter block is built (building one would already run the skinny kernel)
        return linear(**a), linear(**b)
    pa, st = _linear_params(**a)
    pb, _ = _linear_params(**b)
    if pa is None or pb is None:
        raise ValueError("linear_pair: both problems must take the MFMA GEMM path (M > 8)")
    lib = L.load()
    pair = tuning.lookup_pair(pa, pb, st) if TUNING else None
    if pair is not None:
        sep = (tuning.table().get(tuning.key_of(pa)), tuning.table().get(tuning.key_of(pb)))
        if all(sep) and pair[2] >= sep[0][2] + sep[1][2]:
            pair = None
    if pair is None:
        _launch_gemm(pa, st, "da_gemm_bf16(linear)")
        _launch_gemm(pb, st, "da_gemm_bf16(linear)")
    else:
        pa.tile, pa.staging, pa.split_k, pb.split_k = pair[0], pair[1], 1, 1
        L.check(lib.da_gemm_pair_bf16(C.byref(pa), C.byref(pb), st), "da_gemm_pair_bf16")
    return pa._out, pb._out


def _linear_params(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, act: int = L.ACT_NONE,
                   residual: Optional[torch.Tensor] = None, rowvec: Optional[torch.Tensor] = None,
                   rows_per_batch: int = 0, alpha: float = 1.0, out_scale: float = 1.0,
                   out: Optional[torch.Tensor] = None, out_f32: bool = False, bias_rows: Optional[torch.Tensor] = None,
                   gate: Optional[torch.Tensor] = None, tile: Optional[int] = None, staging: Optional[int] = None,
                   split_k: Optional[int] = None, stats_out: Optional[RowStats] = None, ln: Optional[tuple] = None,
                   k_valid: int = 0, prefetch: Optional[torch.Tensor] = None, vt_out: Optional[tuple] = None,
                   xattn: Optional[dict] = None):
    """Checks + da_gemm_params of one nn.Linear problem; returns (params, stream), or (None, result) when the skinny-M
    kernel handled it."""
    _req(x, "x"), _req(w, "w")
    M, K = x.shape
    N, Kw = w.shape
    if K != Kw:
        raise ValueError(f"linear: K mismatch {K} vs {Kw}")
    n_out = N // 2 if act in (L.ACT_GEGLU, L.ACT_GEGLU_TANH) else N
    if vt_out is not None:      # (vt [N - col0][>= M] bf16, col0): columns >= col0 leave transposed, C holds the first col0 columns
        vt_t, vt_col0 = vt_out
        _req(vt_t, "vt_out")
        if vt_t.dim() != 2 or vt_t.stride(1) != 1 or vt_t.shape[0] != N - vt_col0 or vt_t.shape[1] < M or not 0 < vt_col0 < N:
            raise ValueError("linear(vt_out=): expected (bf16 [N - col0][>= M], col0)")
        if M <= 8 or act != L.ACT_NONE or residual is not None or gate is not None or rowvec is not None or stats_out is not None:
            raise ValueError("linear(vt_out=): plain / LayerNorm-folded projections of more than 8 rows only")
        n_out = vt_col0
    if M <= 8 and act in (L.ACT_NONE, L.ACT_SILU, L.ACT_GELU_TANH) and rowvec is None and not out_f32 \
            and alpha == 1.0 and out_scale == 1.0 and bias_rows is None and gate is None and stats_out is None and ln is None:
        return None, linear_small_m(x, w, bias, act_out=act, residual=residual, out=out)
    if out is None:
        out = torch.empty((M, n_out), device=x.device, dtype=torch.float32 if out_f32 else bf16)
    p = L.GemmParams()
    p.A, p.A2, p.W, p.C = x.data_ptr(), None, w.data_ptr(), out.data_ptr()
    p.bias, p.rowvec, p.residual = _ptr(bias), _ptr(rowvec), _ptr(residual)
    p.M, p.N, p.K = M, N, K
    p.lda, p.ldw, p.ldc = _rows2d(x, "x"), _rows2d(w, "w"), _rows2d(out, "out")
    p.ldr = _rows2d(residual, "residual") if residual is not None else 0
    p.ld_rowvec = _rows2d(rowvec, "rowvec") if rowvec is not None else 0
    p.bias_rows, p.gate = _ptr(bias_rows), _ptr(gate)
    p.ld_gate = _rows2d(gate, "gate") if gate is not None else 0
    _prefetch_hook(p, x, w)       # (always: the recorded launch sequence must stay in step)
    if prefetch is not None:      # the caller knows better what is met cold next (cross-attention: the layer's K / V^T)
        p.prefetch, p.prefetch_bytes = prefetch.data_ptr(), (prefetch.numel() * prefetch.element_size()) & ~15
    if gate is not None:
        if gate.dtype not in (bf16, torch.float32):
            raise TypeError("linear: gate must be bf16 (Flux rounding) or float32 (Wan rounding)")
        p.gate_f32 = int(gate.dtype == torch.float32)
    # out may BE the residual (accumulating launch: one lane reads and writes each element); variant tuning then
    # times its repeated launches on a scratch output
    inplace = residual is not None and out.data_ptr() == residual.data_ptr()
    if inplace and p.ldr != p.ldc:
        raise ValueError("linear: an aliased residual must have the output's row stride")
    p.rows_per_batch = rows_per_batch
    p.alpha, p.out_scale, p.act, p.out_f32, p.conv = alpha, out_scale, act, int(out_f32), 0
    if not 0 <= k_valid <= K:
        raise ValueError(f"linear: k_valid {k_valid} outside [0, K = {K}]")
    p.k_valid = 0 if k_valid == K else k_valid
    st = _stream()
    if ln is not None:
        rs, fold = ln
        if rs.parts <= 0 or rs.buf.shape[0] != M or fold.s.numel() != N or fold.c.numel() != N:
            raise ValueError("linear(ln=): statistics / fold vectors do not match this problem (was the producer run?)")
        p.ln_stats, p.ln_stats_ld, p.ln_parts = rs.buf.data_ptr(), rs.buf.shape[1] * 2, rs.parts
        p.ln_s, p.ln_c, p.ln_eps = fold.s.data_ptr(), fold.c.data_ptr(), fold.eps
    if vt_out is not None:
        p.vt, p.vt_col0, p.ld_vt = vt_out[0].data_ptr(), int(vt_out[1]), vt_out[0].stride(0)
    if xattn is not None:
        xk, xvt = xattn["k"], xattn["vt"]
        _req(xk, "xattn k"), _req(xvt, "xattn vt")
        if not xattn_eligible(M, N, xattn["seq"], 64, xattn["skv_alloc"]) or xk.shape[1] != N or xvt.shape[0] != N:
            raise ValueError("linear(xattn=): heads of 64 channels in pairs (N % 128 == 0), row tiles of 128 inside one batch, "
                             f"<= {XATTN_MAX_KEYS} text tokens")
        if act != L.ACT_NONE or residual is not None or gate is not None or rowvec is not None or stats_out is not None \
                or vt_out is not None or out_f32 or out_scale != 1.0 or bias_rows is not None or M <= 8:
            raise ValueError("linear(xattn=): nothing but bias / the LayerNorm fold combines with the attention epilogue")
        p.xa_k, p.xa_vt = xk.data_ptr(), xvt.data_ptr()
        p.xa_skv, p.xa_skv_alloc, p.xa_k_ld, p.xa_vt_ld = int(xattn["skv"]), int(xattn["skv_alloc"]), _rows2d(xk, "xattn k"), _rows2d(xvt, "xattn vt")
        p.xa_scale = float(xattn["scale"])
        p.rows_per_batch = int(xattn["seq"])
        p._keep_xa = (xk, xvt)
        if tile is None:
            tile, staging, split_k = L.TILE_K2_128x128, XATTN_STAGING, 1
    tile_given = tile is not None
    if vt_out is not None and tile is None:
        tile, staging = qkv_variant(p, st)       # the transposed column block exists on two tiles of the second family only
        split_k = 1
    if (stats_out is not None or ln is not None) and tile is None:
        # The LayerNorm fold lives in its own kernel instantiations (a subset of the variants).  Round 4: the second kernel family
        # carries it on the tiles the SDXL transformer blocks use -- when the per-shape table sends the SAME problem without the
        # fold to one of them (and the operands have the 16-byte aligned rows its row-contiguous store path needs), the folded
        # launch takes that (tile, staging); otherwise the first family's fixed choices (profiles/r02d_layernorm_fold.md).
        geglu_ = act in (L.ACT_GEGLU, L.ACT_GEGLU_TANH)
        ent = tuning.table().get(tuning.key_of(p)) if TUNING else None
        # round 6: the GEGLU projection's own eight-phase tile (k3:256x320) carries the consumer side of the fold itself (whole tiles,
        # statistics rows of 16-byte aligned slots); DIFFUSERS_AMD_LN_FOLD_K3=0 sends folded GEGLU launches back to k1:128x320
        k3_geglu = (geglu_ and LN_FOLD_K3 and ent is not None and ent[0] == L.TILE_K3_256x320 and ln is not None and stats_out is None
                    and M % 256 == 0 and N % 320 == 0)
        if geglu_ and not k3_geglu and ent is not None and ent[0] in (L.TILE_K1_256x256, L.TILE_K1_128x256, L.TILE_K1_256x128,
                                                                      L.TILE_K1_256x320, L.TILE_K3_256x256, L.TILE_K3_256x320):
            ent = (L.TILE_K1_128x320, L.STAGE_LDS_DIRECT) + tuple(ent[2:])     # the GEGLU tile of that family that carries the fold
        k2_ok = (ent is not None and ent[0] in ((L.TILE_K1_128x320, L.TILE_K3_256x320) if geglu_ else (L.TILE_K2_128x80, L.TILE_K2_128x160))
                 and not out_f32 and p.ldc % 8 == 0 and out.data_ptr() % 16 == 0 and N % 16 == 0 and gate is None
                 and (residual is None or (p.ldr % 8 == 0 and residual.data_ptr() % 16 == 0))
                 and not (ent[0] == L.TILE_K2_128x160 and ent[1] in (L.STAGE_LDS_DIRECT3, L.STAGE_PINGPONG3))
                 and not (geglu_ and ent[1] != L.STAGE_LDS_DIRECT))
        if k2_ok and LN_FOLD_K2:
            tile, staging, split_k = ent[0], ent[1], 1
        else:
            tile, staging, split_k = ((L.TILE_128x128, L.STAGE_LDS_DIRECT) if geglu_ else (L.TILE_128x64, L.STAGE_LDS_DIRECT3)) + (1,)
    _select_variant(p, tile, staging, st, inplace=inplace, split_k=split_k, device=x.device)
    if stats_out is not None:
        if act in (L.ACT_GEGLU, L.ACT_GEGLU_TANH) or out_f32 or stats_out.buf.shape[0] != M:
            raise ValueError("linear(stats_out=): bf16 non-GEGLU outputs only, one statistics row per output row")
        p.stats_out, p.stats_ld = stats_out.buf.data_ptr(), stats_out.buf.shape[1] * 2
        stats_out.parts = int(L.load().da_gemm_stats_parts(C.byref(p)))
        if stats_out.parts > STATS_MAX_CONSUMED and not tile_given:
            # too many narrow column tiles for the consumer's one-batch read: take a 128- / 256-wide tile instead (every
            # tile computes the same bits, so this is a speed decision only)
            p.tile, p.staging, p.split_k = (L.TILE_128x128 if N <= 128 * STATS_MAX_CONSUMED else L.TILE_128x256), L.STAGE_LDS_DIRECT, 1
            stats_out.parts = int(L.load().da_gemm_stats_parts(C.byref(p)))
        if not 0 < stats_out.parts <= min(stats_out.buf.shape[1], STATS_MAX_CONSUMED):
            raise ValueError(f"linear(stats_out=): {stats_out.parts} column tiles: the consumer reads at most "
                             f"{STATS_MAX_CONSUMED} partials per row (use a wider tile for this producer)")
    p._out = out            # keeps the output (and through it nothing else) alive next to the raw pointers
    p._keep = (x, w, bias, residual, rowvec, bias_rows, gate, stats_out, ln, vt_out)
    return p, st


def conv2d_nhwc(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, ksize: int = 3,
                x2: Optional[torch.Tensor] = None, stride: int = 1, up: bool = False, pad: Optional[int] = None,
                rowvec: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                out_scale: float = 1.0, act: int = L.ACT_NONE, tile: Optional[int] = None,
                staging: Optional[int] = None, pad_after: int = 0, out: Optional[torch.Tensor] = None,
                split_k: Optional[int] = None, k_valid: int = 0) -> torch.Tensor:
    """Implicit-GEMM Conv2d on channels-last tensors.  x: [B][H][W][C1] (x2: [B][H][W][C2] = fused channel concat),
    w: [Cout][k][k][C1+C2] flattened to [Cout][k*k*(C1+C2)].  up=True fuses a nearest 2x upsample of the input.
    rowvec [B][Cout] is added per batch (time embedding); residual is [B][Hout][Wout][Cout].  ``out`` (contiguous
    [B][Hout][Wout][Cout]) may be the SAME tensor as ``residual``: every output element is read and written by one
    lane, which is how the temporal taps of a causal Conv3d accumulate in place (autoencoder_kl_wan.py).
    ``k_valid`` > 0 (single source only): channels >= k_valid of x AND of every tap of w are zero padding up to the 64-wide
    K granule; the kernel skips the MFMA steps that would multiply them (da_gemm_params.k_valid)."""
    _req(x, "x"), _req(w, "w")
    B, H, W_, C1 = x.shape
    C2 = 0
    if x2 is not None:
        _req(x2, "x2")
        if x2.shape[:3] != x.shape[:3]:
            raise ValueError("conv2d_nhwc: x2 spatial shape mismatch")
        C2 = x2.shape[3]
    if not x.is_contiguous() or (x2 is not None and not x2.is_contiguous()):
        raise ValueError("conv2d_nhwc: inputs must be contiguous NHWC")
    Cout, K = w.shape
    if K != ksize * ksize * (C1 + C2):
        raise ValueError(f"conv2d_nhwc: weight K={K} != {ksize}*{ksize}*{C1 + C2}")
    if pad is None:
        pad = (ksize - 1) // 2
    Hv, Wv = (2 * H, 2 * W_) if up else (H, W_)
    # pad_after: extra zero rows / columns on the bottom / right only (Downsample2D with padding=0 pads (0, 1, 0, 1),
    # downsampling.py:139-141); out-of-range taps read zeros, so only the output extent changes
    Hout = (Hv + 2 * pad + pad_after - ksize) // stride + 1
    Wout = (Wv + 2 * pad + pad_after - ksize) // stride + 1
    if out is None:
        out = torch.empty((B, Hout, Wout, Cout), device=x.device, dtype=bf16)
    else:
        _req(out, "out")
        if tuple(out.shape) != (B, Hout, Wout, Cout) or not out.is_contiguous():
            raise ValueError("conv2d_nhwc: out must be contiguous [B][Hout][Wout][Cout]")
    inplace = residual is not None and residual.data_ptr() == out.data_ptr()
    p = L.GemmParams()
    p.A, p.A2, p.W, p.C = x.data_ptr(), _ptr(x2), w.data_ptr(), out.data_ptr()
    p.bias, p.rowvec, p.residual = _ptr(bias), _ptr(rowvec), _ptr(residual)
    p.M, p.N, p.K = B * Hout * Wout, Cout, K
    p.lda, p.ldw, p.ldc = 0, K, Cout
    if residual is not None:
        _req(residual, "residual")
        if tuple(residual.shape) != (B, Hout, Wout, Cout) or not residual.is_contiguous():
            raise ValueError("conv2d_nhwc: residual must be contiguous [B][Hout][Wout][Cout]")
        p.ldr = Cout
    if rowvec is not None:
        _req(rowvec, "rowvec")
        p.ld_rowvec = _rows2d(rowvec, "rowvec")
        p.rows_per_batch = Hout * Wout
    p.alpha, p.out_scale, p.act, p.out_f32, p.conv = 1.0, out_scale, act, 0, ksize
    p.Hin, p.Win, p.C1, p.C2, p.Hout, p.Wout = H, W_, C1, C2, Hout, Wout
    p.stride, p.up, p.pad = stride, int(up), pad
    if not 0 <= k_valid <= C1 or (k_valid and C2):
        raise ValueError(f"conv2d_nhwc: k_valid {k_valid} outside [0, C1 = {C1}] (or given with a second source)")
    p.k_valid = 0 if k_valid == C1 else k_valid
    _prefetch_hook(p, x, w)
    st = _stream()
    _select_variant(p, tile, staging, st, inplace=inplace, split_k=split_k, device=x.device)
    _launch_gemm(p, st, "da_gemm_bf16(conv)")
    return out


def linear_small_m(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, act_in: int = L.ACT_NONE,
                   act_out: int = L.ACT_NONE, residual: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(x, "x"), _req(w, "w")
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K or not w.is_contiguous():
        raise ValueError("linear_small_m: weight must be contiguous [N][K]")
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=bf16)
    L.check(L.load().da_linear_small_m_bf16(
        x.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(), M, N, K, _rows2d(x, "x"),
        _rows2d(out, "out"), _rows2d(residual, "residual") if residual is not None else 0, act_in, act_out, _stream()),
        "da_linear_small_m_bf16")
    return out


def pack_geglu(w: torch.Tensor, bias: Optional[torch.Tensor]):
    """Reorder GEGLU.proj rows ([value(4d) ; gate(4d)], activations.py:104) into 64-row groups [32 value | 32 gate] so
    one wave holds matching value/gate columns in the same lanes."""
    n2 = w.shape[0]
    n = n2 // 2
    if n % 64:
        raise ValueError("pack_geglu: inner dim must be a multiple of 64")
    idx = torch.arange(n, device=w.device).view(n // 32, 32)
    order = torch.cat([idx, idx + n], dim=1).reshape(-1)
    wp = w.index_select(0, order).contiguous()
    bp = bias.index_select(0, order).contiguous() if bias is not None else None
    return wp, bp


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """[Cout][Cin][kh][kw] (torch) -> [Cout][kh*kw*Cin] (K index = tap * Cin + c)."""
    co, ci, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci).contiguous()


# ----------------------------------------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------------------------------------
def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, *, B: int, H: int, D: int, Sq: int, Skv: int,
              Skv_alloc: int, q_row_stride: int, k_row_stride: int, q_batch_stride: int, k_batch_stride: int,
              vt_ld: int, vt_batch_stride: int, scale: Optional[float] = None,
              out: Optional[torch.Tensor] = None, ring_slots: int = 0, causal: bool = False,
              bias: Optional[torch.Tensor] = None, q_block: int = 0, pv_delay: int = 0, algo: int = 0,
              kv_split: int = 0) -> torch.Tensor:
    """Flash attention over strided views; returns out [B*Sq][H*D].  ``causal`` / ``bias`` (D = 64 only): the masked
    variant for the text encoders -- ``bias`` is [B or 1][H or 1][Sq][>= ceil64(Skv)] (bf16 or fp32), added to
    scale * q.k^T; entries <= -1e29 mask a key.  ``kv_split``: 0 = the library decides whether the launch's last partial round of
    query blocks is split over the keys (module flag ATTN_SPLIT), 1 = never, 2..8 = pinned (tests)."""
    _req(q, "q"), _req(k, "k"), _req(vt, "vt")
    if out is None:
        out = torch.empty((B * Sq, H * D), device=q.device, dtype=bf16)
    o_row_stride = _rows2d(out, "out")  # `out` may be a column block of a wider buffer (fused torch.cat on channels)
    p = L.AttentionParams()
    p.q, p.k, p.vt, p.out = q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr()
    p.B, p.H, p.Sq, p.Skv, p.Skv_alloc, p.D = B, H, Sq, Skv, Skv_alloc, D
    p.q_batch_stride, p.k_batch_stride, p.vt_batch_stride = q_batch_stride, k_batch_stride, vt_batch_stride
    p.o_batch_stride = Sq * o_row_stride
    p.q_row_stride, p.k_row_stride, p.vt_ld, p.o_row_stride = q_row_stride, k_row_stride, vt_ld, o_row_stride
    p.scale = (D ** -0.5) if scale is None else scale
    p.ring_slots = ring_slots
    p.q_block = q_block
    p.pv_delay = pv_delay
    p.algo = algo
    p.causal = int(causal)
    p.kv_split = kv_split if ATTN_SPLIT or kv_split > 1 else 1
    # the workspace rides along only where a split can pay: long key sequences and at least a quarter of the chip's CUs in blocks
    # (small launches -- tiny models, text encoders -- keep a struct without it: their plans stay free of the 128 MiB region)
    if p.kv_split != 1 and bias is None and not causal and D in (64, 128) and Skv >= 256 and B * H * ((Sq + 127) // 128) >= 64:
        ws = attn_split_workspace(q.device, _stream())
        p.split_ws, p.split_ws_bytes = ws.data_ptr(), ws.numel()
    if bias is not None:
        _req(bias, "bias", None)
        if bias.dtype not in (bf16, torch.float32) or bias.dim() != 4 or bias.stride(3) != 1 or \
                (bias.shape[2] != 1 and bias.shape[2] < Sq):
            raise ValueError("attention: bias must be a bf16 / fp32 [B|1][H|1][Sq|1][>= ceil64(Skv)] tensor, last dim contiguous")
        if bias.shape[0] not in (1, B) or bias.shape[1] not in (1, H):
            raise ValueError("attention: bias batch / head dims must be 1 or match")
        p.bias, p.bias_f32 = bias.data_ptr(), int(bias.dtype == torch.float32)
        p.bias_batch_stride = bias.stride(0) if bias.shape[0] > 1 else 0
        p.bias_head_stride = bias.stride(1) if bias.shape[1] > 1 else 0
        p.bias_row_stride = bias.stride(2) if bias.shape[2] > 1 else 0   # one row for every query (key-padding mask)
    L.check(L.load().da_attention_bf16(C.byref(p), _stream()), "da_attention_bf16")
    return out


def softmax_rows(scores: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Row softmax of fp32 scores [M][N] -> bf16.  ``out`` may be wider than N (row stride free): columns >= N are left
    untouched, so a zero-initialised [M][ceil64(N)] buffer is a valid K-padded GEMM operand."""
    _req(scores, "scores", torch.float32)
    M, N = scores.shape
    if out is None:
        out = torch.empty((M, N), device=scores.device, dtype=bf16)
    else:
        _req(out, "out")
        if out.shape[0] != M or out.shape[1] < N:
            raise ValueError("softmax_rows: out must be [M][>= N]")
    L.check(L.load().da_softmax_rows_f32_bf16(scores.data_ptr(), out.data_ptr(), M, N, _rows2d(scores, "scores"),
                                              _rows2d(out, "out"), _stream()), "da_softmax_rows_f32_bf16")
    return out


# ----------------------------------------------------------------------------------------------------------------------
# norms
# ----------------------------------------------------------------------------------------------------------------------
_gn_ws = {}
# One-launch GroupNorm over several workgroups per slab (include/diffusers_amd.h da_groupnorm_nhwc_bf16 `sync`): the arrival counters
# and partial statistics the parts of a slab exchange -- one buffer per (device, stream), zeroed ONCE here, then the kernels' own.
# OFF by default (DIFFUSERS_AMD_GN_MULTI=1 turns it on): measured on the GroupNorm shapes of an SDXL step it wins 2.6-3.5 us on five
# launches of a step and is a wash or a loss elsewhere -- 703 -> 696 us over the step's 37 GroupNorms, + 0.4 % on the image (inside the
# pair-to-pair spread), profiles/r06_groupnorm_several_workgroups.jsonl -- and its parts WAIT for each other, which is only safe
# while every part is resident: two such launches from two concurrent streams can starve each other until the bounded wait gives up.
GN_MULTI = os.environ.get("DIFFUSERS_AMD_GN_MULTI", "0") == "1"
_gn_sync = {}


def gn_sync_workspace(device: torch.device, stream: int) -> torch.Tensor:
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream)
    ws = _gn_sync.get(key)
    if ws is None:
        ws = torch.zeros(int(L.load().da_groupnorm_sync_bytes()), dtype=torch.uint8, device=device)
        _gn_sync[key] = ws
    return ws


def gn_sync_error(device=None) -> bool:
    """True if a part of a multi-workgroup GroupNorm ever gave up waiting for its peers on this device (diagnostics / tests)."""
    off = 4096 * 4
    return any(bool(w[off:off + 4].view(torch.int32).item()) for (d, _), w in _gn_sync.items()
               if device is None or d == torch.device(device).index)


def group_norm_nhwc(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float,
                    silu: bool = False, x2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GroupNorm (+SiLU) on [B][H][W][C] or [B][HW][C]; x2 = second channel-concat source."""
    _req(x, "x"), _req(gamma, "gamma"), _req(beta, "beta")
    if not x.is_contiguous() or (x2 is not None and not x2.is_contiguous()):
        raise ValueError("group_norm_nhwc: contiguous channels-last input required")
    B = x.shape[0]
    C1 = x.shape[-1]
    Ctot = C1 + (x2.shape[-1] if x2 is not None else 0)
    HW = x.numel() // (B * C1)
    lib = L.load()
    nbytes = lib.da_groupnorm_workspace_bytes(B, HW, Ctot, groups)
    ws = torch.empty((max(nbytes, 4) // 4,), device=x.device, dtype=torch.float32)
    y = torch.empty(tuple(x.shape[:-1]) + (Ctot,), device=x.device, dtype=bf16)
    # the sync buffer rides along for tensors of 2 MB and more (smaller ones are LDS-resident per slab or latency-bound either way;
    # tiny models' plans stay free of the region)
    sync = gn_sync_workspace(x.device, _stream()).data_ptr() if GN_MULTI and B * HW * Ctot >= (1 << 20) else None
    L.check(lib.da_groupnorm_nhwc_bf16(x.data_ptr(), _ptr(x2), C1, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                       ws.data_ptr(), B, HW, Ctot, groups, eps, L.ACT_SILU if silu else L.ACT_NONE,
                                       sync, _stream()), "da_groupnorm_nhwc_bf16")
    return y


def rms_norm(x: torch.Tensor, gamma: torch.Tensor, eps: float) -> torch.Tensor:
    """T5LayerNorm over the last dim of a token matrix [M][C] (da_rmsnorm_bf16)."""
    _req(x, "x"), _req(gamma, "gamma")
    M, Cc = x.shape
    y = torch.empty((M, Cc), device=x.device, dtype=bf16)
    L.check(L.load().da_rmsnorm_bf16(x.data_ptr(), gamma.data_ptr(), y.data_ptr(), M, Cc, _rows2d(x, "x"), Cc, eps, _stream()),
            "da_rmsnorm_bf16")
    return y


def rmsnorm_channels(x: torch.Tensor, gamma: torch.Tensor, *, real_channels: int, silu: bool = False) -> torch.Tensor:
    """WanRMS_norm over the last (channel) dim of a contiguous channels-last tensor; ``real_channels`` is the model's
    channel count when the stored one is zero-padded (it sets the sqrt(C) scale)."""
    _req(x, "x"), _req(gamma, "gamma")
    Cc = x.shape[-1]
    if not x.is_contiguous() or gamma.numel() != Cc:
        raise ValueError("rmsnorm_channels: contiguous channels-last input and a [C] gamma required")
    y = torch.empty_like(x)
    L.check(L.load().da_rmsnorm_channels_bf16(x.data_ptr(), gamma.data_ptr(), y.data_ptr(), x.numel() // Cc, Cc,
                                              float(real_channels) ** 0.5, L.ACT_SILU if silu else L.ACT_NONE,
                                              _stream()), "da_rmsnorm_channels_bf16")
    return y


def layer_norm(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float, *,
               mod_scale: Optional[torch.Tensor] = None, mod_shift: Optional[torch.Tensor] = None,
               rows_per_batch: int = 0) -> torch.Tensor:
    _req(x, "x")
    M, Cc = x.shape
    y = torch.empty((M, Cc), device=x.device, dtype=bf16)
    mod_ld = _rows2d(mod_scale, "mod_scale") if mod_scale is not None else 0
    mod_f32 = 0
    if mod_scale is not None:
        if mod_scale.dtype != mod_shift.dtype or mod_scale.dtype not in (bf16, torch.float32):
            raise TypeError("layer_norm: mod_scale / mod_shift must both be bf16 or both float32")
        mod_f32 = int(mod_scale.dtype == torch.float32)
    L.check(L.load().da_layernorm_bf16(x.data_ptr(), _ptr(gamma), _ptr(beta), y.data_ptr(), _ptr(mod_scale),
                                       _ptr(mod_shift), mod_ld, mod_f32, rows_per_batch, M, Cc, _rows2d(x, "x"), Cc,
                                       eps, _stream()), "da_layernorm_bf16")
    return y


def rmsnorm_rope_(x: torch.Tensor, *, heads: int, head_dim: int, col_offsets, weights=None, eps: float = 1e-6,
                  cos: Optional[torch.Tensor] = None, sin: Optional[torch.Tensor] = None, rope_row0: int = 0,
                  rows_per_batch: int = 0, norm: str = "per_head") -> torch.Tensor:
    """IN PLACE per-head RMSNorm (+ weight) and rotary embedding on column blocks (e.g. the q and k thirds of a fused
    QKV projection) of the token-major buffer x [rows][ld].  cos / sin: fp32 [>= rope_row0 + rows_per_batch][head_dim]."""
    _req(x, "x")
    rows = x.shape[0]
    ld = _rows2d(x, "x")
    parts = len(col_offsets)
    offs = (C.c_int * parts)(*[int(o) for o in col_offsets])
    wptr = None
    if weights is not None:
        for w_ in weights:
            if w_ is not None:
                _req(w_, "weight")
        wptr = (C.c_void_p * parts)(*[(w_.data_ptr() if w_ is not None else None) for w_ in weights])
    if cos is not None:
        _req(cos, "cos", torch.float32), _req(sin, "sin", torch.float32)
        if cos.shape[-1] != head_dim or not cos.is_contiguous() or not sin.is_contiguous():
            raise ValueError("rmsnorm_rope_: cos / sin must be contiguous [rows][head_dim] fp32")
    mode = {"none": 0, "per_head": 1, "across_heads": 2}[norm]
    L.check(L.load().da_rmsnorm_rope_bf16(x.data_ptr(), ld, rows, rows_per_batch or rows, heads, head_dim, parts, offs,
                                          wptr, eps, _ptr(cos), _ptr(sin), rope_row0, mode, _stream()),
            "da_rmsnorm_rope_bf16")
    return x


# ----------------------------------------------------------------------------------------------------------------------
# sampler
# ----------------------------------------------------------------------------------------------------------------------
def _dt(t: torch.Tensor) -> int:
    if t.dtype == bf16:
        return L.DTYPE_BF16
    if t.dtype == torch.float32:
        return L.DTYPE_F32
    raise TypeError(f"sampler kernels support bf16 / fp32 latents, got {t.dtype}")


def _check_sampler(what: str, x: torch.Tensor, model_out: torch.Tensor, out: Optional[torch.Tensor], cfg: bool,
                   same_dtype: bool = True) -> None:
    """The sampler kernels index their operands flat: every tensor must be a contiguous HIP tensor of the right size."""
    _req(x, "sample", None), _req(model_out, "model_output", None)
    if model_out.numel() != x.numel() * (2 if cfg else 1):
        raise ValueError(f"{what}: model_output must hold {'[2 x sample] (uncond, cond)' if cfg else 'as many'} "
                         f"elements{'' if cfg else ' as the sample'} (got {model_out.numel()} vs {x.numel()})")
    if same_dtype and model_out.dtype != x.dtype:
        raise TypeError(f"{what}: model_output ({model_out.dtype}) and sample ({x.dtype}) must have the same dtype")
    if not x.is_contiguous() or not model_out.is_contiguous():
        raise ValueError(f"{what}: sample and model_output must be contiguous")
    if out is not None:
        _req(out, "out", None)
        if out.numel() != x.numel() or not out.is_contiguous():
            raise ValueError(f"{what}: out must be a contiguous tensor of the sample's size")


def euler_scale_model_input(x: torch.Tensor, table: torch.Tensor, step_idx: torch.Tensor, rep: int = 1) -> torch.Tensor:
    _req(x, "x", None)
    if not x.is_contiguous():
        raise ValueError("euler_scale_model_input: contiguous sample required")
    out = torch.empty((rep * x.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
    L.check(L.load().da_euler_scale_model_input(x.data_ptr(), out.data_ptr(), table.data_ptr(), step_idx.data_ptr(),
                                                rep, x.numel(), _dt(x), _stream()), "da_euler_scale_model_input")
    return out


def euler_step(eps: torch.Tensor, x: torch.Tensor, table: torch.Tensor, step_idx: torch.Tensor, *, cfg: bool,
               guidance: float, out: Optional[torch.Tensor] = None, pred_type: int = L.PRED_EPSILON) -> torch.Tensor:
    _check_sampler("euler_step", x, eps, out, cfg)
    if out is None:
        out = torch.empty_like(x)
    elif out.dtype != x.dtype:
        raise TypeError("euler_step: out must have the sample's dtype")
    L.check(L.load().da_euler_step(eps.data_ptr(), x.data_ptr(), out.data_ptr(), table.data_ptr(), step_idx.data_ptr(),
                                   int(cfg), guidance, x.numel(), _dt(x), int(pred_type), _stream()), "da_euler_step")
    return out


def x0_linear_step(eps: torch.Tensor, x: torch.Tensor, noise: Optional[torch.Tensor], table: torch.Tensor,
                   step_idx: torch.Tensor, *, cfg: bool, guidance: float,
                   out: Optional[torch.Tensor] = None, noise_step_stride: int = 0,
                   pred_type: int = L.PRED_EPSILON) -> torch.Tensor:
    _check_sampler("x0_linear_step", x, eps, out, cfg)
    if out is None:
        out = torch.empty_like(x)
    elif out.dtype != x.dtype:
        raise TypeError("x0_linear_step: out must have the sample's dtype")
    if noise is not None:
        _req(noise, "noise", None)
        if noise.dtype != x.dtype or not noise.is_contiguous():
            raise ValueError("x0_linear_step: noise must be contiguous and have the dtype of the sample")
        if noise.numel() < x.numel() or (noise_step_stride and noise.numel() % x.numel()):
            raise ValueError("x0_linear_step: noise must hold one sample-sized block (per step, with a step stride)")
    L.check(L.load().da_x0_linear_step(eps.data_ptr(), x.data_ptr(), _ptr(noise), noise_step_stride, out.data_ptr(),
                                       table.data_ptr(),
                                       step_idx.data_ptr(), int(cfg), guidance, x.numel(), _dt(x), int(pred_type),
                                       _stream()),
            "da_x0_linear_step")
    return out


def flowmatch_step(v: torch.Tensor, x: torch.Tensor, table: torch.Tensor, step_idx: torch.Tensor, *, cfg: bool = False,
                   guidance: float = 0.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """FlowMatch-Euler update.  The result has the MODEL OUTPUT's dtype, as in the reference (the sample is upcast to
    fp32 for the update and the result cast ``.to(model_output.dtype)``): (sample, model output) = bf16/bf16, f32/f32 or
    f32/bf16 -- the last is what the reference's Wan loop passes on its first step."""
    _check_sampler("flowmatch_step", x, v, out, cfg, same_dtype=False)
    if (_dt(x), _dt(v)) not in ((L.DTYPE_BF16, L.DTYPE_BF16), (L.DTYPE_F32, L.DTYPE_F32), (L.DTYPE_F32, L.DTYPE_BF16)):
        raise TypeError("flowmatch_step: (sample, model output) dtypes must be bf16/bf16, f32/f32 or f32/bf16")
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=v.dtype)
    elif out.dtype != v.dtype:
        raise TypeError("flowmatch_step: out must have the model output's dtype (the reference casts the result to it)")
    L.check(L.load().da_flowmatch_step(v.data_ptr(), x.data_ptr(), out.data_ptr(), table.data_ptr(),
                                       step_idx.data_ptr(), int(cfg), guidance, x.numel(), _dt(v), _dt(x), _stream()),
            "da_flowmatch_step")
    return out


def unipc_flow_step_(v: torch.Tensor, x: torch.Tensor, last: torch.Tensor, m1: torch.Tensor, m2: torch.Tensor,
                     coef: torch.Tensor, step_idx: torch.Tensor, *, cfg: bool = False, guidance: float = 0.0) -> torch.Tensor:
    """IN PLACE UniPC step (see da_unipc_flow_step): x <- next sample; last / m1 / m2 roll."""
    _req(x, "x", None), _req(v, "v", None)
    for t_ in (last, m1, m2):
        if t_.dtype != x.dtype or t_.numel() != x.numel() or not t_.is_contiguous():
            raise ValueError("unipc_flow_step_: history tensors must match the sample (dtype, size, contiguous)")
    if v.numel() != x.numel() * (2 if cfg else 1):
        raise ValueError("unipc_flow_step_: model output must be [2 x sample] with cfg")
    if (_dt(x), _dt(v)) not in ((L.DTYPE_F32, L.DTYPE_F32), (L.DTYPE_F32, L.DTYPE_BF16), (L.DTYPE_BF16, L.DTYPE_BF16)):
        raise TypeError("unipc_flow_step_: (sample, model output) dtypes must be f32/f32, f32/bf16 or bf16/bf16")
    L.check(L.load().da_unipc_flow_step(v.data_ptr(), x.data_ptr(), last.data_ptr(), m1.data_ptr(), m2.data_ptr(),
                                        coef.data_ptr(), step_idx.data_ptr(), int(cfg), float(guidance), x.numel(),
                                        _dt(x), _dt(v), _stream()), "da_unipc_flow_step")
    return x


def cfg_rescale(eps2b: torch.Tensor, guidance: float, guidance_rescale: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``rescale_noise_cfg`` after the CFG combine (pipeline_stable_diffusion.py:69-92): eps2b [2B][...] = (uncond, cond) ->
    [B][...] guided and rescaled noise prediction, rounding as the reference's op chain does.  Two launches (per-sample
    statistics, apply); only on the ``guidance_rescale > 0`` path, which is off by default."""
    _req(eps2b, "model_output", None)
    if eps2b.dtype not in (bf16, torch.float32) or eps2b.shape[0] % 2 or not eps2b.is_contiguous():
        raise ValueError("cfg_rescale: contiguous bf16 / fp32 model output of an even batch (uncond, cond)")
    B = eps2b.shape[0] // 2
    n_per = eps2b[0].numel()
    if out is None:
        out = torch.empty((B,) + tuple(eps2b.shape[1:]), device=eps2b.device, dtype=eps2b.dtype)
    ws = torch.empty((B,), device=eps2b.device, dtype=torch.float32)
    L.check(L.load().da_cfg_rescale(eps2b.data_ptr(), out.data_ptr(), ws.data_ptr(), B, n_per, float(guidance),
                                    float(guidance_rescale), _dt(eps2b), _stream()), "da_cfg_rescale")
    return out


def cast_f32_bf16(x: torch.Tensor, rep: int = 1) -> torch.Tensor:
    """fp32 -> bf16, replicated ``rep`` times along the batch dim."""
    _req(x, "x", torch.float32)
    out = torch.empty((rep * x.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=bf16)
    L.check(L.load().da_cast_f32_bf16(x.data_ptr(), out.data_ptr(), rep, x.numel(), _stream()), "da_cast_f32_bf16")
    return out


def mul_scalar(x: torch.Tensor, s: float, rep: int = 1) -> torch.Tensor:
    """x * s (in the tensor dtype), replicated ``rep`` times along the batch dim (fused torch.cat([x] * rep))."""
    _req(x, "x", None)
    out = torch.empty((rep * x.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
    L.check(L.load().da_mul_scalar(x.data_ptr(), out.data_ptr(), float(s), rep, x.numel(), _dt(x), _stream()),
            "da_mul_scalar")
    return out


def advance_step(step_idx: torch.Tensor) -> None:
    L.check(L.load().da_advance_step(step_idx.data_ptr(), _stream()), "da_advance_step")


# ----------------------------------------------------------------------------------------------------------------------
# misc
# ----------------------------------------------------------------------------------------------------------------------
def timestep_embedding(t: Optional[torch.Tensor], dim: int, *, batch: int, flip_sin_to_cos: bool, shift: float,
                       scale: float = 1.0, max_period: float = 10000.0, table: Optional[torch.Tensor] = None,
                       step_idx: Optional[torch.Tensor] = None, out_f32: bool = False) -> torch.Tensor:
    """t: float32 device tensor [batch] (or table/step_idx: timestep read from column 7 of the sampler table)."""
    dev = t.device if t is not None else table.device
    out = torch.empty((batch, dim), device=dev, dtype=torch.float32 if out_f32 else bf16)
    if t is not None:
        _req(t, "t", torch.float32)
    L.check(L.load().da_timestep_embedding(_ptr(t), _ptr(table), _ptr(step_idx), out.data_ptr(), batch, dim,
                                           int(flip_sin_to_cos), shift, scale, max_period, int(out_f32), _stream()),
            "da_timestep_embedding")
    return out


def transpose(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[R][C] (row stride free) -> [C][R]."""
    _req(x, "x")
    R, Cc = x.shape
    if out is None:
        out = torch.empty((Cc, R), device=x.device, dtype=bf16)
    L.check(L.load().da_transpose_bf16(x.data_ptr(), out.data_ptr(), R, Cc, _rows2d(x, "x"), _rows2d(out, "out"),
                                       _stream()), "da_transpose_bf16")
    return out


def permute_0213(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[D0][D1][D2][D3] -> [D0][D2][D1][D3] (D3 % 8 == 0); ``out``: any contiguous bf16 tensor of the same size."""
    _req(x, "x")
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError("permute_0213: contiguous 4-D tensor required")
    D0, D1, D2, D3 = x.shape
    if out is None:
        out = torch.empty((D0, D2, D1, D3), device=x.device, dtype=bf16)
    else:
        _req(out, "out")
        if out.numel() != x.numel() or not out.is_contiguous() or out.data_ptr() == x.data_ptr():
            raise ValueError("permute_0213: out must be a distinct contiguous tensor of the same size")
    L.check(L.load().da_permute_0213_bf16(x.data_ptr(), out.data_ptr(), D0, D1, D2, D3, _stream()), "da_permute_0213_bf16")
    return out


def frames_to_ncthw(x: torch.Tensor, *, batch: int, channels: int, lo: float = -1.0, hi: float = 1.0,
                    out_f32: bool = False) -> torch.Tensor:
    """Channels-last frames [B*T][H][W][Cs] -> video [B][channels][T][H][W], clamped to [lo, hi]."""
    _req(x, "x")
    if x.dim() != 4 or not x.is_contiguous() or x.shape[0] % batch:
        raise ValueError("frames_to_ncthw: contiguous [B*T][H][W][Cs] required")
    BT, H, W_, Cs = x.shape
    T = BT // batch
    out = torch.empty((batch, channels, T, H, W_), device=x.device, dtype=torch.float32 if out_f32 else bf16)
    L.check(L.load().da_frames_to_ncthw_bf16(x.data_ptr(), out.data_ptr(), batch, T, H * W_, Cs, channels, lo, hi,
                                             int(out_f32), _stream()), "da_frames_to_ncthw_bf16")
    return out


_PP_MODES = {"pt": 0, "np": 1, "uint8": 2}


def image_postprocess(img: torch.Tensor, output_type: str) -> torch.Tensor:
    """VaeImageProcessor.postprocess on a decoded image [B][C][H][W] or video [B][C][T][H][W] (bf16 / fp32):
    "pt" -> same layout, fp32 in [0, 1]; "np" -> channels last fp32; "uint8" -> channels last bytes (what numpy_to_pil
    hands to PIL)."""
    if not img.is_cuda or img.dtype not in (bf16, torch.float32):
        raise TypeError("image_postprocess: bf16 / fp32 HIP tensor required")
    if not img.is_contiguous() or img.dim() not in (4, 5) or img.shape[1] > 4:
        raise ValueError("image_postprocess: contiguous [B][C<=4][...] tensor required")
    mode = _PP_MODES.get(output_type)
    if mode is None:
        raise ValueError(f"image_postprocess: output_type {output_type!r} (use 'pt', 'np' or 'uint8')")
    B, Cc = img.shape[:2]
    sp = tuple(img.shape[2:])
    HW = 1
    for d in sp:
        HW *= d
    shape = tuple(img.shape) if mode == 0 else (B,) + sp + (Cc,)
    out = torch.empty(shape, device=img.device, dtype=torch.uint8 if mode == 2 else torch.float32)
    L.check(L.load().da_image_postprocess(img.data_ptr(), out.data_ptr(), B, Cc, HW, int(img.dtype == torch.float32), mode,
                                          _stream()), "da_image_postprocess")
    return out


def bcast_add_f32(a: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    """out[b] = a (fp32 [n]) + m[b] (bf16 [B][n]) in fp32."""
    _req(a, "a", torch.float32), _req(m, "m")
    B, n = m.shape
    if a.numel() != n or not a.is_contiguous() or not m.is_contiguous():
        raise ValueError("bcast_add_f32: a must be contiguous [n], m contiguous [B][n]")
    out = torch.empty((B, n), device=m.device, dtype=torch.float32)
    L.check(L.load().da_bcast_add_f32(a.data_ptr(), m.data_ptr(), out.data_ptr(), B, n, _stream()), "da_bcast_add_f32")
    return out


def patchify3d(x: torch.Tensor, patch) -> torch.Tensor:
    """[B][C][F][H][W] -> tokens [B*f*h*w][C*pt*ph*pw] (feature order c, dt, dh, dw = Conv3d weight order)."""
    _req(x, "x")
    B, Cc, Fr, H, W_ = x.shape
    pt, ph, pw = patch
    if not x.is_contiguous():
        raise ValueError("patchify3d: contiguous input required")
    tok = torch.empty((B * (Fr // pt) * (H // ph) * (W_ // pw), Cc * pt * ph * pw), device=x.device, dtype=bf16)
    L.check(L.load().da_patchify3d_bf16(x.data_ptr(), tok.data_ptr(), B, Cc, Fr, H, W_, pt, ph, pw, _stream()),
            "da_patchify3d_bf16")
    return tok


def unpatchify3d(tok: torch.Tensor, shape, patch) -> torch.Tensor:
    """tokens [B*f*h*w][pt*ph*pw*C] (feature order dt, dh, dw, c) -> [B][C][F][H][W]."""
    _req(tok, "tok")
    B, Cc, Fr, H, W_ = shape
    pt, ph, pw = patch
    if not tok.is_contiguous() or tok.numel() != B * Cc * Fr * H * W_:
        raise ValueError("unpatchify3d: token buffer does not match the target shape")
    x = torch.empty((B, Cc, Fr, H, W_), device=tok.device, dtype=bf16)
    L.check(L.load().da_unpatchify3d_bf16(tok.data_ptr(), x.data_ptr(), B, Cc, Fr, H, W_, pt, ph, pw, _stream()),
            "da_unpatchify3d_bf16")
    return x


def conv_thin_in(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], *, ksize: int, in_nchw: bool,
                 in_div: float = 1.0, in_add: float = 0.0) -> torch.Tensor:
    """Conv2d with Cin <= 16.  x: NCHW [B][Cin][H][W] or NHWC; w: [Cout][k*k*Cin]; returns NHWC [B][H][W][Cout]."""
    _req(x, "x"), _req(w, "w")
    if not x.is_contiguous():
        raise ValueError("conv_thin_in: contiguous input required")
    if in_nchw:
        B, Cin, H, W_ = x.shape
    else:
        B, H, W_, Cin = x.shape
    Cout = w.shape[0]
    y = torch.empty((B, H, W_, Cout), device=x.device, dtype=bf16)
    L.check(L.load().da_conv_thin_in_bf16(x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), B, H, W_, Cin, Cout,
                                          ksize, int(in_nchw), in_div, in_add, _stream()), "da_conv_thin_in_bf16")
    return y


THIN_OUT_PAD = 16     # output channels of a thin-output conv on the implicit-GEMM path (one 16-column MFMA tile)
_thin_out_cache = {}


def pad_thin_out(w: torch.Tensor, bias: Optional[torch.Tensor]):
    """[Cout][K] (+ bias) zero-padded to THIN_OUT_PAD output channels, cached on the packed weight's STORAGE (models pack once;
    the padded copy must exist before a HIP-graph capture, which the pipelines' warm-up pass guarantees).  An in-place edit of the
    weight / bias (`fuse_lora` / `unfuse_lora` re-pack in place, loading.py) bumps their `_version`: the padded copies are then
    REFRESHED IN PLACE -- same addresses -- so that a captured graph or plan, which holds them by raw pointer, replays with the
    edited weight (round 4 made a new entry per version: the graph kept replaying the old copy and every fuse / unfuse cycle leaked
    one entry)."""
    key = (w.data_ptr(), tuple(w.shape), bias.data_ptr() if bias is not None else 0)
    ver = (w._version if not w.is_inference() else -1, (bias._version if not bias.is_inference() else -1) if bias is not None else 0)
    ent = _thin_out_cache.get(key)
    if ent is None:
        wp = torch.zeros((THIN_OUT_PAD, w.shape[1]), device=w.device, dtype=w.dtype)
        wp[: w.shape[0]] = w
        bp = None
        if bias is not None:
            bp = torch.zeros((THIN_OUT_PAD,), device=w.device, dtype=bias.dtype)
            bp[: w.shape[0]] = bias
        # Entries are never evicted: a captured HIP graph references the padded copies by raw pointer only (a few KiB per
        # thin-output conv, one or two per model).
        ent = _thin_out_cache[key] = [wp, bp, w, bias, ver]      # keeps `w` / `bias` alive: their addresses are the key
    elif ent[4] != ver:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("pad_thin_out: a weight was edited in place since its padded copy was made; run one eager step before "
                               "capturing (the pipelines' warm-up pass does)")
        ent[0][: w.shape[0]].copy_(w)
        if bias is not None:
            ent[1][: w.shape[0]].copy_(bias)
        ent[4] = ver
    return ent[0], ent[1]


def refresh_thin_out(tensors) -> int:
    """Refresh (in place) the padded copies of :func:`pad_thin_out` whose weight or bias is one of ``tensors`` -- called after an
    in-place re-pack (loading._repack_in_place) so that captured graphs see the edit; returns how many were refreshed."""
    ptrs = {t.data_ptr() for t in tensors}
    n = 0
    for ent in _thin_out_cache.values():
        wp, bp, w, bias, _ = ent
        if w.data_ptr() in ptrs or (bias is not None and bias.data_ptr() in ptrs):
            wp[: w.shape[0]].copy_(w)
            if bias is not None:
                bp[: w.shape[0]].copy_(bias)
            ent[4] = (w._version if not w.is_inference() else -1, (bias._version if not bias.is_inference() else -1) if bias is not None else 0)
            n += 1
    return n


def conv_thin_out(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], *, out_f32: bool = False,
                  postprocess: Optional[str] = None) -> torch.Tensor:
    """Conv2d 3x3 with small Cout.  x: NHWC; w: [Cout][9*Cin]; returns NCHW [B][Cout][H][W].

    Where the input has whole K slices (Cin % 64 == 0) and enough pixels to fill the chip, the conv runs on the MFMA
    implicit-GEMM kernel with its output channels zero-padded to 16 (one 16-column tile; the padded columns cost nothing that
    matters: the launch is bound by reading the input once) and a one-pass kernel lays the real channels out as NCHW planes:
    VAE conv_out 128 -> 3 at 1024 x 1024, 2.19 ms -> ~0.15 ms; U-Net conv_out 320 -> 4, 79 us -> ~20 us.  The one-thread-per-
    pixel kernel remains for ragged channel counts, tiny images and fp32 output."""
    _req(x, "x"), _req(w, "w")
    B, H, W_, Cin = x.shape
    Cout = w.shape[0]
    if postprocess is not None and (out_f32 or postprocess not in _PP_MODES):
        raise ValueError(f"conv_thin_out: postprocess={postprocess!r} (use 'pt', 'np' or 'uint8' on the bf16 result)")
    if not out_f32 and Cin % 64 == 0 and Cout <= 8 and B * H * W_ >= 4096 and w.shape[1] == 9 * Cin:
        wp, bp = pad_thin_out(w, bias)
        y16 = conv2d_nhwc(x, wp, bp, ksize=3)
        if postprocess is not None and Cout <= 4:
            # VaeImageProcessor.postprocess as the epilogue of the layout pass: the decoded image is written once
            mode = _PP_MODES[postprocess]
            shape = (B, Cout, H, W_) if mode == 0 else (B, H, W_, Cout)
            y = torch.empty(shape, device=x.device, dtype=torch.uint8 if mode == 2 else torch.float32)
            L.check(L.load().da_nhwc_take_postprocess(y16.data_ptr(), y.data_ptr(), B, H * W_, THIN_OUT_PAD, Cout, mode, _stream()),
                    "da_nhwc_take_postprocess")
            return y
        y = torch.empty((B, Cout, H, W_), device=x.device, dtype=bf16)
        L.check(L.load().da_nhwc_take_nchw_bf16(y16.data_ptr(), y.data_ptr(), B, H * W_, THIN_OUT_PAD, Cout, _stream()),
                "da_nhwc_take_nchw_bf16")
        return y if postprocess is None else image_postprocess(y, postprocess)
    y = torch.empty((B, Cout, H, W_), device=x.device, dtype=torch.float32 if out_f32 else bf16)
    L.check(L.load().da_conv_thin_out_bf16(x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), B, H, W_, Cin, Cout,
                                           int(out_f32), _stream()), "da_conv_thin_out_bf16")
    return y if postprocess is None else image_postprocess(y, postprocess)
