"""Per-shape kernel selection for ``da_gemm_bf16`` (Linear + implicit-GEMM Conv2d).

The reference leaves this choice to the vendor libraries behind ``F.linear`` / ``F.conv2d`` (hipBLASLt heuristics, MIOpen
find-db).  Here the C ABI exposes every (tile, staging) variant of the one hand-written kernel plus ``da_gemm_tune``,
which times them on the caller's stream; this module owns the resulting table:

  * ``lookup(p)``   -> (tile, staging) for a problem, from the table, else tuned live (outside graph capture), else the
                      library's untuned heuristic (tile = AUTO).
  * the table is keyed by problem shape only, is plain JSON (``tuned/gfx950.json`` ships the shapes of the BASELINE
    configs measured on an MI355X) and can be extended / saved with ``save()``.

Within one kernel family all variants produce bit-identical outputs (same K order, same MFMA); the two families (csrc/gemm_kernel.cuh
and the K2 family of csrc/gemm2_kernel.cuh) differ in the fp32 summation order of K, i.e. in the last bit before an output is
rounded to bf16 -- ``FAMILY`` restricts live tuning to one of them.
Env: ``DIFFUSERS_AMD_TUNE=0`` disables live tuning, ``DIFFUSERS_AMD_TUNE_DB=<path>`` overrides the table location,
``DIFFUSERS_AMD_TUNE_SAVE=<path>`` writes the (extended) table there at interpreter exit.
"""
from __future__ import annotations

import atexit
import ctypes as C
import json
import os
from pathlib import Path
from typing import Dict, Optional, Tuple

import torch

from . import _lib as L

_PKG = Path(__file__).resolve().parent
DB_PATH = Path(os.environ.get("DIFFUSERS_AMD_TUNE_DB", _PKG / "tuned" / "gfx950.json"))
LIVE = os.environ.get("DIFFUSERS_AMD_TUNE", "1") != "0"
ITERS = 3
FLUSH_BYTES = 320 << 20  # > 256 MiB Infinity Cache: operands are timed at HBM latency, as inside the denoising loop

_table: Dict[str, Tuple[int, int, float, int]] = {}   # key -> (tile, staging, microseconds, split_k)
LIVE_COUNT = 0     # problems tuned live in this process (bench.py reports it: a tuning pass on 8 ranks at once would cost scaling)
_loaded = False
_dirty = False
_scratch = {}
_session_tuned = set()   # unpaired problems timed by THIS process


def _m_key(m: int) -> int:
    """Row counts above 2^20 keep their top three bits only: such a problem is thousands of tiles deep, so the best
    variant does not move with the exact count (e.g. the 81 / 80 / 79-frame temporal taps of one video conv)."""
    if m <= (1 << 20):
        return m
    sh = m.bit_length() - 3
    return (m >> sh) << sh


def key_of(p: "L.GemmParams") -> str:
    """Shape key: everything that changes the kernel's work or its memory pattern, nothing that is a pointer."""
    kv = f":v{p.k_valid}" if p.k_valid else ""   # valid K elements per period: changes the work per slice
    if p.conv:
        return (f"conv{p.conv}:M{_m_key(int(p.M))}:N{p.N}:C{p.C1}+{p.C2}:H{p.Hin}x{p.Win}:s{p.stride}:u{p.up}:a{p.act}"
                f":r{int(bool(p.residual))}{kv}")
    return f"lin:M{_m_key(int(p.M))}:N{p.N}:K{p.K}:a{p.act}:f{p.out_f32}:r{int(bool(p.residual))}{kv}"


def _load() -> None:
    global _loaded
    _loaded = True
    if DB_PATH.exists():
        try:
            raw = json.loads(DB_PATH.read_text())
            for k, v in raw.get("entries", {}).items():
                _table[k] = (int(v[0]), int(v[1]), float(v[2]), int(v[3]) if len(v) > 3 else 1)
        except (ValueError, KeyError, TypeError) as e:  # a corrupt table must not break inference
            raise RuntimeError(f"diffusers_amd: unreadable tuning table {DB_PATH}: {e}") from e


def table() -> Dict[str, Tuple[int, int, float, int]]:
    if not _loaded:
        _load()
    return _table


def save(path: Optional[os.PathLike] = None) -> Path:
    path = Path(path) if path is not None else DB_PATH
    path.parent.mkdir(parents=True, exist_ok=True)
    ent = {k: [v[0], v[1], round(v[2], 2)] + ([v[3]] if v[3] > 1 else []) for k, v in sorted(table().items())}
    path.write_text(json.dumps({"arch": "gfx950", "format": "key -> [tile, staging, microseconds(, split_k if > 1)]",
                                "tiles": list(L.TILE_NAMES), "entries": ent}, indent=0))
    return path


# Let the tuner consider split-K variants.  OFF by default: on every SDXL shape that leaves CUs idle the in-launch
# reduction measured SLOWER than the best unsplit variant (profiles/r02b_kernel_experiments.md: 0.59-0.85x; the fp32
# partial-tile hand-off costs more than the idle CUs), and a split factor changes the fp32 summation order.
SPLIT_K = os.environ.get("DIFFUSERS_AMD_SPLITK", "0") == "1"

# Which kernel families compete in live tuning: "all", "1" (csrc/gemm_kernel.cuh only: every variant bit-identical to
# every other, what rounds 1-2 shipped) or "k2" (csrc/gemm2_kernel.cuh only).  The two families differ in the fp32 summation
# order of K ((even slices) + (odd slices) in K2), i.e. in the last bit before the bf16 rounding of an output.
# Default "1": shapes that are NOT in the shipped table (the table itself was measured with "all") are tuned among variants that
# are bit-identical to each other, so the winner of a noisy timing race cannot change a result between runs or between ranks.
FAMILY = os.environ.get("DIFFUSERS_AMD_GEMM_FAMILY", "1")
_FAMILY_CODE = {"all": L.TILE_AUTO, "1": -1, "k2": -2}


def mark_dirty() -> None:
    """An entry was added outside :func:`tune` (ops.qkv_variant): DIFFUSERS_AMD_TUNE_SAVE writes the table at exit."""
    global _dirty, LIVE_COUNT
    _dirty = True
    LIVE_COUNT += 1


def pair_key(pa: "L.GemmParams", pb: "L.GemmParams") -> str:
    return "pair:" + key_of(pa) + "|" + key_of(pb)


def tune(p: "L.GemmParams", stream: int, pair: Optional["L.GemmParams"] = None) -> Tuple[int, int, float, int]:
    """Run da_gemm_tune for this problem (synchronises the stream) and remember the winner.  ``pair``: the two problems
    are timed as ONE launch (da_gemm_pair_bf16).  When ``p`` carries a split-K workspace the split factors 2..24 compete
    with the unsplit variants."""
    global _dirty, LIVE_COUNT
    LIVE_COUNT += 1
    bt, bs, bk, us = C.c_int(0), C.c_int(0), C.c_int(1), C.c_float(0.0)
    dev = torch.cuda.current_device()
    if FLUSH_BYTES and dev not in _scratch:
        _scratch[dev] = torch.empty(FLUSH_BYTES, dtype=torch.uint8, device=f"cuda:{dev}")
    sp = _scratch[dev].data_ptr() if FLUSH_BYTES else None
    # a launch of more than ~2 TFLOP runs for milliseconds: one timed launch per variant is already stable
    iters = 1 if 2.0 * p.M * p.N * p.K > 2e12 else ITERS
    keep_tile, p.tile = p.tile, _FAMILY_CODE[FAMILY]
    try:
        rc = L.load().da_gemm_tune(C.byref(p), C.byref(pair) if pair is not None else None, stream, iters, sp,
                                   FLUSH_BYTES if sp else 0, C.byref(bt), C.byref(bs), C.byref(bk), C.byref(us))
    finally:
        p.tile = keep_tile
    L.check(rc, "da_gemm_tune")
    ent = (bt.value, bs.value, us.value, bk.value)
    if pair is None:
        _session_tuned.add(key_of(p))
    table()[pair_key(p, pair) if pair is not None else key_of(p)] = ent
    _dirty = True
    return ent


def lookup(p: "L.GemmParams", stream: int, inplace: bool = False) -> Tuple[int, int, int]:
    """(tile, staging, split_k) to launch this problem with.  ``inplace``: the output aliases the residual (accumulating
    launch), so the repeated timing launches write to a scratch output instead of accumulating into the caller's tensor."""
    ent = table().get(key_of(p))
    if ent is None:
        if LIVE and not torch.cuda.is_current_stream_capturing():
            if inplace:
                keep = p.C
                tmp = torch.empty(int(p.M) * int(p.ldc), dtype=torch.float32 if p.out_f32 else torch.bfloat16,
                                  device=f"cuda:{torch.cuda.current_device()}")
                p.C = tmp.data_ptr()
                try:
                    ent = tune(p, stream)
                finally:
                    p.C = keep
            else:
                ent = tune(p, stream)
        else:
            return L.TILE_AUTO, L.STAGE_LDS_DIRECT, 1
    return ent[0], ent[1], ent[3]


def lookup_pair(pa: "L.GemmParams", pb: "L.GemmParams", stream: int) -> Optional[Tuple[int, int, float]]:
    """(tile, staging, microseconds) of the paired launch of two nn.Linear problems, or None when it is not known and
    cannot be measured now (graph capture / live tuning off): the caller then launches them separately."""
    ent = table().get(pair_key(pa, pb))
    if ent is None:
        if not (LIVE and not torch.cuda.is_current_stream_capturing()):
            return None
        ent = tune(pa, stream, pair=pb)
        # the caller compares the pair with the two separate launches: measure those in the SAME session (an entry from an
        # older table was timed on older kernels)
        for q in (pa, pb):
            if key_of(q) not in _session_tuned:
                tune(q, stream)
    return ent[0], ent[1], ent[2]


def _save_at_exit() -> None:
    path = os.environ.get("DIFFUSERS_AMD_TUNE_SAVE")
    if path and _dirty:
        save(path)


atexit.register(_save_at_exit)
