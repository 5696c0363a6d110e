#!/usr/bin/env python
"""K2 GEMM family (csrc/gemm2_kernel.cuh) against the first family on the SDXL shapes: per shape, the shipped table's variant
and every K2 (tile, ring depth) variant, HIP events on the launch stream, three regimes -- cold (320 MiB memset evicts L2 and the
Infinity Cache before each launch), warm (back-to-back, separately timed) and chain (40 launches back to back, launch gaps
included, event overhead amortised).  One JSON object per line."""
from __future__ import annotations

import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import _lib as L  # noqa: E402
from diffusers_amd import ops  # noqa: E402
from tools.ceiling_table import chain_us, rnd, timeit  # noqa: E402
import tools.ceiling_table as CT  # noqa: E402


def variants(geglu=False, conv=False):
    for t in range(L.FIRST_K2_TILE, len(L.TILE_NAMES)):
        if geglu and t not in (L.TILE_K2_128x128, L.TILE_K1_256x128, L.TILE_K1_128x256, L.TILE_K1_256x256, L.TILE_K1_256x320, L.TILE_K1_128x320):
            continue
        if conv and t in (L.TILE_K2_80x128, L.TILE_K1_256x256, L.TILE_K1_256x320):
            continue
        for st in (L.STAGE_LDS_DIRECT, L.STAGE_LDS_DIRECT3, L.STAGE_PINGPONG, L.STAGE_PINGPONG3):
            yield t, st


def measure(fn):
    fn()
    cold, _ = timeit(fn, iters=8, warm=1)
    warm, _ = timeit(fn, iters=8, warm=1, flush=False)
    return round(cold, 1), round(warm, 1), round(chain_us(fn, 30), 1)


def main():
    CT.FLUSH = torch.empty(320 << 20, dtype=torch.uint8, device="cuda")
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    lin = [("to_out 1280 +b+r", 2048, 1280, 1280, True, True, 0), ("to_q 1280", 2048, 1280, 1280, False, False, 0),
           ("qk 1280", 2048, 2560, 1280, False, False, 0), ("vT 1280", 1280, 2048, 1280, False, False, 0),
           ("geglu 1280", 2048, 10240, 1280, True, False, L.ACT_GEGLU), ("geglu-shape plain 1280", 2048, 10240, 1280, True, False, 0),
           ("ff_down 1280 +b+r", 2048, 1280, 5120, True, True, 0),
           ("to_out 640 +b+r", 8192, 640, 640, True, True, 0), ("geglu 640", 8192, 5120, 640, True, False, L.ACT_GEGLU),
           ("ff_down 640 +b+r", 8192, 640, 2560, True, True, 0), ("qk 640", 8192, 1280, 640, False, False, 0),
           ("flux qkv 3072", 4608, 3072, 3072, True, False, 0), ("flux mlp 12288", 4608, 12288, 3072, True, False, L.ACT_GELU_TANH),
           ("sd15 to_out 320", 8192, 320, 320, True, True, 0)]
    for name, M, N, K, hb, hr, act in lin:
        if only and only not in name:
            continue
        x, w = rnd((M, K)), rnd((N, K), K ** -0.5)
        b = rnd((N,)) if hb else None
        geglu = act == L.ACT_GEGLU
        if geglu:
            w, b = ops.pack_geglu(w, b)
        r = rnd((M, N)) if hr else None
        base = lambda: ops.linear(x, w, bias=b, residual=r, act=act)  # noqa: E731
        rec = {"op": "linear", "name": name, "M": M, "N": N, "K": K, "gflop": round(2e-9 * M * N * K, 2), "table": measure(base)}
        ref = base()
        for t, st in variants(geglu=geglu):
            fn = lambda: ops.linear(x, w, bias=b, residual=r, act=act, tile=t, staging=st)  # noqa: E731
            try:
                y = fn()
            except RuntimeError:
                continue
            d = float(((y.float() - ref.float()).abs() / ref.float().abs().clamp_min(2.0 ** -10)).max())
            rec[f"{L.TILE_NAMES[t]}/{st}"] = measure(fn) + (round(d, 4),)
        best = min((v[2], k) for k, v in rec.items() if isinstance(v, tuple) and k != "table")
        rec["best_k2"] = [best[1], best[0]]
        rec["speedup_chain"] = round(rec["table"][2] / best[0], 3)
        print(json.dumps(rec), flush=True)

    conv = [("conv3 320 @128^2", 2, 128, 128, 320, 0, 320), ("conv3 640 @64^2", 2, 64, 64, 640, 0, 640),
            ("conv3 1280 @32^2", 2, 32, 32, 1280, 0, 1280), ("conv3 1280+640->1280 @32^2", 2, 32, 32, 1280, 640, 1280),
            ("conv3 640+320->640 @64^2", 2, 64, 64, 640, 320, 640), ("vae conv3 512 @128^2", 1, 128, 128, 512, 0, 512),
            ("vae conv3 256 @512^2", 1, 512, 512, 256, 0, 256), ("vae conv3 128 @1024^2", 1, 1024, 1024, 128, 0, 128)]
    for name, B, H, W, C1, C2, Co in conv:
        if only and only not in name:
            continue
        x = rnd((B, H, W, C1))
        x2 = rnd((B, H, W, C2)) if C2 else None
        K = 9 * (C1 + C2)
        w, b = rnd((Co, K), K ** -0.5), rnd((Co,))
        base = lambda: ops.conv2d_nhwc(x, w, b, ksize=3, x2=x2)  # noqa: E731
        rec = {"op": "conv", "name": name, "M": B * H * W, "N": Co, "K": K, "gflop": round(2e-9 * B * H * W * Co * K, 2),
               "table": measure(base)}
        ref = base()
        for t, st in variants(conv=True):
            fn = lambda: ops.conv2d_nhwc(x, w, b, ksize=3, x2=x2, tile=t, staging=st)  # noqa: E731
            try:
                y = fn()
            except RuntimeError:
                continue
            d = float(((y.float() - ref.float()).abs() / ref.float().abs().clamp_min(2.0 ** -10)).max())
            rec[f"{L.TILE_NAMES[t]}/{st}"] = measure(fn) + (round(d, 4),)
        best = min((v[2], k) for k, v in rec.items() if isinstance(v, tuple) and k != "table")
        rec["best_k2"] = [best[1], best[0]]
        rec["speedup_chain"] = round(rec["table"][2] / best[0], 3)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
