#!/usr/bin/env python
"""A handful of (shape, variant) launches of the K2 / K1 GEMM family for rocprofv3 --pmc passes: each combination is launched 3
times (after one warm launch) with the operands evicted in between, in a fixed order written to argv[1] so the per-dispatch
counters can be joined (tools/pmc_k2_report.py)."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import _lib as L, ops  # noqa: E402

bf16 = torch.bfloat16


def main():
    flush = torch.empty(320 << 20, dtype=torch.uint8, device="cuda")
    cases = [  # name, M, N, K, act, residual, [(tile, staging)]
        ("ff_down1280", 2048, 1280, 5120, 0, True, [(L.TILE_K2_128x80, 2), (L.TILE_K2_128x80, 1), (L.TILE_128x64, 2)]),
        ("to_out1280", 2048, 1280, 1280, 0, True, [(L.TILE_K2_128x80, 2), (L.TILE_128x64, 2)]),
        ("geglu1280", 2048, 10240, 1280, L.ACT_GEGLU, False, [(L.TILE_K1_128x256, 2), (L.TILE_128x128, 1), (L.TILE_K2_128x128, 1)]),
        ("geglu-shape plain", 2048, 10240, 1280, 0, False, [(L.TILE_K1_128x320, 1), (L.TILE_K1_128x256, 2), (L.TILE_128x128, 1)]),
        ("qk1280", 2048, 2560, 1280, 0, False, [(L.TILE_K2_128x160, 1)]),
    ]
    manifest = []
    for name, M, N, K, act, res, variants in cases:
        x = (torch.randn((M, K), device="cuda")).to(bf16)
        w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(bf16)
        b = torch.randn((N,), device="cuda").to(bf16)
        if act:
            w, b = ops.pack_geglu(w, b)
        r = torch.randn((M, N), device="cuda").to(bf16) if res else None
        for tile, st in variants:
            for i in range(4):
                flush.zero_()
                ops.linear(x, w, bias=b, residual=r, act=act, tile=tile, staging=st)
            manifest.append({"name": name, "M": M, "N": N, "K": K, "tile": L.TILE_NAMES[tile], "staging": st, "reps": 4})
        torch.cuda.synchronize()
    # flash attention at SDXL's dominant shape (B 2, H 20, S 1024, D 64), default variant: same counters
    B, H, S, D = 2, 20, 1024, 64
    inner = H * D
    q, k, vt = (torch.randn((B * S, inner), device="cuda").to(bf16) for _ in range(2)), None, None
    q, k = q
    vt = torch.randn((inner, B * S), device="cuda").to(bf16)
    for i in range(4):
        flush.zero_()
        ops.attention(q, k, vt, B=B, H=H, D=D, Sq=S, Skv=S, Skv_alloc=S, q_row_stride=inner, k_row_stride=inner,
                      q_batch_stride=S * inner, k_batch_stride=S * inner, vt_ld=B * S, vt_batch_stride=S)
    torch.cuda.synchronize()
    Path(sys.argv[1]).write_text(json.dumps(manifest))


if __name__ == "__main__":
    main()
