#!/usr/bin/env python
"""A handful of (shape, variant) GEMM launches for PMC passes (rocprofv3 --pmc ...): each combination is launched
3 times with the operands evicted from the caches in between, so per-dispatch counters can be read off the CSV."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import _lib as L, ops  # noqa: E402

bf16 = torch.bfloat16


def main():
    flush = torch.empty(320 << 20, dtype=torch.uint8, device="cuda")
    cases = [  # name, M, N, K, act, [(tile, staging)]
        ("to_out1280", 2048, 1280, 1280, 0, [(1, 1), (1, 3), (4, 1), (4, 5), (3, 4)]),
        ("geglu1280", 2048, 10240, 1280, L.ACT_GEGLU, [(1, 1), (1, 3), (5, 2), (7, 1)]),
        ("square4096", 4096, 4096, 4096, 0, [(1, 1), (1, 3), (7, 1)]),
    ]
    for name, M, N, K, act, variants in cases:
        x = (torch.randn((M, K), device="cuda")).to(bf16)
        w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(bf16)
        for tile, st in variants:
            for _ in range(3):
                flush.zero_()
                ops.linear(x, w, act=act, tile=tile, staging=st)
        torch.cuda.synchronize()
        print(name, "done", flush=True)


if __name__ == "__main__":
    main()
