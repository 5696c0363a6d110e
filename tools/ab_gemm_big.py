#!/usr/bin/env python
"""Same-box A/B helper: large nn.Linear shapes (Flux / square) timed per (tile, staging) with whatever tree this file is
run from (the script only uses ops.linear(x, w, bias, residual=, tile=, staging=), unchanged since round 1)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path.cwd()))
import os  # noqa: E402

from diffusers_amd import _lib, ops  # noqa: E402

if os.environ.get("DA_AB_LIB"):
    _lib.LIB_PATH = Path(os.environ["DA_AB_LIB"])

bf16 = torch.bfloat16


def timeit(fn, flush, iters=8):
    fn()
    ts = []
    for _ in range(iters):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts)


def main():
    flush = torch.empty(320 << 20, dtype=torch.uint8, device="cuda")
    shapes = [("flux qkv", 4608, 9216, 3072, False), ("flux ff_up", 4608, 12288, 3072, False),
              ("flux ff_down", 4096, 3072, 12288, True), ("sdxl geglu", 2048, 10240, 1280, False), ("sdxl to_q", 2048, 1280, 1280, False)]
    variants = [(5, 2), (1, 1), (3, 2)]   # 256x128/3 slots, 128x128/2, 128x64/3
    for name, M, N, K, res in shapes:
        x = (torch.randn((M, K), device="cuda")).to(bf16)
        w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(bf16)
        b = torch.randn((N,), device="cuda").to(bf16)
        r = torch.randn((M, N), device="cuda").to(bf16) if res else None
        rec = {"name": name}
        for tile, st in variants:
            us = timeit(lambda: ops.linear(x, w, b, residual=r, tile=tile, staging=st), flush)
            rec[f"t{tile}s{st}"] = [round(us, 1), round(2.0 * M * N * K / us / 1e6)]
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
