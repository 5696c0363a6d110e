#!/usr/bin/env python
"""Does da_gemm_params.prefetch pay?  A chain of SDXL-shaped projections over MORE weights than the memory-side cache holds (every
weight is met cold, as inside a denoising step), each launch optionally reading the NEXT launch's weight behind its K loop.
Per-launch time of the chain (HIP events around the whole chain), hint off / on, ABAB."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import ops  # noqa: E402
from tools.ceiling_table import rnd  # noqa: E402


def chain(x, ws, b, r, hint, reps=2):
    n = len(ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for i in range(n):
            ops.linear(x, ws[i], bias=b, residual=r, prefetch=ws[(i + 1) % n] if hint else None)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)


def main():
    for name, M, N, K, nw in (("to_out 1280", 2048, 1280, 1280, 120), ("ff_down 1280", 2048, 1280, 5120, 40), ("to_out 640", 8192, 640, 640, 400)):
        x, b, r = rnd((M, K)), rnd((N,)), rnd((M, N))
        ws = [rnd((N, K), K ** -0.5) for _ in range(nw)]       # nw * N * K * 2 bytes > 256 MiB
        ops.linear(x, ws[0], bias=b, residual=r)                # tune the shape if needed
        rec = {"name": name, "weights_mb": round(nw * N * K * 2 / 2 ** 20), "us_per_launch": {}}
        for rep in range(2):
            for hint in (False, True):
                rec["us_per_launch"][f"{'hint' if hint else 'plain'}{rep}"] = round(chain(x, ws, b, r, hint), 2)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
