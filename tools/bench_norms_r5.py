#!/usr/bin/env python
"""Round 5: GroupNorm as one launch (gn_fused_kernel) against the two-kernel form at the shapes of an SDXL step and of an SD1.5 step.
Chained microseconds per GroupNorm (HIP-graph replay of back-to-back launches).  usage: bench_norms_r5.py out.jsonl"""
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import ops  # noqa: E402
from tools.ceiling_table import rnd  # noqa: E402


def chain_us(fn, n=30):
    """Microseconds per call of n back-to-back calls replayed from ONE HIP graph (as the step replays them): an eager Python loop
    is host-bound for kernels this short."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (4 * n)

# (B, HW, C1, C2, count per step)
SDXL = [(2, 16384, 320, 0, 4), (2, 16384, 640, 320, 1), (2, 16384, 320, 320, 2), (2, 16384, 640, 0, 0), (2, 4096, 320, 0, 1), (2, 4096, 640, 0, 7),
        (2, 4096, 1280, 640, 1), (2, 4096, 640, 640, 1), (2, 4096, 640, 320, 1), (2, 4096, 1280, 0, 0), (2, 1024, 640, 0, 1), (2, 1024, 1280, 0, 14),
        (2, 1024, 1280, 1280, 2), (2, 1024, 1280, 640, 1)]
SD15 = [(2, 4096, 320, 0, 12), (2, 1024, 320, 0, 1), (2, 1024, 640, 0, 10), (2, 256, 640, 0, 1), (2, 256, 1280, 0, 10), (2, 64, 1280, 0, 10),
        (2, 64, 2560, 0, 3), (2, 256, 2560, 0, 2), (2, 256, 1920, 0, 1), (2, 1024, 1920, 0, 1), (2, 1024, 1280, 0, 3), (2, 1024, 960, 0, 1),
        (2, 4096, 960, 0, 1), (2, 4096, 640, 0, 3)]


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    for name, shapes in (("sdxl", SDXL), ("sd15", SD15)):
        tot = {"two": 0.0, "one": 0.0, "one_l2": 0.0}
        for B, HW, C1, C2, cnt in shapes:
            C = C1 + C2
            x = rnd((B, HW, C1))
            x2 = rnd((B, HW, C2)) if C2 else None
            g, b = rnd((C,)), rnd((C,))
            fn = lambda: ops.group_norm_nhwc(x, g, b, 32, 1e-5, silu=True, x2=x2)  # noqa: E731
            rec = {"model": name, "B": B, "HW": HW, "C": f"{C1}+{C2}", "per_step": cnt}
            for key, env in (("two", {"DA_GN_FUSED": "0"}), ("one", {"DA_GN_FUSED": "1", "DA_GN_FUSED_KB": "256"}),
                             ("one_l2", {"DA_GN_FUSED": "1", "DA_GN_FUSED_KB": "1100"})):
                os.environ.update(env)
                us = min(chain_us(fn, 30) for _ in range(3))
                rec[key + "_us"] = round(us, 1)
                tot[key] += us * cnt
            print(json.dumps(rec), flush=True)
            if out:
                out.write(json.dumps(rec) + "\n")
        rec = {"model": name, "op": "sum over one step's GroupNorms (us)", **{k: round(v, 1) for k, v in tot.items()}}
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
