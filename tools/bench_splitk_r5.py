#!/usr/bin/env python
"""Round 5: in-launch split-K on the large first-family tiles for SDXL's deep-K, 256-tile shapes -- the 3 x 3 convs of the 32 x 32
level (M 2048 x N 1280, K 5.8 k ... 23 k: one 128 x 80 tile per CU whose K loop is bound by the L1 -> LDS fill) and
FF-down (M 2048 x N 1280 x K 5120).  A 256 x 128 / 256 x 256 tile stages half the bytes per flop; split 3 / 6 puts it on ~240 CUs.
Each candidate: a chain of `layers` launches with distinct weights replayed from one HIP graph (us per launch).
usage: bench_splitk_r5.py out.jsonl"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import _lib as L, ops  # noqa: E402

bf16 = torch.bfloat16
DEV = "cuda"


def rnd(shape, seed, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(bf16).to(DEV)


def graph_us(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(4):
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3


CANDS = [("table", None, None, None)] + \
        [(f"{L.TILE_NAMES[t]} st{s} split{k}", t, s, k) for t in (L.TILE_256x128, L.TILE_128x256, L.TILE_256x256, L.TILE_128x128)
         for s in (L.STAGE_LDS_DIRECT, L.STAGE_LDS_DIRECT3) for k in (1, 2, 3, 4, 6)]


def main(out):
    recs = []
    nl = 6
    shapes = [("conv3", 2, 32, 1280, 0, 1280), ("conv3", 2, 32, 1280, 1280, 1280), ("conv3", 2, 32, 1280, 640, 1280), ("conv3", 2, 32, 640, 0, 1280),
              ("conv3", 2, 64, 640, 0, 640), ("conv3", 2, 64, 1280, 640, 640), ("conv3", 2, 128, 320, 0, 320), ("lin", 2048, 5120, 1280, 0, 0),
              ("lin", 2048, 1280, 1280, 0, 0)]
    for sh in shapes:
        if sh[0] == "conv3":
            _, B, H, C1, C2, N = sh
            xs = rnd((B, H, H, C1), 1)
            x2 = rnd((B, H, H, C2), 2) if C2 else None
            ws = [rnd((N, 9 * (C1 + C2)), 10 + i, (9 * (C1 + C2)) ** -0.5) for i in range(nl)]
            bs = rnd((N,), 3)
            res = rnd((B, H, H, N), 4)
            name = f"conv3 M{B * H * H} N{N} C{C1}+{C2}"
            flop = 2.0 * B * H * H * N * 9 * (C1 + C2)

            def mk(t, s, k):
                def run():
                    for w in ws:
                        ops.conv2d_nhwc(xs, w, bs, ksize=3, x2=x2, residual=res, tile=t, staging=s, split_k=k)
                return run
        else:
            _, M, K, N, _, _ = sh
            xs = rnd((M, K), 1)
            ws = [rnd((N, K), 10 + i, K ** -0.5) for i in range(nl)]
            bs = rnd((N,), 3)
            res = rnd((M, N), 4)
            name = f"lin M{M} N{N} K{K}"
            flop = 2.0 * M * N * K

            def mk(t, s, k):
                def run():
                    for w in ws:
                        ops.linear(xs, w, bs, residual=res, tile=t, staging=s, split_k=k)
                return run
        rows = []
        for label, t, s, k in CANDS:
            try:
                us = graph_us(mk(t, s, k)) / nl
            except Exception as e:   # a variant the library refuses for this shape
                continue
            rows.append((us, label))
        rows.sort()
        base = next(u for u, lb in rows if lb == "table")
        rec = {"shape": name, "gflop": round(flop / 1e9, 1), "table_us": round(base, 1), "table_tflops": round(flop / base / 1e6, 0),
               "best": [{"variant": lb, "us": round(u, 1), "tflops": round(flop / u / 1e6, 0)} for u, lb in rows[:4]]}
        print(json.dumps(rec), flush=True)
        recs.append(rec)
    Path(out).write_text("".join(json.dumps(r) + "\n" for r in recs))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else str(ROOT / "gpurun_out" / "splitk_r5.jsonl"))
