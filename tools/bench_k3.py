#!/usr/bin/env python
"""Round 5: the eight-phase 256 x 256 tile (csrc/gemm3.hip, k3:256x256) against the shipped table's variant and the K1 tiles on the
large nn.Linear shapes (Flux, Wan, SDXL's GEGLU / fused QKV): HIP-graph chains of 20 launches (launch gaps included), random operands.
usage: bench_k3.py out.jsonl [name filter]"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import _lib as L  # noqa: E402
from diffusers_amd import ops  # noqa: E402

bf16 = torch.bfloat16
g = torch.Generator("cpu").manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(s, generator=g) * scale).to(bf16).to("cuda")  # noqa: E731


def graph_us(fn, n=20):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        e0.record()
        gr.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3


SHAPES = [  # name, M, N, K, act, residual
    ("flux proj_mlp", 4608, 12288, 3072, L.ACT_GELU_TANH, False),
    ("flux proj_mlp shape, plain epilogue", 4608, 12288, 3072, L.ACT_NONE, False),
    ("flux proj_mlp shape, 4 exact rounds (M 5120), plain epilogue", 5120, 12800, 3072, L.ACT_NONE, False),
    ("flux qk", 4608, 6144, 3072, L.ACT_NONE, False),
    ("flux to_out", 4608, 3072, 3072, L.ACT_NONE, True),
    ("flux proj_out", 4608, 3072, 15360, L.ACT_NONE, True),
    ("flux ff_down", 4096, 3072, 12288, L.ACT_NONE, True),
    ("wan qk", 32760, 10240, 5120, L.ACT_NONE, False),
    ("wan ffn up", 32760, 13824, 5120, L.ACT_GELU_TANH, False),
    ("wan ffn down", 32760, 5120, 13824, L.ACT_NONE, True),
    ("sdxl geglu 1280", 2048, 10240, 1280, L.ACT_GEGLU, False),
    ("sdxl geglu 640", 8192, 5120, 640, L.ACT_GEGLU, False),
    ("sdxl qkv 1280", 2048, 3840, 1280, L.ACT_NONE, False),
    ("sdxl qkv 640", 8192, 1920, 640, L.ACT_NONE, False),
    ("sdxl ff_down 640", 8192, 640, 2560, L.ACT_NONE, True),
    ("square 4096", 4096, 4096, 4096, L.ACT_NONE, False),
    ("square 8192", 8192, 8192, 8192, L.ACT_NONE, False),
]


def main():
    out = open(sys.argv[1], "a") if len(sys.argv) > 1 else None
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, M, N, K, act, has_r in SHAPES:
        if only and only not in name:
            continue
        x, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
        if act == L.ACT_GEGLU:
            w, b = ops.pack_geglu(w, b)
        r = rnd(M, N) if has_r else None
        fl = 2.0 * M * N * K
        rec = {"op": "linear", "name": name, "M": M, "N": N, "K": K, "gflop": round(fl * 1e-9, 1)}
        ref = ops.linear(x, w, b, act=act, residual=r, tile=L.TILE_K1_256x256, staging=L.STAGE_LDS_DIRECT)
        cands = [("table", None, None), ("k3:256x256", L.TILE_K3_256x256, L.STAGE_LDS_DIRECT), ("k3:256x320", L.TILE_K3_256x320, L.STAGE_LDS_DIRECT), ("k1:256x256", L.TILE_K1_256x256, L.STAGE_LDS_DIRECT),
                 ("k1:256x128/3", L.TILE_K1_256x128, L.STAGE_LDS_DIRECT3), ("k1:128x256/3", L.TILE_K1_128x256, L.STAGE_LDS_DIRECT3),
                 ("k1:128x320", L.TILE_K1_128x320, L.STAGE_LDS_DIRECT), ("k1:256x320", L.TILE_K1_256x320, L.STAGE_LDS_DIRECT)]
        for label, t, st in cands:
            fn = lambda: ops.linear(x, w, b, act=act, residual=r, tile=t, staging=st)  # noqa: E731
            try:
                y = fn()
            except RuntimeError:
                continue
            us = graph_us(fn)
            rec[label] = [round(us, 1), round(fl / us * 1e-6), bool(torch.equal(y, ref)) if t is not None else None]
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
            out.flush()
        del x, w, b, r, ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
