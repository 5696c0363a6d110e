#!/usr/bin/env python
"""Round 6: flash attention with / without the key-split tail at the BASELINE shapes, chained launches from a HIP graph, off / on /
off / on in ONE process (the split is a per-launch argument).  usage: bench_attn_r6.py out.jsonl (appends)"""
import ctypes as C
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import _lib as L, ops  # noqa: E402

bf16 = torch.bfloat16
g = torch.Generator("cpu").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g).to(bf16).to("cuda")  # noqa: E731


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


def plan(B, H, S, D, kv):
    p = L.AttentionParams()
    p.B, p.H, p.Sq, p.Skv, p.Skv_alloc, p.D, p.kv_split = B, H, S, S, S, D, kv
    f, t, s = C.c_int(), C.c_int(), C.c_int()
    L.load().da_attention_split_plan(C.byref(p), C.byref(f), C.byref(t), C.byref(s))
    return f.value, t.value, s.value


if __name__ == "__main__":
    out = open(sys.argv[1], "a") if len(sys.argv) > 1 else None
    shapes = ((2, 20, 1024, 64, "SDXL 32x32 level"), (2, 10, 4096, 64, "SDXL 64x64 level"), (1, 24, 4608, 128, "Flux joint"),
              (2, 8, 4096, 64, "SD1.5 64x64 level (head 40 padded)"), (2, 12, 32760, 128, "Wan self"))
    for (B, H, S, D, what) in shapes:
        inner = H * D
        qk = rnd(B * S, 2 * inner)
        vt = rnd(inner, B * S)
        for kv in (1, 0, 1, 0, 2, 3, 4, 6, 8):
            if kv > 1 and plan(B, H, S, D, kv)[2] != kv:
                continue
            if S > 8192 and kv > 1:
                continue
            fn = lambda: ops.attention(qk, qk[:, inner:], vt, B=B, H=H, D=D, Sq=S, Skv=S, Skv_alloc=S, q_row_stride=2 * inner,  # noqa: E731
                                       k_row_stride=2 * inner, q_batch_stride=S * 2 * inner, k_batch_stride=S * 2 * inner, vt_ld=B * S,
                                       vt_batch_stride=S, kv_split=kv)
            us = graph_us(fn, n=20 if S <= 8192 else 3)
            fl = 4.0 * B * H * S * S * D
            f, t, s = plan(B, H, S, D, kv)
            rec = {"op": "attention v2", "what": what, "B": B, "H": H, "S": S, "D": D, "kv_split": kv, "plan": {"whole": f, "tail": t, "units": s},
                   "us": round(us, 1), "tflops": round(fl / us / 1e6, 0), "frac": round(fl / us / 1e6 / 2500, 3)}
            print(json.dumps(rec), flush=True)
            if out:
                out.write(json.dumps(rec) + "\n")
