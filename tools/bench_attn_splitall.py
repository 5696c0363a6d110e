#!/usr/bin/env python
"""Round 6 (second session): DA_ATTN_SPLIT_ALL=<s> (read once per process: run once per value) -- the keys of EVERY query block of a sparse
D = 64 launch split over s units.  Chained launches from a HIP graph; the output is compared with the unsplit launch's (kv_split = 1)."""
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import ops  # noqa: E402
from tools.bench_attn_r6 import graph_us, plan, rnd  # noqa: E402,F401

mode = os.environ.get("DA_ATTN_SPLIT_ALL", "0")
for (B, H, S, D, what) in ((2, 20, 1024, 64, "SDXL 32x32 level"), (2, 8, 4096, 64, "SD1.5 64x64 level"), (2, 10, 4096, 64, "SDXL 64x64 level")):
    inner = H * D
    qk, vt = rnd(B * S, 2 * inner), rnd(inner, B * S)
    kw = dict(B=B, H=H, D=D, Sq=S, Skv=S, Skv_alloc=S, q_row_stride=2 * inner, k_row_stride=2 * inner, q_batch_stride=S * 2 * inner,
              k_batch_stride=S * 2 * inner, vt_ld=B * S, vt_batch_stride=S)
    whole = ops.attention(qk, qk[:, inner:], vt, kv_split=1, **kw).clone()
    got = ops.attention(qk, qk[:, inner:], vt, kv_split=0, **kw).clone()
    ref = torch.nn.functional.scaled_dot_product_attention(
        qk[:, :inner].view(B, S, H, D).transpose(1, 2).float(), qk[:, inner:].view(B, S, H, D).transpose(1, 2).float(),
        vt.view(H, D, B, S).permute(2, 0, 3, 1).float()).transpose(1, 2).reshape(B * S, inner)
    rel = lambda a: float((a.float() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())  # noqa: E731
    us = {k: round(graph_us(lambda k=k: ops.attention(qk, qk[:, inner:], vt, kv_split=k, **kw)), 2) for k in (1, 0, 1, 0)}
    us0 = round(graph_us(lambda: ops.attention(qk, qk[:, inner:], vt, kv_split=0, **kw)), 2)
    us1 = round(graph_us(lambda: ops.attention(qk, qk[:, inner:], vt, kv_split=1, **kw)), 2)
    print(json.dumps({"DA_ATTN_SPLIT_ALL": mode, "what": what, "B": B, "H": H, "S": S, "plan(default)": plan(B, H, S, D, 0), "us_whole": us1,
                      "us_default": us0, "max_abs_vs_whole": float((got.float() - whole.float()).abs().max()),
                      "rel_rms_vs_fp32": {"whole": round(rel(whole), 5), "default": round(rel(got), 5)}}), flush=True)
