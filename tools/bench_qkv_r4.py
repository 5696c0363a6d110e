#!/usr/bin/env python
"""Fused Q | K | V projection (da_gemm_params.vt) against the paired Q|K + V^T launch, round 4: chained microseconds of
[FF-down (+ statistics) -> LayerNorm -> paired launch] vs [FF-down (+ statistics) -> fused projection with the norm folded],
and of the launches on their own, at both SDXL transformer levels.  One JSON object per line (argv[1]: appended)."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import _lib as L, ops  # noqa: E402
from tools.ceiling_table import chain_us, rnd  # noqa: E402


def main():
    out = open(sys.argv[1], "a") if len(sys.argv) > 1 else None
    for level, M, C in (("1280", 2048, 1280), ("640", 8192, 640)):
        h4, res = rnd((M, 4 * C)), rnd((M, C))
        w2, b2 = rnd((C, 4 * C), (4 * C) ** -0.5), rnd((C,))
        gamma, beta = rnd((C,)) * 0.2 + 1, rnd((C,)) * 0.1
        wq, wk, wv = (rnd((C, C), C ** -0.5) for _ in range(3))
        wqk = torch.cat([wq, wk]).contiguous()
        wqkv = torch.cat([wq, wk, wv]).contiguous()
        wl, fold = ops.fold_layernorm(wqkv, gamma, beta, 1e-5)
        st = ops.RowStats(M, "cuda")
        x = ops.linear(h4, w2, b2, residual=res, stats_out=st)
        xn = ops.layer_norm(x, gamma, beta, 1e-5)
        rec = {"op": "fused qkv", "level": level}
        for t, s in ops.QKV_CANDIDATES:
            vt = torch.empty((C, M), device="cuda", dtype=torch.bfloat16)
            try:
                ops.linear(x, wl, ln=(st, fold), tile=t, staging=s, vt_out=(vt, 2 * C))
            except RuntimeError:
                continue
            rec[f"qkv folded {L.TILE_NAMES[t]}/{s}"] = round(min(chain_us(lambda: ops.linear(x, wl, ln=(st, fold), tile=t, staging=s, vt_out=(vt, 2 * C)), 30) for _ in range(3)), 1)
            rec[f"qkv plain {L.TILE_NAMES[t]}/{s}"] = round(min(chain_us(lambda: ops.linear(xn, wqkv, tile=t, staging=s, vt_out=(vt, 2 * C)), 30) for _ in range(3)), 1)
        rec["pair"] = round(min(chain_us(lambda: ops.linear_pair({"x": xn, "w": wqk}, {"x": wv, "w": xn}), 30) for _ in range(3)), 1)
        rec["layernorm"] = round(min(chain_us(lambda: ops.layer_norm(x, gamma, beta, 1e-5), 30) for _ in range(3)), 1)

        def chain_old():
            y = ops.linear(h4, w2, b2, residual=res)
            yn = ops.layer_norm(y, gamma, beta, 1e-5)
            return ops.linear_pair({"x": yn, "w": wqk}, {"x": wv, "w": yn})

        def chain_new():
            y = ops.linear(h4, w2, b2, residual=res, stats_out=st)
            return ops.linear_qkv(y, wl, 2 * C, ln=(st, fold))
        rec["chain FF-down > LN > pair"] = round(min(chain_us(chain_old, 20) for _ in range(3)), 1)
        rec["chain FF-down+stats > fused qkv folded"] = round(min(chain_us(chain_new, 20) for _ in range(3)), 1)
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
