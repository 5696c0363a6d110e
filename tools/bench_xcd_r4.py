#!/usr/bin/env python
"""XCD tile-mapping experiment (round 4): the chain attn.to_out (+ residual, + statistics) -> to_q (folded
LayerNorm) and the chain to_out -> LayerNorm -> GEGLU projection -> FF-down at the SDXL 1280 level, timed as chains (the consumer
reads what the producer just wrote), under the XCD column count the environment pins (DA_XCD_GX, read once by the library).
One JSON line (argv[1]: appended)."""
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import _lib as L, ops  # noqa: E402
from tools.ceiling_table import chain_us, rnd  # noqa: E402


def main():
    out = open(sys.argv[1], "a") if len(sys.argv) > 1 else None
    M, C = 2048, 1280
    a, res = rnd((M, C)), rnd((M, C))
    wo, bo = rnd((C, C), C ** -0.5), rnd((C,))
    gamma, beta = rnd((C,)) * 0.2 + 1, rnd((C,)) * 0.1
    wq = rnd((C, C), C ** -0.5)
    wql, foldq = ops.fold_layernorm(wq, gamma, beta, 1e-5)
    w1, b1 = rnd((8 * C, C), C ** -0.5), rnd((8 * C,))
    w1p, b1p = ops.pack_geglu(w1, b1)
    w2, b2 = rnd((C, 4 * C), (4 * C) ** -0.5), rnd((C,))
    st = ops.RowStats(M, "cuda")
    # a pool of distinct weights so that every launch of a chain meets its weight cold in L2 (as inside the denoising step)
    pool = [rnd((C, C), C ** -0.5) for _ in range(24)]

    def chain_proj():
        x = a
        for i in range(0, 24, 2):
            x = ops.linear(x, pool[i], bo, residual=res)
            x = ops.linear(x, pool[i + 1])
        return x

    def chain_fold():
        x = ops.linear(a, wo, bo, residual=res, stats_out=st)
        return ops.linear(x, wql, ln=(st, foldq))

    def chain_ff():
        x = ops.linear(a, wo, bo, residual=res)
        h = ops.linear(ops.layer_norm(x, gamma, beta, 1e-5), w1p, b1p, act=L.ACT_GEGLU)
        return ops.linear(h, w2, b2, residual=x)
    rec = {"op": "xcd mapping", "DA_XCD_GX": os.environ.get("DA_XCD_GX", "auto")}
    rec["24 projections 2048x1280x1280, distinct weights (us per launch)"] = round(min(chain_us(chain_proj, 10) for _ in range(3)) / 24, 2)
    rec["to_out+stats > to_q folded (us per chain)"] = round(min(chain_us(chain_fold, 30) for _ in range(3)), 1)
    rec["to_out > LN > GEGLU > FF-down (us per chain)"] = round(min(chain_us(chain_ff, 20) for _ in range(3)), 1)
    print(json.dumps(rec), flush=True)
    if out:
        out.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
