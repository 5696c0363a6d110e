#!/usr/bin/env python
"""Every (tile, ring) variant of the second GEMM family on the plain nn.Linear shapes of the fused Q | K | V projection (and, for
comparison, of the Q | K block and of one projection): [warm, 12 distinct weights back to back] microseconds per launch.  This is
the probe that showed the eight-waves-per-slice tiles winning wherever they cover the problem in ONE round of the 256 CUs
(M 2048 x N 3840: 240 tiles of 256 x 128 at 23.4 us against 29.8 us on 128 x 160 and 32.2 us on 128 x 80); the result is
profiles/r04m_fused_qkv_one_round_tiles.jsonl.  One JSON object per shape."""
import json, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import _lib as L, ops
from tools.ceiling_table import chain_us, rnd
for M, N, K in ((2048, 3840, 1280), (8192, 1920, 640), (2048, 2560, 1280), (2048, 1280, 1280)):
    x, w = rnd((M, K)), rnd((N, K), K ** -0.5)
    # 12 distinct weights so that the chain meets them colder than one repeated launch does
    ws = [rnd((N, K), K ** -0.5) for _ in range(12)]
    rec = {"shape": [M, N, K]}
    for t in range(L.FIRST_K2_TILE, len(L.TILE_NAMES)):
        for s in (1, 2, 6, 7):
            try:
                ops.linear(x, w, tile=t, staging=s)
                torch.cuda.synchronize()
            except RuntimeError:
                continue
            warm = min(chain_us(lambda: ops.linear(x, w, tile=t, staging=s), 30) for _ in range(2))
            def many():
                for ww in ws:
                    ops.linear(x, ww, tile=t, staging=s)
            cold = min(chain_us(many, 5) for _ in range(2)) / len(ws)
            rec[f"{L.TILE_NAMES[t]}/{s}"] = [round(warm, 1), round(cold, 1)]
    print(json.dumps(rec), flush=True)
