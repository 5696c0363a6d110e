#!/usr/bin/env python
"""LayerNorm-fold cost check (one JSON line per case): consumer GEMM with ln=(stats, fold) vs the same GEMM on a
pre-normalised input, producer GEMM with / without stats_out, and the LayerNorm kernel the fold removes."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
from diffusers_amd import _lib as L, ops  # noqa: E402
import bench_kernels_r2 as B  # noqa: E402

bf16 = torch.bfloat16


def main():
    B.FLUSH = torch.empty(320 << 20, dtype=torch.uint8, device="cuda")
    M, C = 2048, 1280
    x, res = B.rnd((M, C)), B.rnd((M, C))
    wo = B.rnd((C, C), C ** -0.5)
    gamma, beta = B.rnd((C,)) * 0.2 + 1, B.rnd((C,)) * 0.1
    for ptile, pst in ((L.TILE_128x64, 2), (L.TILE_64x64, 1), (L.TILE_128x128, 1)):
        st = ops.RowStats(M, "cuda")
        t0, _ = B.timeit(lambda: ops.linear(x, wo, residual=res, tile=ptile, staging=pst))
        t1, _ = B.timeit(lambda: ops.linear(x, wo, residual=res, tile=ptile, staging=pst, stats_out=st))
        print(json.dumps({"op": "producer", "tile": L.TILE_NAMES[ptile], "staging": pst, "plain_us": round(t0, 1),
                          "with_stats_us": round(t1, 1), "parts": st.parts}), flush=True)
    h = ops.linear(x, wo, residual=res, tile=L.TILE_128x64, staging=2, stats_out=st)
    tln, _ = B.timeit(lambda: ops.layer_norm(h, gamma, beta, 1e-5))
    print(json.dumps({"op": "layernorm_kernel", "us": round(tln, 1)}), flush=True)
    hn = ops.layer_norm(h, gamma, beta, 1e-5)
    for name, N, act in (("to_q", 1280, 0), ("geglu", 10240, L.ACT_GEGLU)):
        w = B.rnd((N, C), C ** -0.5)
        b = B.rnd((N,)) if act else None
        wl, fold = ops.fold_layernorm(w, gamma, beta, 1e-5)
        if act:
            w, b2 = ops.pack_geglu(w, b)
            wl, _ = ops.pack_geglu(wl, None)
            b = b2
        for tile, stg in ((L.TILE_64x64, 1), (L.TILE_128x64, 2), (L.TILE_128x128, 1), (L.TILE_256x128, 2)):
            try:
                t0, _ = B.timeit(lambda: ops.linear(hn, w, b, act=act, tile=tile, staging=stg))
                t1, _ = B.timeit(lambda: ops.linear(h, wl, b, act=act, tile=tile, staging=stg, ln=(st, fold)))
            except RuntimeError:
                continue
            print(json.dumps({"op": "consumer", "name": name, "tile": L.TILE_NAMES[tile], "staging": stg, "plain_us": round(t0, 1),
                              "folded_us": round(t1, 1), "ln_parts": st.parts}), flush=True)


if __name__ == "__main__":
    main()
