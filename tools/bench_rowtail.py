#!/usr/bin/env python
"""Round 6: tile-count quantisation of Flux's single-block proj_mlp (4608 x 12288 x 3072, tanh-GELU: 18 x 48 = 864 tiles of 256 x 256 =
3.375 rounds of 256 CUs -- the fourth round is 37 % full).  The launch as ONE k3:256x256 GEMM against a ROW SPLIT: rows [0, 4096) =
16 x 48 = 768 tiles = exactly three rounds on k3:256x256, rows [4096, 4608) as a second launch on a tile that covers the chip once with
half-size work (k1:256x128 / k1:128x256: 192 workgroups).  Chains over rotating (cold) weights, HIP events around the chain; every form
writes the same bits (checked).  usage: bench_rowtail.py [out.jsonl]"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import _lib as L  # noqa: E402
from diffusers_amd import ops  # noqa: E402

bf16 = torch.bfloat16
dev = "cuda"


def rnd(shape, seed, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(bf16).to(dev)


def chain(fn, n, reps=2):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for i in range(n):
            fn(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)


def main():
    out_path = Path(sys.argv[1]) if len(sys.argv) > 1 else None
    recs = []
    for name, M, N, K, act, rows in (("flux single proj_mlp", 4608, 12288, 3072, L.ACT_GELU_TANH, 4096),):
        nw = max(2, int(600e6 // (N * K * 2)) + 1)
        x, b = rnd((M, K), 1), rnd((N,), 2)
        ws = [rnd((N, K), 10 + i, K ** -0.5) for i in range(nw)]
        out = torch.empty((M, N), device=dev, dtype=bf16)
        k3 = dict(tile=L.TILE_K3_256x256, staging=L.STAGE_LDS_DIRECT)

        def whole(i):
            ops.linear(x, ws[i], b, act=act, out=out, **k3)
        forms = {"one launch k3:256x256": whole}
        for tname, tkw in (("k1:256x128/3", dict(tile=L.TILE_K1_256x128, staging=L.STAGE_LDS_DIRECT3)),
                           ("k1:256x128/2", dict(tile=L.TILE_K1_256x128, staging=L.STAGE_LDS_DIRECT)),
                           ("k1:128x256/3", dict(tile=L.TILE_K1_128x256, staging=L.STAGE_LDS_DIRECT3)),
                           ("k3:256x256", k3)):
            def split(i, tkw=tkw):
                ops.linear(x[:rows], ws[i], b, act=act, out=out[:rows], **k3)
                ops.linear(x[rows:], ws[i], b, act=act, out=out[rows:], **tkw)
            forms[f"rows {rows} on k3:256x256 + {M - rows} on {tname}"] = split
        whole(0)
        ref = out.clone()
        rec = {"shape": f"{name} {M}x{N}x{K}", "weights": nw, "us": {}}
        for fname, fn in forms.items():
            out.zero_()
            fn(0)
            torch.cuda.synchronize()
            rec.setdefault("bit_identical", {})[fname] = bool(torch.equal(out, ref))
        for rep in range(2):
            for fname, fn in forms.items():
                rec["us"].setdefault(fname, []).append(round(chain(fn, nw), 1))
        print(json.dumps(rec), flush=True)
        recs.append(rec)
    if out_path:
        out_path.write_text("\n".join(json.dumps(r) for r in recs) + "\n")


if __name__ == "__main__":
    main()
