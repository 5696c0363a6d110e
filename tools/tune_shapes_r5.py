#!/usr/bin/env python
"""Round 5: da_gemm_tune (cold weights, re-warmed activations -- the in-situ condition of a denoising step) on SDXL's 32 x 32-level
3 x 3 convs with the split-K variants of the first family competing.  Run with DIFFUSERS_AMD_SPLITK=1 DIFFUSERS_AMD_GEMM_FAMILY=all
DIFFUSERS_AMD_TUNE_DB=<table without these entries>.  Prints old entry -> tuned entry."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import _lib as L, ops, tuning  # noqa: E402

bf16 = torch.bfloat16
old = json.loads((ROOT / "diffusers_amd" / "tuned" / "gfx950.json").read_text())["entries"]
g = torch.Generator("cpu").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g).to(bf16).to("cuda")  # noqa: E731
out = []
for (H, C, N, up, res) in ((32, 1280, 1280, False, False), (32, 1280, 1280, False, True), (16, 1280, 1280, True, False), (32, 1920, 1280, False, False),
                           (32, 2560, 1280, False, False), (32, 640, 1280, False, False)):
    x = rnd(2, H, H, C)
    w = rnd(N, 9 * C) * (9 * C) ** -0.5
    b = rnd(N)
    Ho = 2 * H if up else H
    r = rnd(2, Ho, Ho, N) if res else None
    rv = rnd(2, N)
    before = set(tuning.table())
    ops.conv2d_nhwc(x, w, b, ksize=3, up=up, residual=r, rowvec=None if res else rv)
    torch.cuda.synchronize()
    new = [k for k in tuning.table() if k not in before]
    for k in new:
        e = tuning.table()[k]
        rec = {"key": k, "old": old.get(k), "tuned": [L.TILE_NAMES[e[0]], e[1], round(e[2], 1), e[3]]}
        print(json.dumps(rec), flush=True)
        out.append(rec)
(ROOT / "gpurun_out" / "tune_shapes_r5.jsonl").write_text("".join(json.dumps(r) + "\n" for r in out))
