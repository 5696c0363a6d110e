#!/usr/bin/env python
"""K / M sweep of the igemm kernel for rocprofv3 --kernel-trace: pure kernel durations (no event overhead) separate a
launch's fixed cost (dispatch ramp + prologue + epilogue) from its steady-state K-loop rate.
Writes the launch manifest (labels in dispatch order) to argv[1]; tools/ksweep_report.py joins it with the trace."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from diffusers_amd import _lib as L  # noqa: E402
from diffusers_amd import ops  # noqa: E402

REPS = 6
bf16 = torch.bfloat16


def rnd(shape, scale=1.0):
    return (torch.randn(shape, device="cuda") * scale).to(bf16)


def main():
    manifest = []
    flush = torch.empty(320 << 20, dtype=torch.uint8, device="cuda")
    variants = [(L.TILE_128x64, 2), (L.TILE_128x128_W8, 2), (L.TILE_K2_128x80, 1), (L.TILE_K2_128x80, 2), (L.TILE_K2_128x80, 6), (L.TILE_K2_128x80, 7), (L.TILE_K2_128x160, 1),
                (L.TILE_K2_128x160, 6), (L.TILE_K2_128x128, 6),
                (L.TILE_K2_128x128, 1), (L.TILE_K1_256x128, 2), (L.TILE_K1_128x320, 1)]
    if len(sys.argv) > 2 and sys.argv[2] == "r2":       # the round-2 variant set
        variants = [(L.TILE_64x64, 1), (L.TILE_128x64, 2), (L.TILE_128x128, 1), (L.TILE_128x128_W8, 2), (L.TILE_256x128, 2)]
    cases = [(2048, 1280, k) for k in (64, 320, 1280, 2560, 5120)]
    cases += [(m, 1280, 1280) for m in (256, 1024, 4096, 8192)]
    cases += [(2048, n, 1280) for n in (320, 640, 2560, 5120)]
    for cold in (False, True):
        for M, N, K in cases:
            x, w, b, r = rnd((M, K)), rnd((N, K), K ** -0.5), rnd((N,)), rnd((M, N))
            for tile, st in variants:
                for epi in ("plain", "bias+res"):
                    kw = dict(bias=b, residual=r) if epi != "plain" else {}
                    try:
                        ops.linear(x, w, tile=tile, staging=st, **kw)   # warm-up launch: counted in the manifest too
                    except RuntimeError:
                        continue
                    manifest.append({"M": M, "N": N, "K": K, "tile": L.TILE_NAMES[tile], "staging": st, "epi": epi,
                                     "cold": cold, "reps": REPS + 1})
                    for _ in range(REPS):
                        if cold:
                            flush.zero_()
                        ops.linear(x, w, tile=tile, staging=st, **kw)
            torch.cuda.synchronize()
    Path(sys.argv[1]).write_text(json.dumps(manifest))


if __name__ == "__main__":
    main()
