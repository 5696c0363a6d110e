#!/usr/bin/env python
"""GroupNorm / LayerNorm at the SDXL U-Net's shapes, round 4: chained microseconds per launch and achieved GB/s (algorithmic
bytes: GroupNorm 3 passes = 6 B / element, LayerNorm 2 = 4 B / element).  The GroupNorm plan knobs come from the environment
(DA_GN_THREADS / DA_GN_MINPIX / DA_GN_MAXBLK / DA_GN_CAP, read by the library at first use): run once per setting.
One JSON object per line (argv[1]: appended)."""
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import ops  # noqa: E402
from tools.ceiling_table import chain_us, rnd  # noqa: E402


def main():
    out = open(sys.argv[1], "a") if len(sys.argv) > 1 else None
    knobs = {k: os.environ.get(k) for k in ("DA_GN_THREADS", "DA_GN_MINPIX", "DA_GN_MAXBLK", "DA_GN_CAP") if os.environ.get(k)}
    # (B, HW, C1, C2): every GroupNorm of an SDXL denoising step (count per step in the comment)
    gn_shapes = [(2, 16384, 320, 0), (2, 16384, 640, 320), (2, 16384, 320, 320), (2, 16384, 640, 0),
                 (2, 4096, 320, 0), (2, 4096, 640, 0), (2, 4096, 1280, 640), (2, 4096, 640, 640), (2, 4096, 640, 320),
                 (2, 4096, 1280, 0), (2, 1024, 640, 0), (2, 1024, 1280, 0), (2, 1024, 1280, 1280), (2, 1024, 1280, 640)]
    tot = 0.0
    for B, HW, C1, C2 in gn_shapes:
        C = C1 + C2
        x = rnd((B, HW, C1))
        x2 = rnd((B, HW, C2)) if C2 else None
        g, b = rnd((C,)), rnd((C,))
        fn = lambda: ops.group_norm_nhwc(x, g, b, 32, 1e-5, silu=True, x2=x2)  # noqa: E731
        us = min(chain_us(fn, 30) for _ in range(3))
        mb = B * HW * C * 6 / 1e6
        tot += us
        rec = {"op": "groupnorm", "B": B, "HW": HW, "C": f"{C1}+{C2}", "us": round(us, 1), "GBps": round(mb / us * 1e3, 0), "knobs": knobs}
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
    print(json.dumps({"op": "groupnorm sum over the 14 shapes", "us": round(tot, 1), "knobs": knobs}), flush=True)
    if out:
        out.write(json.dumps({"op": "groupnorm sum over the 14 shapes", "us": round(tot, 1), "knobs": knobs}) + "\n")
    for M, C in ((2048, 1280), (8192, 640)):
        x = rnd((M, C))
        g, b = rnd((C,)), rnd((C,))
        us = min(chain_us(lambda: ops.layer_norm(x, g, b, 1e-5), 40) for _ in range(3))
        rec = {"op": "layernorm", "M": M, "C": C, "us": round(us, 1), "GBps": round(M * C * 4 / 1e6 / us * 1e3, 0)}
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
