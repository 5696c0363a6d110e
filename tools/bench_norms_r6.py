#!/usr/bin/env python
"""Round 6: GroupNorm over several workgroups per slab in ONE launch (gn_fused_kernel with gridDim.z parts + the sync buffer) against
the round-5 forms at the shapes of an SDXL / SD1.5 step.  Chained microseconds per GroupNorm (HIP-graph replay of back-to-back
launches), the forms switched per call in one process.  usage: bench_norms_r6.py out.jsonl"""
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import ops  # noqa: E402
from tools.bench_norms_r5 import SD15, SDXL, chain_us  # noqa: E402
from tools.ceiling_table import rnd  # noqa: E402


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    for name, shapes in (("sdxl", SDXL), ("sd15", SD15)):
        tot = {"r5": 0.0, "r6": 0.0}
        for B, HW, C1, C2, cnt in shapes:
            C = C1 + C2
            x = rnd((B, HW, C1))
            x2 = rnd((B, HW, C2)) if C2 else None
            g, b = rnd((C,)), rnd((C,))
            fn = lambda: ops.group_norm_nhwc(x, g, b, 32, 1e-5, silu=True, x2=x2)  # noqa: E731
            rec = {"model": name, "B": B, "HW": HW, "C": f"{C1}+{C2}", "per_step": cnt, "MB": round(B * HW * C * 2 / 1e6, 1)}
            for key, env in (("r5", {"DA_GN_MULTI": "0"}), ("r6", {"DA_GN_MULTI": "1"}), ("r5b", {"DA_GN_MULTI": "0"}), ("r6b", {"DA_GN_MULTI": "1"})):
                os.environ.update(env)
                us = min(chain_us(fn, 30) for _ in range(3))
                rec[key + "_us"] = round(us, 1)
                if key in tot:
                    tot[key] += us * cnt
            rec["GBps_r6"] = round(2 * B * HW * C * 2 / rec["r6_us"] / 1e3, 0)
            print(json.dumps(rec), flush=True)
            if out:
                out.write(json.dumps(rec) + "\n")
        rec = {"model": name, "op": "sum over one step's GroupNorms (us)", **{k: round(v, 1) for k, v in tot.items()}}
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
    torch.cuda.synchronize()
    print(json.dumps({"gn_sync_error": ops.gn_sync_error()}))


if __name__ == "__main__":
    main()
