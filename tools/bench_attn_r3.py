#!/usr/bin/env python
"""Flash attention at SDXL's shapes, round 3: ring depth 2 / 3 / 4, the PV-delayed loop, 64-query workgroups -- each timed cold (K / V
evicted: how the kernel meets its operands inside the denoising step, right after the projection GEMM of another XCD wrote them),
warm and as a chain.  One JSON object per line."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import ops  # noqa: E402
from tools.ceiling_table import chain_us, rnd, timeit  # noqa: E402
import tools.ceiling_table as CT  # noqa: E402


def main():
    CT.FLUSH = torch.empty(320 << 20, dtype=torch.uint8, device="cuda")
    for name, B, H, S, Skv, D in [("sdxl self 1024", 2, 20, 1024, 1024, 64), ("sdxl self 4096", 2, 10, 4096, 4096, 64),
                                  ("sdxl cross 1024", 2, 20, 1024, 77, 64), ("flux joint", 1, 24, 4608, 4608, 128)]:
        inner = H * D
        sa = ((Skv + 15) // 16) * 16
        q, k, vt = rnd((B * S, inner)), rnd((B * sa, inner)), rnd((inner, B * sa))

        def run(**kw):
            return ops.attention(q, k, vt, B=B, H=H, D=D, Sq=S, Skv=Skv, Skv_alloc=sa, q_row_stride=inner, k_row_stride=inner,
                                 q_batch_stride=S * inner, k_batch_stride=sa * inner, vt_ld=B * sa, vt_batch_stride=sa, **kw)
        ref = run(ring_slots=2, pv_delay=-1)
        rec = {"op": "attn", "name": name, "tflop": round(4.0 * B * H * S * Skv * D / 1e12, 4)}
        variants = {"default": {}, "ring2": dict(ring_slots=2, pv_delay=-1), "ring3": dict(ring_slots=3, pv_delay=-1),
                    "ring4": dict(ring_slots=4, pv_delay=-1), "pipe": dict(pv_delay=1), "pipe2": dict(pv_delay=2)}
        if D == 64:
            variants["q64"] = dict(q_block=64, ring_slots=2, pv_delay=-1)
        for vn, kw in variants.items():
            try:
                y = run(**kw)
            except RuntimeError as e:
                rec[vn] = str(e)[:40]
                continue
            fn = lambda: run(**kw)  # noqa: E731
            cold, _ = timeit(fn, iters=8, warm=1)
            warm, _ = timeit(fn, iters=8, warm=1, flush=False)
            rec[vn] = [round(cold, 1), round(warm, 1), round(chain_us(fn, 30), 1), bool(torch.equal(y, ref))]
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
