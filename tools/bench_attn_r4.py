#!/usr/bin/env python
"""Flash attention, round 4: the second-generation kernel (csrc/attention2.hip) against the first at the shapes of the BASELINE
configs, plus occupancy probes (the same sequence length at 1 / 2 / 3 workgroups per CU) that separate the cost of one tile from
the load balance of a launch.  Per variant: chained microseconds per launch (back-to-back launches, HIP events), TFLOP/s, fraction
of the 2.5 PFLOP/s bf16 MFMA peak; per shape: rel-rms of each generation against an fp32 torch reference.
One JSON object per line (argv[1]: also appended to that file)."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import ops  # noqa: E402
from tools.ceiling_table import chain_us, rnd  # noqa: E402

PEAK = 2500.0


def ref_fp32(q, k, vt, B, H, D, S, Skv, sa):
    """fp32 softmax(q k^T / sqrt D) v on the GPU, chunked over queries."""
    inner = H * D
    qf = q.view(B, S, H, D).permute(0, 2, 1, 3).float()
    kf = k.view(B, sa, H, D)[:, :Skv].permute(0, 2, 1, 3).float()
    vf = vt.view(H, D, B, sa)[..., :Skv].permute(2, 0, 3, 1).float()      # [B][H][Skv][D]
    out = torch.empty((B, H, S, D), device=q.device, dtype=torch.float32)
    step = max(1, (1 << 26) // max(1, Skv * H * B))
    for s0 in range(0, S, step):
        sc = torch.matmul(qf[:, :, s0:s0 + step], kf.transpose(-1, -2)) * (D ** -0.5)
        out[:, :, s0:s0 + step] = torch.matmul(torch.softmax(sc, dim=-1), vf)
    return out.permute(0, 2, 1, 3).reshape(B * S, inner)


def relrms(a, b):
    return float(((a.float() - b).pow(2).mean() / b.pow(2).mean()).sqrt())


def main():
    out = open(sys.argv[1], "a") if len(sys.argv) > 1 else None
    quick = "--quick" in sys.argv
    shapes = [
        # name, B, H, S, Skv, D, check accuracy
        ("probe S1024 32 pairs (1 wg/CU at 128q)", 1, 32, 1024, 1024, 64, False),
        ("probe S1024 64 pairs (2 wg/CU)", 1, 64, 1024, 1024, 64, False),
        ("probe S1024 96 pairs (3 wg/CU)", 1, 96, 1024, 1024, 64, False),
        ("probe S4096 8 pairs (1 wg/CU)", 1, 8, 4096, 4096, 64, False),
        ("probe S4096 16 pairs (2 wg/CU)", 1, 16, 4096, 4096, 64, False),
        ("probe S4096 24 pairs (3 wg/CU)", 1, 24, 4096, 4096, 64, False),
        ("sdxl self 1024 (B2 H20)", 2, 20, 1024, 1024, 64, True),
        ("sdxl self 4096 (B2 H10)", 2, 10, 4096, 4096, 64, True),
        ("sdxl cross 1024 (B2 H20 Skv77)", 2, 20, 1024, 77, 64, True),
        ("sdxl cross 4096 (B2 H10 Skv77)", 2, 10, 4096, 77, 64, True),
        ("flux joint 4608 (H24 D128)", 1, 24, 4608, 4608, 128, True),
        ("wan self 32760 (H12 D128)", 1, 12, 32760, 32760, 128, False),
        ("wan cross 32760 x 512 (H12 D128)", 1, 12, 32760, 512, 128, False),
    ]
    if quick:
        shapes = [s for s in shapes if "wan" not in s[0]]
    for name, B, H, S, Skv, D, check in shapes:
        inner = H * D
        sa = ((Skv + 15) // 16) * 16
        torch.manual_seed(0)
        q, k, vt = rnd((B * S, inner)), rnd((B * sa, inner)), rnd((inner, B * sa))
        if sa != Skv:                                  # the padding the host side provides: zero keys / values past Skv
            k.view(B, sa, inner)[:, Skv:] = 0
            vt.view(inner, B, sa)[:, :, Skv:] = 0

        def run(**kw):
            return ops.attention(q, k, vt, B=B, H=H, D=D, Sq=S, Skv=Skv, Skv_alloc=sa, q_row_stride=inner, k_row_stride=inner,
                                 q_batch_stride=S * inner, k_batch_stride=sa * inner, vt_ld=B * sa, vt_batch_stride=sa, **kw)
        tflop = 4.0 * B * H * S * Skv * D / 1e12
        rec = {"op": "attn", "name": name, "B": B, "H": H, "S": S, "Skv": Skv, "D": D, "tflop": round(tflop, 4)}
        variants = {
            "v1 default": dict(algo=1),
            "v2 128q ring3": dict(algo=2, q_block=128, ring_slots=3),
            "v2 128q ring4": dict(algo=2, q_block=128, ring_slots=4),
            "v2 256q ring3": dict(algo=2, q_block=256, ring_slots=3),
            "v2 256q ring4": dict(algo=2, q_block=256, ring_slots=4),
            "v2aug 128q ring3": dict(algo=3, q_block=128, ring_slots=3),
            "v2aug 256q ring3": dict(algo=3, q_block=256, ring_slots=3),
            "v2rsm 128q ring3": dict(algo=4, q_block=128, ring_slots=3),
            "v2rsm 256q ring3": dict(algo=4, q_block=256, ring_slots=3),
            "v2augrsm 128q ring3": dict(algo=5, q_block=128, ring_slots=3),
            "v2augrsm 256q ring3": dict(algo=5, q_block=256, ring_slots=3),
            "default": dict(),
        }
        n = 6 if S > 8192 else 30
        outs = {}
        for vn, kw in variants.items():
            try:
                outs[vn] = run(**kw)
                torch.cuda.synchronize()
            except RuntimeError as e:
                rec[vn] = str(e)[:60]
                continue
            us = min(chain_us(lambda: run(**kw), n) for _ in range(2))
            rec[vn] = {"us": round(us, 1), "tflops": round(tflop / us * 1e6, 1), "frac": round(tflop / us * 1e6 / PEAK, 4)}
        if "v2 128q ring3" in outs:
            base = outs["v2 128q ring3"]
            rec["v2 variants bit-identical"] = all(torch.equal(outs[v], base) for v in outs if v.startswith("v2 "))
            if "v2aug 128q ring3" in outs and "v2aug 256q ring3" in outs:
                rec["v2aug variants bit-identical"] = bool(torch.equal(outs["v2aug 128q ring3"], outs["v2aug 256q ring3"]))
        if check:
            ref = ref_fp32(q, k, vt, B, H, D, S, Skv, sa)
            for vn in ("v1 default", "v2 128q ring3", "v2aug 128q ring3", "v2rsm 128q ring3", "v2augrsm 128q ring3", "default"):
                if vn in outs:
                    rec["relrms " + vn] = float(f"{relrms(outs[vn], ref):.3e}")
                    rec["finite " + vn] = bool(torch.isfinite(outs[vn].float()).all())
        line = json.dumps(rec)
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()


if __name__ == "__main__":
    main()
