#!/usr/bin/env python
"""Flash-attention launches for `rocprofv3 --kernel-trace --pmc <SQ counters>` (round 4): both kernel generations and the opt-in
variants at SDXL's / Flux's shapes, 3 launches each after a warm one, in a fixed order (`run <manifest>`); `report <manifest> <csv>`
joins the per-dispatch counters in dispatch order and prints the shares of a wave's cycles."""
import csv
import json
import sys
from collections import OrderedDict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
REPS = 4
SHAPES = [("sdxl self S1024 (B2 H20 D64)", 2, 20, 1024, 64), ("sdxl self S4096 (B2 H10 D64)", 2, 10, 4096, 64),
          ("S4096 24 pairs (3 wg/CU, D64)", 1, 24, 4096, 64), ("flux joint S4608 (H24 D128)", 1, 24, 4608, 128)]
VARIANTS = [("v1", dict(algo=1)), ("v2", dict(algo=2)), ("v2 aug", dict(algo=3)), ("v2 rsm", dict(algo=4))]


def run(manifest):
    import torch
    from diffusers_amd import ops
    bf16 = torch.bfloat16
    recs = []
    for name, B, H, S, D in SHAPES:
        inner = H * D
        q, k = (torch.randn((B * S, inner), device="cuda").to(bf16) for _ in range(2))
        vt = torch.randn((inner, B * S), device="cuda").to(bf16)
        for vn, kw in VARIANTS:
            for _ in range(REPS):
                ops.attention(q, k, vt, B=B, H=H, D=D, Sq=S, Skv=S, Skv_alloc=S, q_row_stride=inner, k_row_stride=inner,
                              q_batch_stride=S * inner, k_batch_stride=S * inner, vt_ld=B * S, vt_batch_stride=S, **kw)
            torch.cuda.synchronize()
            recs.append({"shape": name, "variant": vn, "reps": REPS, "gflop": 4.0 * B * H * S * S * D / 1e9})
    json.dump(recs, open(manifest, "w"))
    print("pmc_attn: done", flush=True)


def report(manifest, path):
    man = json.load(open(manifest))
    disp = OrderedDict()
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            if "attn" not in r["Kernel_Name"]:
                continue
            d = disp.setdefault(int(r["Dispatch_Id"]), {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(disp)
    names = sorted({c for d in disp.values() for c in d})
    print("| shape | variant | " + " | ".join(names) + " | parked | issue stall | active | MFMA busy / wave-cycle budget | LDS conflict / LDS active |")
    print("|---|---|" + "---:|" * (len(names) + 5))
    i = 0
    for m in man:
        rows = [disp[j] for j in ids[i:i + m["reps"]]][1:]
        i += m["reps"]
        if not rows:
            continue
        mean = {c: sum(r.get(c, 0.0) for r in rows) / len(rows) for c in names}
        wc = mean.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        lds = mean.get("SQ_LDS_IDX_ACTIVE", 0.0) or 1.0
        # SQ_WAVE_CYCLES etc. count quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD: budget = 4 x wave quad-cycles / waves per SIMD ... reported raw
        print(f"| {m['shape']} | {m['variant']} | " + " | ".join(f"{mean[c]:.3g}" for c in names) +
              f" | {mean.get('SQ_WAIT_ANY', 0) / wc:.3f} | {mean.get('SQ_WAIT_INST_ANY', 0) / wc:.3f} | {mean.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f} | "
              f"{mean.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * wc):.3f} | {mean.get('SQ_LDS_BANK_CONFLICT', 0) / lds:.3f} |")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        report(sys.argv[2], sys.argv[3])
