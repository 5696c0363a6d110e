#!/usr/bin/env python
"""Join tools/pmc_k2.py's manifest with a rocprofv3 --pmc counter_collection.csv (dispatch order): per (shape, variant) the mean
of every counter over the timed launches, and the derived shares of a wave's cycles.  Markdown to stdout."""
import csv
import json
import sys
from collections import OrderedDict, defaultdict


def main():
    manifest = json.load(open(sys.argv[1]))
    disp = OrderedDict()
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            if "igemm" not in r["Kernel_Name"]:
                continue
            d = disp.setdefault(int(r["Dispatch_Id"]), {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(disp)
    need = sum(m["reps"] for m in manifest)
    if len(ids) != need:
        print(f"trace has {len(ids)} igemm dispatches, manifest expects {need}", file=sys.stderr)
    i = 0
    names = sorted({c for d in disp.values() for c in d})
    print("| shape | variant | " + " | ".join(names) + " | parked (WAIT_ANY) | issue stall (WAIT_INST_ANY) | active | MFMA busy / wave-cycle budget | LDS conflict / LDS active |")
    print("|---|---|" + "---:|" * (len(names) + 5))
    attn = OrderedDict()
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            if "attn_fwd" in r["Kernel_Name"]:
                d = attn.setdefault(int(r["Dispatch_Id"]), {})
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if attn:
        manifest = manifest + [{"name": "flash attention B2 H20 S1024 D64", "M": 2048, "N": 1280, "K": 1024, "tile": "attn_fwd (default)",
                                "staging": 0, "reps": len(attn), "_attn": True}]
    for m in manifest:
        if m.get("_attn"):
            rows = [attn[j] for j in sorted(attn)[1:]]
            mean = {c: sum(r.get(c, 0.0) for r in rows) / max(len(rows), 1) for c in names}
            wc = mean.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            derived = [mean.get("SQ_WAIT_ANY", 0) / wc, mean.get("SQ_WAIT_INST_ANY", 0) / wc, mean.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                       mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4.0 * wc), mean.get("SQ_LDS_BANK_CONFLICT", 0) / max(mean.get("SQ_LDS_IDX_ACTIVE", 0), 1.0)]
            print(f"| {m['name']} | {m['tile']} | " + " | ".join(f"{mean[c]:.3g}" for c in names) + " | " + " | ".join(f"{v:.3f}" for v in derived) + " |")
            continue
        rows = [disp[j] for j in ids[i + 1:i + m["reps"]]]
        i += m["reps"]
        mean = {c: sum(r.get(c, 0.0) for r in rows) / max(len(rows), 1) for c in names}
        wc = mean.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        # SQ_WAVE_CYCLES / WAIT / ACTIVE count quad-cycles; MFMA busy counts cycles (MI355X_MICROARCH.md)
        derived = [mean.get("SQ_WAIT_ANY", 0) / wc, mean.get("SQ_WAIT_INST_ANY", 0) / wc, mean.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                   mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4.0 * wc), mean.get("SQ_LDS_BANK_CONFLICT", 0) / max(mean.get("SQ_LDS_IDX_ACTIVE", 0), 1.0)]
        print(f"| {m['name']} {m['M']}x{m['N']}x{m['K']} | {m['tile']}/{m['staging']} | " + " | ".join(f"{mean[c]:.3g}" for c in names) +
              " | " + " | ".join(f"{v:.3f}" for v in derived) + " |")


if __name__ == "__main__":
    main()
