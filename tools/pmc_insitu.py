#!/usr/bin/env python
"""Round 6: what bounds the implicit-GEMM launch populations IN SITU (inside an eager SDXL denoising step, every operand met outside the
XCD's L2 as in the real loop), from rocprofv3 counter passes over tools/pmc_one_step.py joined with its launch log.

usage: pmc_insitu.py <launch_log.json> <skip> <out.md> <pass_dir> [<pass_dir> ...]
Each pass directory holds one `--kernel-trace --pmc ...` run (counter_collection.csv + kernel_trace.csv).  Per population (shape x role
x kernel) the report gives the launch duration from the trace of that pass and the mean of every counter; derived columns:
  GUI GHz       = GRBM_GUI_ACTIVE / 8 XCDs / duration: the effective clock (MI355X_MICROARCH.md "DVFS give-back") for long launches; an
                  UPPER bound for short ones (GUI-active also counts the dispatch ramp outside the kernel's begin / end timestamps)
  L2 hit        = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  mfma busy     = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): share of the GUI-active cycles the matrix pipes were busy
  parked / stall / active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
Profiled runs clock lower than unprofiled ones (guide: never compare a profiled arm with an unprofiled one): ratios only."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from pmc_traffic import is_igemm, short_kernel  # noqa: E402


def load_pass(root: Path, skip: int, n_log: int):
    cnt = defaultdict(dict)                      # dispatch id -> {counter: value}
    names = {}
    for f in root.rglob("*counter_collection.csv"):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                d = int(r["Dispatch_Id"])
                cnt[d][r["Counter_Name"]] = cnt[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                names[d] = r["Kernel_Name"]
    dur = {}
    for f in root.rglob("*kernel_trace.csv"):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                try:
                    dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3     # us
                except (KeyError, ValueError):
                    pass
    ids = sorted(d for d in cnt if is_igemm(names[d]))[skip:]
    if len(ids) != n_log:
        print(f"{root}: {len(ids)} implicit-GEMM dispatches behind the hoisted ones, {n_log} logged launches: pass skipped")
        return None
    return [(names[d], cnt[d], dur.get(d)) for d in ids]


def main():
    log = json.loads(Path(sys.argv[1]).read_text())
    skip, out_md = int(sys.argv[2]), Path(sys.argv[3])
    pops = defaultdict(lambda: {"n": 0, "flop": 0.0, "c": defaultdict(float), "cn": defaultdict(int), "us": 0.0, "usn": 0})
    for root in map(Path, sys.argv[4:]):
        rows = load_pass(root, skip, len(log))
        if rows is None:
            continue
        for ent, (name, counters, us) in zip(log, rows):
            p = pops[(ent["pop"], short_kernel(name))]
            p["flop"] = ent["flop"]
            for k, v in counters.items():
                p["c"][k] += v
                p["cn"][k] += 1
            if us is not None:
                p["us"] += us
                p["usn"] += 1
    lines = ["# Implicit-GEMM launch populations in situ: counters per launch (rocprofv3 --pmc over an eager SDXL step; profiled clocks)", "",
             "| population | kernel | us (profiled) | TFLOP/s | GUI GHz | L2 hit | mfma busy | parked | stalled | issuing |",
             "|---|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
    rec = {}
    for (pop, kern), p in sorted(pops.items(), key=lambda kv: -kv[1]["us"]):
        m = {k: p["c"][k] / p["cn"][k] for k in p["c"]}
        us = p["us"] / max(p["usn"], 1)
        gui = m.get("GRBM_GUI_ACTIVE")
        clock = gui / 8.0 / (us * 1e3) if gui and us else None
        hit = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m else None
        wc = m.get("SQ_WAVE_CYCLES")
        busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * gui) if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in m else None
        f = lambda v, fmt="{:.2f}": fmt.format(v) if v is not None else "-"   # noqa: E731
        frac = lambda k: (m[k] / wc if wc and k in m else None)             # noqa: E731
        lines.append(f"| {pop} | `{kern}` | {us:.1f} | {p['flop'] / us / 1e6 if us else 0:.0f} | {f(clock)} | {f(hit)} | {f(busy)} | "
                     f"{f(frac('SQ_WAIT_ANY'))} | {f(frac('SQ_WAIT_INST_ANY'))} | {f(frac('SQ_ACTIVE_INST_ANY'))} |")
        rec[f"{pop} :: {kern}"] = {"us_profiled": us, "clock_ghz": clock, "l2_hit": hit, "mfma_busy": busy, **{k: v for k, v in m.items()}}
    out_md.write_text("\n".join(lines) + "\n")
    out_md.with_suffix(".json").write_text(json.dumps(rec, indent=1) + "\n")
    print("\n".join(lines[:24]))


if __name__ == "__main__":
    main()
