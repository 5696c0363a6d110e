#!/usr/bin/env python
"""Workload + report for `rocprofv3 --kernel-trace --pmc FETCH_SIZE`: the five hottest GEMM shapes of an SDXL denoising step, each
launched REPS times back to back on distinct weights (so every launch meets its weight outside L2, as inside the step), under the
XCD column count DA_XCD_GX pins.  `run <manifest.json>` launches; `report <manifest.json> <counter_collection.csv> ...` prints the
fetched MB per launch next to the algorithmic bytes (2 x FETCH_SIZE KiB: the gfx950 wide-stream correction of MI355X_MICROARCH.md)."""
import csv
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
REPS = 6
SHAPES = [  # name, M, N, K, act, residual
    ("GEGLU up 2048x10240x1280", 2048, 10240, 1280, "geglu", False),
    ("FF-down 2048x1280x5120 (+res)", 2048, 1280, 5120, "none", True),
    ("projection 2048x1280x1280 (+res)", 2048, 1280, 1280, "none", True),
    ("Q|K 2048x2560x1280", 2048, 2560, 1280, "none", False),
    ("to_out 8192x640x640 (+res)", 8192, 640, 640, "none", True),
]


def run(manifest):
    import torch
    from diffusers_amd import _lib as L, ops
    from tools.ceiling_table import rnd
    recs = []
    for name, M, N, K, act, has_res in SHAPES:
        x = rnd((M, K))
        ws = [rnd((N, K), K ** -0.5) for _ in range(REPS)]
        res = rnd((M, N)) if has_res else None
        b = rnd((N,))
        if act == "geglu":
            ws = [ops.pack_geglu(w, b)[0] for w in ws]
        ops.linear(x, ws[0], b, act=L.ACT_GEGLU if act == "geglu" else L.ACT_NONE, residual=res)   # variant lookup, not counted apart
        torch.cuda.synchronize()
        for w in ws:
            ops.linear(x, w, b, act=L.ACT_GEGLU if act == "geglu" else L.ACT_NONE, residual=res)
        torch.cuda.synchronize()
        out_cols = N // 2 if act == "geglu" else N
        recs.append({"name": name, "launches": REPS + 1,
                     "algorithmic_bytes": 2 * (M * K + N * K + M * out_cols + (M * N if has_res else 0))})
    json.dump({"gx": os.environ.get("DA_XCD_GX", "auto"), "shapes": recs}, open(manifest, "w"))
    print("pmc_shapes: done", flush=True)


def report(manifest, csvs):
    man = json.load(open(manifest))
    rows = []
    for f in csvs:
        with open(f, newline="") as fh:
            rows += [r for r in csv.DictReader(fh) if r["Counter_Name"] == "FETCH_SIZE" and "igemm" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    i = 0
    print(f"| shape (DA_XCD_GX = {man['gx']}) | fetched MB / launch | algorithmic MB | ratio |")
    print("|---|---:|---:|---:|")
    for s in man["shapes"]:
        mine = rows[i:i + s["launches"]][1:]          # the first launch of a shape is the variant lookup / warm-up
        i += s["launches"]
        if not mine:
            continue
        fb = 2.0 * 1024.0 * sum(float(r["Counter_Value"]) for r in mine) / len(mine)
        print(f"| {s['name']} | {fb / 1e6:.2f} | {s['algorithmic_bytes'] / 1e6:.2f} | {fb / s['algorithmic_bytes']:.2f} |")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        report(sys.argv[2], sys.argv[3:])
