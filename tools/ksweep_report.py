#!/usr/bin/env python
"""Join tools/ksweep.py's manifest with the rocprofv3 kernel trace: per configuration the minimum / median kernel
duration, then per (tile, epilogue) the fixed cost and the per-1280-K slope from the K sweep.  Markdown to stdout."""
import csv
import json
import sys
from collections import defaultdict


def main():
    manifest = json.load(open(sys.argv[1]))
    rows = []
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            if "igemm_bf16_kernel" in r["Kernel_Name"] or "igemm2_bf16_kernel" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    need = sum(m["reps"] for m in manifest)
    if len(rows) != need:
        print(f"trace has {len(rows)} igemm dispatches, manifest expects {need}", file=sys.stderr)
    i = 0
    res = {}
    for m in manifest:
        d = sorted(x[1] / 1e3 for x in rows[i + 1:i + m["reps"]])   # drop the warm-up launch
        i += m["reps"]
        res[(m["cold"], m["M"], m["N"], m["K"], m["tile"], m["staging"], m["epi"])] = (d[0], d[len(d) // 2])
    print("| caches | M | N | K | tile/staging | epilogue | min us | median us | TFLOP/s (min) |\n|---|---:|---:|---:|---|---|---:|---:|---:|")
    for k, (mn, md) in res.items():
        cold, M, N, K, tile, st, epi = k
        print(f"| {'cold' if cold else 'warm'} | {M} | {N} | {K} | {tile}/{st} | {epi} | {mn:.1f} | {md:.1f} | {2.0 * M * N * K / mn / 1e6:.0f} |")
    print("\n## K sweep at M 2048, N 1280: duration = fixed + slope * K/1280 (least squares over K = 320 .. 5120)\n")
    print("| caches | tile/staging | epilogue | fixed us | us per 1280 of K | steady-state TFLOP/s | K = 64 launch us |\n|---|---|---|---:|---:|---:|---:|")
    groups = defaultdict(list)
    for (cold, M, N, K, tile, st, epi), (mn, md) in res.items():
        if M == 2048 and N == 1280:
            groups[(cold, tile, st, epi)].append((K, mn))
    for (cold, tile, st, epi), pts in groups.items():
        fit = [(k / 1280.0, t) for k, t in pts if k >= 320]
        n = len(fit)
        if n < 2:
            continue
        sx, sy = sum(p[0] for p in fit), sum(p[1] for p in fit)
        sxx, sxy = sum(p[0] ** 2 for p in fit), sum(p[0] * p[1] for p in fit)
        slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
        icpt = (sy - slope * sx) / n
        k64 = dict(pts).get(64, float("nan"))
        print(f"| {'cold' if cold else 'warm'} | {tile}/{st} | {epi} | {icpt:.1f} | {slope:.1f} | {2.0 * 2048 * 1280 * 1280 / slope / 1e6:.0f} | {k64:.1f} |")


if __name__ == "__main__":
    main()
