#!/usr/bin/env python
"""Aggregate a rocprofv3 *_counter_collection.csv per (kernel, grid size): dispatch count and mean of every counter.
usage: pmc_agg.py <dir> > summary.csv   (walks <dir> for *counter_collection.csv)"""
import csv
import sys
from collections import defaultdict
from pathlib import Path


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:90]


def main():
    root = Path(sys.argv[1])
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in root.rglob("*counter_collection.csv"):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = (short(r["Kernel_Name"]), r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
                a = acc[k][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    counters = sorted({c for v in acc.values() for c in v})
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "grid", "lds", "dispatches"] + [f"mean_{c}" for c in counters])
    for k, v in sorted(acc.items(), key=lambda kv: -max(a[1] for a in kv[1].values())):
        n = max(a[0] for a in v.values())
        w.writerow(list(k) + [n] + [f"{v[c][1] / v[c][0]:.1f}" if c in v else "" for c in counters])


if __name__ == "__main__":
    main()
