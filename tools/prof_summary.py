#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd .db or *_kernel_stats.csv) as a small markdown table
(per-kernel calls / total / average / share), the form committed under profiles/."""
import csv
import sqlite3
import sys
from pathlib import Path


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if name.startswith("at::native") or "at::native" in name[:40]:
        return "torch:" + name.split("<")[0].split("::")[-1][:50]
    cut = name.find("(")
    return name if cut < 0 else name[:cut]


def rows_from_db(path):
    c = sqlite3.connect(path)
    return [(r[0], int(r[1]), float(r[2]) / 1e3, float(r[3]) / 1e3) for r in
            c.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name")]


def rows_from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
    return out


def main():
    src = Path(sys.argv[1])
    title = sys.argv[2] if len(sys.argv) > 2 else src.name
    rows = rows_from_db(src) if src.suffix == ".db" else rows_from_csv(src)
    agg = {}
    for name, calls, tot, _ in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0])
        a[0] += calls
        a[1] += tot
    total = sum(v[1] for v in agg.values())
    print(f"# {title}\n")
    print(f"total kernel time {total / 1e3:.1f} ms over {sum(v[0] for v in agg.values())} dispatches\n")
    print("| kernel | calls | total ms | avg us | share |")
    print("|---|---:|---:|---:|---:|")
    for k, (calls, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if tot / total < 0.0005:
            continue
        print(f"| `{k}` | {calls} | {tot / 1e3:.2f} | {tot / calls:.1f} | {100 * tot / total:.1f}% |")


if __name__ == "__main__":
    main()
