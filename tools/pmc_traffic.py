#!/usr/bin/env python
"""HBM-side traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE in one, WRITE_SIZE in the other: they do not fit
one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") of `bench.py --no-graph --steps 1 --warmup 0 ...`.

Corrections applied exactly as the guide's HBM section prescribes: FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950
FETCH_SIZE counts 128-byte requests as 64 bytes for wide coalesced streams (every load of these kernels is 16 B / lane),
so fetched bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is taken as reported (uncalibrated, stated in the output).

usage: pmc_traffic.py <fetch_dir> <write_dir> <out.md> <out.json> [algorithmic_bytes_per_igemm_launch [hoisted igemm dispatches to set aside]]
Writes the per-kernel-family table (markdown) and the JSON bench.py reads for roofline.traffic."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path


def family(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if "igemm2_bf16_kernel" in name or "gemm3_bf16_kernel" in name:   # K2 / K1 / K3 count with the implicit-GEMM family (one roofline entry)
        return "igemm_bf16_kernel"
    for key in ("igemm_bf16_kernel", "attn2_fwd_kernel", "attn_fwd_kernel", "layernorm_kernel", "gn_apply_kernel", "gn_stats_kernel",
                "linear_small_m_kernel", "conv_thin_in_kernel", "conv_thin_out_kernel", "softmax_rows_kernel",
                "euler_step_kernel", "euler_scale_input_kernel"):
        if key in name:
            return key
    if "at::native" in name or name.startswith("at::"):
        return "torch:*"
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:60]


def collect(root: Path, counter: str, skip_igemm: int = 0):
    """``skip_igemm``: the first N implicit-GEMM dispatches (in dispatch order) are the hoisted, step-invariant cross-attention
    K / V^T projections of `precompute_conditioning` (140 small launches for SDXL): they are tallied as their own family so that
    the per-launch mean of `igemm_bf16_kernel` is over the launches of the denoising steps only -- the population
    bench.py's algorithmic bytes per launch are averaged over."""
    acc = defaultdict(lambda: [0, 0.0])
    rows = []
    for f in root.rglob("*counter_collection.csv"):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    rows.append(r)
    key = "Dispatch_Id" if rows and "Dispatch_Id" in rows[0] else None
    if key:
        rows.sort(key=lambda r: int(r[key]))
    seen = 0
    for r in rows:
        fam = family(r["Kernel_Name"])
        if fam == "igemm_bf16_kernel" and key:
            seen += 1
            if seen <= skip_igemm:
                fam = "igemm_bf16_kernel (hoisted K / V^T projections, outside the step)"
        a = acc[fam]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def is_igemm(name: str) -> bool:
    return any(k in name for k in ("igemm_bf16_kernel", "igemm2_bf16_kernel", "gemm3_bf16_kernel", "gemm3_geglu_kernel"))


def short_kernel(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    name = name if cut < 0 else name[:cut]
    for ns in ("da_gemm2::", "da_gemm3::", "da_gemm::"):
        name = name.replace(ns, "")
    return name.replace(" ", "")


def per_population(root_f: Path, root_w: Path, log_path: Path, skip: int):
    """Round 6: one row per launch POPULATION.  The PMC rows carry kernel names only; tools/pmc_one_step.py logs
    every implicit-GEMM launch of the steps in issue order (population label, algorithmic bytes); the i-th implicit-GEMM dispatch
    behind the `skip` hoisted ones IS the i-th logged launch (one process, one stream, eager).  Returns None when the two
    sequences do not have the same length in both passes (then only the per-family table is written)."""
    log = json.loads(log_path.read_text())

    def seq(root, counter):
        rows = []
        for f in root.rglob("*counter_collection.csv"):
            with open(f, newline="") as fh:
                rows += [r for r in csv.DictReader(fh) if r["Counter_Name"] == counter and is_igemm(r["Kernel_Name"])]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        return rows[skip:]
    fe, wr = seq(root_f, "FETCH_SIZE"), seq(root_w, "WRITE_SIZE")
    if len(fe) != len(log) or len(wr) != len(log):
        print(f"per-population table skipped: {len(log)} logged launches, {len(fe)} / {len(wr)} implicit-GEMM dispatches in the passes")
        return None
    pops = {}
    for ent, rf, rw in zip(log, fe, wr):
        a = pops.setdefault((ent["pop"], short_kernel(rf["Kernel_Name"])), [0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += 2.0 * float(rf["Counter_Value"]) * 1024.0
        a[2] += float(rw["Counter_Value"]) * 1024.0
        a[3] += ent["bytes"]
        a[4] += ent["flop"]
    return pops


def main():
    fetch_dir, write_dir, out_md, out_json = (Path(p) for p in sys.argv[1:5])
    algo = float(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5] else None
    skip = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    log_path = Path(sys.argv[7]) if len(sys.argv) > 7 else None
    steps = int(sys.argv[8]) if len(sys.argv) > 8 else 2
    fe, wr = collect(fetch_dir, "FETCH_SIZE", skip), collect(write_dir, "WRITE_SIZE", skip)
    fams = sorted(set(fe) | set(wr), key=lambda k: -(2 * fe.get(k, [0, 0])[1] + wr.get(k, [0, 0])[1]))
    lines = ["# HBM-side traffic per launch, SDXL denoising step (eager launches, rocprofv3 --pmc; FETCH and WRITE in separate passes)",
             "",
             "fetched = 2 x FETCH_SIZE KiB (gfx950 wide-stream correction, MI355X_MICROARCH.md HBM section); written = "
             "WRITE_SIZE KiB as reported (uncalibrated).  Per launch = counter sum / dispatches of the family.", "",
             "| kernel family | dispatches | fetched MB / launch | written MB / launch | total MB / launch | total GB (run) |",
             "|---|---:|---:|---:|---:|---:|"]
    rec = {}
    for k in fams:
        nf, sf = fe.get(k, [0, 0.0])
        nw, sw = wr.get(k, [0, 0.0])
        n = max(nf, nw)
        if n == 0:
            continue
        fb = 2.0 * sf * 1024.0 / max(nf, 1)
        wb = sw * 1024.0 / max(nw, 1)
        rec[k] = {"dispatches": n, "fetched_bytes_per_launch": fb, "written_bytes_per_launch": wb}
        if (2 * sf + sw) * 1024 < 1e6:
            continue
        lines.append(f"| `{k}` | {n} | {fb / 1e6:.3f} | {wb / 1e6:.3f} | {(fb + wb) / 1e6:.3f} | {(fb * nf + wb * nw) / 1e9:.2f} |")
    ig = rec.get("igemm_bf16_kernel")
    stamp = Path(__file__).resolve().parent.parent / "diffusers_amd" / "_C" / "build.stamp"
    out = {"build_fingerprint": stamp.read_text().strip()[:16] if stamp.exists() else "unknown",   # as bench.build_fingerprint()
           "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, 2x FETCH_SIZE correction) on the SDXL "
                     f"denoising-step kernels launched eagerly (tools/pmc_one_step.py); profiles/{out_md.name}",
           "measured_on": __import__("time").strftime("%Y-%m-%d"),
           "unit": "bytes", "families": rec}
    out["source"] = out["source"].replace("igemm_bf16_kernel", "igemm")
    if ig:
        tot = ig["fetched_bytes_per_launch"] + ig["written_bytes_per_launch"]
        out["igemm_bytes_per_launch"] = tot
        lines += ["", f"igemm_bf16_kernel: {tot / 1e6:.3f} MB of HBM-side traffic per launch"
                  + (f" against {algo / 1e6:.3f} MB of algorithmic bytes (unique A + W + C per launch, averaged over the "
                     f"launches of a denoising step): ratio {tot / algo:.2f}x" if algo else "")]
        if algo:
            out["igemm_algorithmic_bytes_per_launch"] = algo
    pops = per_population(fetch_dir, write_dir, log_path, skip) if log_path is not None and log_path.exists() else None
    if pops:
        lines += ["", "## Implicit-GEMM launches by population (shape x role x kernel instantiation)", "",
                  "fetched / written as above; algorithmic = each operand and the output (and the residual, where fused) once.  "
                  "`floor x8` = what eight non-coherent L2s must fetch at least for this launch's XCD tile rectangles is NOT computed here; "
                  "the ratio column is total / algorithmic.", "",
                  "| population | kernel | launches / step | fetched MB | written MB | total MB | algorithmic MB | ratio | GFLOP |",
                  "|---|---|---:|---:|---:|---:|---:|---:|---:|"]
        fam_tot = {}
        for (pop, kern), (n, fb, wb, ab, fl) in sorted(pops.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
            lines.append(f"| {pop} | `{kern}` | {n / steps:.1f} | {fb / n / 1e6:.2f} | {wb / n / 1e6:.2f} | {(fb + wb) / n / 1e6:.2f} | "
                         f"{ab / n / 1e6:.2f} | {(fb + wb) / ab:.2f} | {fl / n / 1e9:.1f} |")
            fam = kern.split("<")[0]
            t = fam_tot.setdefault(fam, [0, 0.0, 0.0])
            t[0] += n
            t[1] += fb + wb
            t[2] += ab
        lines += ["", "| kernel family | launches / step | total MB / launch | algorithmic MB / launch | ratio |", "|---|---:|---:|---:|---:|"]
        for fam, (n, tb, ab) in sorted(fam_tot.items(), key=lambda kv: -kv[1][1]):
            lines.append(f"| `{fam}` | {n / steps:.1f} | {tb / n / 1e6:.2f} | {ab / n / 1e6:.2f} | {tb / ab:.2f} |")
        out["igemm_family_ratios"] = {fam: {"launches_per_step": n / steps, "traffic_bytes_per_launch": tb / n,
                                            "algorithmic_bytes_per_launch": ab / n, "ratio": tb / ab} for fam, (n, tb, ab) in fam_tot.items()}
        allt = [sum(v[i] for v in fam_tot.values()) for i in range(3)]
        out["igemm_ratio_by_logged_algorithmic_bytes"] = allt[1] / allt[2]
        lines += ["", f"all implicit-GEMM launches of the steps: {allt[1] / allt[0] / 1e6:.2f} MB per launch against {allt[2] / allt[0] / 1e6:.2f} MB "
                  f"algorithmic = {allt[1] / allt[2]:.2f}x"]
    out_md.write_text("\n".join(lines) + "\n")
    out_json.write_text(json.dumps(out, indent=1) + "\n")
    print("\n".join(lines[-3:]))


if __name__ == "__main__":
    main()
