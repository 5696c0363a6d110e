#!/usr/bin/env python
"""Per-shape ceiling table: the engine's igemm launch vs the vendor libraries behind the reference's
``F.linear`` / ``F.conv2d`` (hipBLASLt / MIOpen through PyTorch-ROCm) on the SDXL shapes that carry the denoising step,
same box, same random data, HIP events on the launch stream.  Two regimes per shape: "cold" (a 320 MiB memset evicts L2 and
the Infinity Cache before every launch: how a weight is met inside the denoising loop) and "warm" (back-to-back launches).
One JSON object per line + a markdown table (argv[1], default gpurun_out/ceiling_table.md).

Any row where the vendor library wins is a bug list for csrc/gemm_kernel.cuh."""
from __future__ import annotations

import json
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import _lib as L  # noqa: E402
from diffusers_amd import ops  # noqa: E402

bf16 = torch.bfloat16
DEV = "cuda"
FLUSH = None


def rnd(shape, scale=1.0):
    return (torch.randn(shape, device=DEV) * scale).to(bf16)


def timeit(fn, iters=12, warm=3, flush=True):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if flush:
            FLUSH.zero_()
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


def chain_us(fn, n=40):
    """Average microseconds per launch of n back-to-back launches (launch gaps included, event overhead amortised)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    global FLUSH
    out_md = Path(sys.argv[1] if len(sys.argv) > 1 else ROOT / "gpurun_out" / "ceiling_table.md")
    FLUSH = torch.empty(320 << 20, dtype=torch.uint8, device=DEV)
    rows = []

    # name, M, N, K, bias, residual, act
    lin = [("to_out / proj 1280 (+bias +res)", 2048, 1280, 1280, True, True, 0),
           ("to_q 1280 (no bias)", 2048, 1280, 1280, False, False, 0),
           ("q|k fused 1280", 2048, 2560, 1280, False, False, 0),
           ("GEGLU up 1280", 2048, 10240, 1280, True, False, L.ACT_GEGLU),
           ("FF down 1280 (+bias +res)", 2048, 1280, 5120, True, True, 0),
           ("to_out 640 (+bias +res)", 8192, 640, 640, True, True, 0),
           ("GEGLU up 640", 8192, 5120, 640, True, False, L.ACT_GEGLU),
           ("FF down 640 (+bias +res)", 8192, 640, 2560, True, True, 0)]
    for name, M, N, K, has_b, has_r, act in lin:
        x, w = rnd((M, K)), rnd((N, K), K ** -0.5)
        b = rnd((N,)) if has_b else None
        r = rnd((M, N if act == 0 else N // 2)) if has_r else None
        if act:
            wp, bp = ops.pack_geglu(w, b)
            ours = lambda: ops.linear(x, wp, bias=bp, act=act)  # noqa: E731

            def ref():
                h = F.linear(x, w, b)
                a, g = h.chunk(2, dim=-1)
                return a * F.gelu(g)
            ref_mm = lambda: F.linear(x, w, b)  # noqa: E731
        else:
            ours = lambda: ops.linear(x, w, bias=b, residual=r)  # noqa: E731
            if r is not None:
                ref = lambda: torch.addmm(b, x, w.t()) + r if b is not None else x @ w.t() + r  # noqa: E731
            else:
                ref = lambda: F.linear(x, w, b)  # noqa: E731
            ref_mm = lambda: F.linear(x, w, b)  # noqa: E731
        ours()  # tunes the shape if the shipped table does not hold it
        rec = {"op": "linear", "name": name, "M": M, "N": N, "K": K, "gflop": round(2e-9 * M * N * K, 2)}
        rec["ours_cold_us"], _ = timeit(ours)
        rec["ours_warm_us"], _ = timeit(ours, flush=False)
        rec["ours_chain_us"] = chain_us(ours)
        rec["lib_gemm_cold_us"], _ = timeit(ref_mm)
        rec["lib_gemm_warm_us"], _ = timeit(ref_mm, flush=False)
        rec["lib_gemm_chain_us"] = chain_us(ref_mm)
        rec["lib_full_cold_us"], _ = timeit(ref)          # the reference's op chain for the same result (epilogue unfused)
        rec["lib_full_chain_us"] = chain_us(ref)
        rows.append(rec)
        print(json.dumps(rec), flush=True)

    # name, B, H, W, Cin, Cout
    conv = [("conv3x3 320 @128^2", 2, 128, 128, 320, 320), ("conv3x3 640 @64^2", 2, 64, 64, 640, 640),
            ("conv3x3 1280 @32^2", 2, 32, 32, 1280, 1280), ("conv3x3 1920->1280 @32^2", 2, 32, 32, 1920, 1280),
            ("conv3x3 960->640 @64^2", 2, 64, 64, 960, 640),
            ("VAE conv3x3 512 @128^2", 1, 128, 128, 512, 512), ("VAE conv3x3 512 @256^2", 1, 256, 256, 512, 512),
            ("VAE conv3x3 256 @512^2", 1, 512, 512, 256, 256), ("VAE conv3x3 128 @1024^2", 1, 1024, 1024, 128, 128)]
    for name, B, H, W, Ci, Co in conv:
        x = rnd((B, H, W, Ci))
        w4 = rnd((Co, Ci, 3, 3), (9 * Ci) ** -0.5)
        b = rnd((Co,))
        wp = w4.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()
        xn = x.permute(0, 3, 1, 2)                                  # NCHW view of channels-last storage
        w4cl = w4.contiguous(memory_format=torch.channels_last)
        ours = lambda: ops.conv2d_nhwc(x, wp, b, ksize=3)  # noqa: E731
        ref = lambda: F.conv2d(xn, w4cl, b, padding=1)  # noqa: E731
        xc = xn.contiguous()                                        # plain NCHW, as the reference pipeline runs it
        ref_nchw = lambda: F.conv2d(xc, w4, b, padding=1)  # noqa: E731
        ours()
        rec = {"op": "conv", "name": name, "M": B * H * W, "N": Co, "K": 9 * Ci, "gflop": round(2e-9 * B * H * W * Co * 9 * Ci, 2)}
        rec["ours_cold_us"], _ = timeit(ours, iters=8)
        rec["ours_warm_us"], _ = timeit(ours, iters=8, flush=False)
        rec["ours_chain_us"] = chain_us(ours, 20)
        rec["lib_gemm_cold_us"], _ = timeit(ref, iters=8)
        rec["lib_gemm_warm_us"], _ = timeit(ref, iters=8, flush=False)
        rec["lib_gemm_chain_us"] = chain_us(ref, 20)
        rec["lib_full_cold_us"], _ = timeit(ref_nchw, iters=8)
        rec["lib_full_chain_us"] = chain_us(ref_nchw, 20)
        rows.append(rec)
        print(json.dumps(rec), flush=True)

    lines = ["| shape | GFLOP | ours cold us | lib cold us | ours warm us | lib warm us | ours chain us | lib chain us | "
             "lib full-chain us | ours TFLOP/s (chain) | lib TFLOP/s (chain) | ours / lib (chain) |",
             "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for r in rows:
        lines.append(f"| {r['name']} ({r['M']}x{r['N']}x{r['K']}) | {r['gflop']} | {r['ours_cold_us']:.1f} | {r['lib_gemm_cold_us']:.1f} | "
                     f"{r['ours_warm_us']:.1f} | {r['lib_gemm_warm_us']:.1f} | {r['ours_chain_us']:.1f} | {r['lib_gemm_chain_us']:.1f} | "
                     f"{r['lib_full_chain_us']:.1f} | {r['gflop'] / r['ours_chain_us']:.0f} | {r['gflop'] / r['lib_gemm_chain_us']:.0f} | "
                     f"{r['ours_chain_us'] / r['lib_gemm_chain_us']:.2f} |")
    out_md.parent.mkdir(parents=True, exist_ok=True)
    out_md.write_text("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
