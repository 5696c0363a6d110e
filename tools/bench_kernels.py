#!/usr/bin/env python
"""Per-kernel microbenchmarks on the shapes of the BASELINE configs (SDXL U-Net at B=2 / 128x128 latents, SDXL VAE
decoder at 1024x1024): every (tile, staging) variant of da_gemm_bf16 per shape, plus attention / GroupNorm / LayerNorm.
Writes one JSON object per line to gpurun_out/kernels.jsonl (HIP-event timing on the launch stream, min of N)."""
from __future__ import annotations

import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import _lib as L  # noqa: E402
from diffusers_amd import ops  # noqa: E402

bf16 = torch.bfloat16
DEV = "cuda"
OUT = ROOT / "gpurun_out"
OUT.mkdir(exist_ok=True)
VARIANTS = [(t, s) for s in range(1, 6) for t in range(1, 8)]  # unsupported (tile, ring depth) pairs are skipped


def rnd(shape, scale=1.0):
    return (torch.randn(shape, device=DEV) * scale).to(bf16)


FLUSH = None  # set in main(): a 320 MiB buffer zeroed before every timed launch (operands come from HBM)


def timeit(fn, iters=8, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if FLUSH is not None:
            FLUSH.zero_()
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


def main():
    global FLUSH
    lines = []
    if "--warm" not in sys.argv:
        FLUSH = torch.empty(320 << 20, dtype=torch.uint8, device=DEV)

    def emit(rec):
        lines.append(rec)
        print(json.dumps(rec), flush=True)

    # ---- Linear shapes of the SDXL U-Net (B=2) and Flux/Wan-like large ones -------------------------------------
    lin = [("sdxl.to_out/proj 1280", 2048, 1280, 1280, 0), ("sdxl.qk 1280", 2048, 2560, 1280, 0),
           ("sdxl.vT 1280", 1280, 2048, 1280, 0), ("sdxl.geglu 1280", 2048, 10240, 1280, L.ACT_GEGLU),
           ("sdxl.ff_down 1280", 2048, 1280, 5120, 0), ("sdxl.to_out/proj 640", 8192, 640, 640, 0),
           ("sdxl.qk 640", 8192, 1280, 640, 0), ("sdxl.vT 640", 640, 8192, 640, 0),
           ("sdxl.geglu 640", 8192, 5120, 640, L.ACT_GEGLU), ("sdxl.ff_down 640", 8192, 640, 2560, 0),
           ("flux.qkv", 4608, 9216, 3072, 0), ("flux.ff_up", 4096, 12288, 3072, L.ACT_GELU_TANH),
           ("square 4096", 4096, 4096, 4096, 0)]
    for name, M, N, K, act in lin:
        x, w = rnd((M, K)), rnd((N, K), K ** -0.5)
        flops = 2.0 * M * N * K
        best = None
        for tile, st in VARIANTS:
            try:
                tmin, tmed = timeit(lambda: ops.linear(x, w, act=act, tile=tile, staging=st))
            except RuntimeError:
                continue
            rec = {"op": "linear", "name": name, "M": M, "N": N, "K": K, "act": act, "tile": L.TILE_NAMES[tile],
                   "staging": st, "us": round(tmin, 1), "us_med": round(tmed, 1), "tflops": round(flops / tmin / 1e6, 1)}
            emit(rec)
            if best is None or tmin < best["us"]:
                best = rec
        emit({"op": "linear.best", "name": name, **{k: best[k] for k in ("tile", "staging", "us", "tflops")}})

    # ---- Conv shapes: SDXL U-Net resnets (B=2) and the VAE decoder (B=1) --------------------------------------------
    conv = [("unet 320@128", 2, 128, 128, 320, 0, 320, 1, False), ("unet 640@64", 2, 64, 64, 640, 0, 640, 1, False),
            ("unet 1280@32", 2, 32, 32, 1280, 0, 1280, 1, False),
            ("unet cat2560@32", 2, 32, 32, 1280, 1280, 1280, 1, False),
            ("unet cat1920@64", 2, 64, 64, 1280, 640, 640, 1, False),
            ("unet cat960@128", 2, 128, 128, 640, 320, 320, 1, False),
            ("unet down 320", 2, 128, 128, 320, 0, 320, 2, False), ("unet up 1280", 2, 32, 32, 1280, 0, 1280, 1, True),
            ("vae 512@128", 1, 128, 128, 512, 0, 512, 1, False), ("vae 512@256", 1, 256, 256, 512, 0, 512, 1, False),
            ("vae up512@256", 1, 256, 256, 512, 0, 512, 1, True), ("vae 256@512", 1, 512, 512, 256, 0, 256, 1, False),
            ("vae up256@512", 1, 512, 512, 256, 0, 256, 1, True), ("vae 128@1024", 1, 1024, 1024, 128, 0, 128, 1, False)]
    for name, B, H, W, C1, C2, Co, stride, up in conv:
        x1 = rnd((B, H, W, C1))
        x2 = rnd((B, H, W, C2)) if C2 else None
        w = rnd((Co, 9 * (C1 + C2)), (9 * (C1 + C2)) ** -0.5)
        b = rnd((Co,))
        Ho = (2 * H if up else H) // stride
        Wo = (2 * W if up else W) // stride
        flops = 2.0 * B * Ho * Wo * Co * 9 * (C1 + C2)
        best = None
        for tile, st in VARIANTS:
            try:
                tmin, tmed = timeit(lambda: ops.conv2d_nhwc(x1, w, b, ksize=3, x2=x2, stride=stride, up=up, tile=tile,
                                                            staging=st), iters=6)
            except RuntimeError:
                continue
            rec = {"op": "conv3x3", "name": name, "B": B, "H": H, "W": W, "C1": C1, "C2": C2, "Cout": Co,
                   "stride": stride, "up": up, "tile": L.TILE_NAMES[tile], "staging": st, "us": round(tmin, 1),
                   "us_med": round(tmed, 1), "tflops": round(flops / tmin / 1e6, 1)}
            emit(rec)
            if best is None or tmin < best["us"]:
                best = rec
        emit({"op": "conv3x3.best", "name": name, **{k: best[k] for k in ("tile", "staging", "us", "tflops")}})
        del x1, x2, w

    # ---- attention ----------------------------------------------------------------------------------------------------
    for name, B, H, S, Skv, D in [("sdxl self 1024", 2, 20, 1024, 1024, 64), ("sdxl self 4096", 2, 10, 4096, 4096, 64),
                                  ("sdxl cross 1024", 2, 20, 1024, 77, 64), ("sdxl cross 4096", 2, 10, 4096, 77, 64),
                                  ("flux joint", 1, 24, 4608, 4608, 128)]:
        inner = H * D
        sa = ((Skv + 15) // 16) * 16
        q = rnd((B * S, inner))
        k = rnd((B * sa, inner))
        vt = rnd((inner, B * sa))
        fn = lambda: ops.attention(q, k, vt, B=B, H=H, D=D, Sq=S, Skv=Skv, Skv_alloc=sa, q_row_stride=inner,  # noqa: E731
                                   k_row_stride=inner, q_batch_stride=S * inner, k_batch_stride=sa * inner,
                                   vt_ld=B * sa, vt_batch_stride=sa)
        tmin, tmed = timeit(fn)
        flops = 4.0 * B * H * S * Skv * D
        emit({"op": "attention", "name": name, "B": B, "H": H, "Sq": S, "Skv": Skv, "D": D, "us": round(tmin, 1),
              "us_med": round(tmed, 1), "tflops": round(flops / tmin / 1e6, 1)})

    # ---- GroupNorm(+SiLU) / LayerNorm: algorithmic bytes = read once + write once ----------------------------------
    for name, B, HW, Cc in [("unet 320@128", 2, 16384, 320), ("unet 640@64", 2, 4096, 640), ("unet 1280@32", 2, 1024, 1280),
                            ("unet cat2560@32", 2, 1024, 2560), ("vae 512@128", 1, 16384, 512),
                            ("vae 512@256", 1, 65536, 512), ("vae 256@512", 1, 262144, 256),
                            ("vae 128@1024", 1, 1048576, 128)]:
        x = rnd((B, HW, Cc))
        g, b = rnd((Cc,)), rnd((Cc,))
        tmin, tmed = timeit(lambda: ops.group_norm_nhwc(x, g, b, 32, 1e-5, silu=True))
        nbytes = 2.0 * x.numel() * 2
        emit({"op": "groupnorm_silu", "name": name, "B": B, "HW": HW, "C": Cc, "us": round(tmin, 1),
              "us_med": round(tmed, 1), "GBps": round(nbytes / tmin / 1e3, 1),
              "GBps_3pass": round(1.5 * nbytes / tmin / 1e3, 1)})
        del x
    for name, M, Cc in [("sdxl 1280", 2048, 1280), ("sdxl 640", 8192, 640), ("flux 3072", 4608, 3072)]:
        x = rnd((M, Cc))
        g, b = rnd((Cc,)), rnd((Cc,))
        tmin, tmed = timeit(lambda: ops.layer_norm(x, g, b, 1e-5))
        emit({"op": "layernorm", "name": name, "M": M, "C": Cc, "us": round(tmin, 1), "us_med": round(tmed, 1),
              "GBps": round(2.0 * x.numel() * 2 / tmin / 1e3, 1)})

    (OUT / "kernels.jsonl").write_text("\n".join(json.dumps(r) for r in lines) + "\n")


if __name__ == "__main__":
    main()
