"""Round-2 follow-up microbenchmarks (MI355X): the 8-wave 128x128 GEMM tile against every other variant on the SDXL
shapes that dominate the denoising step, and 64- vs 128-query attention workgroups.  Prints JSON lines."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from diffusers_amd import _lib as L  # noqa: E402
from diffusers_amd import ops  # noqa: E402
from tools.bench_kernels_r2 import rnd, timeit  # noqa: E402
import tools.bench_kernels_r2 as K  # noqa: E402


def main():
    K.FLUSH = torch.empty(320 << 20, dtype=torch.uint8, device="cuda")
    shapes = [("to_out 1280", 2048, 1280, 1280, True, 0), ("ff_down 1280", 2048, 1280, 5120, True, 0),
              ("qk 1280", 2048, 2560, 1280, False, 0), ("geglu 1280", 2048, 10240, 1280, False, L.ACT_GEGLU),
              ("to_out 640", 8192, 640, 640, True, 0), ("ff_down 640", 8192, 640, 2560, True, 0),
              ("geglu 640", 8192, 5120, 640, False, L.ACT_GEGLU)]
    for name, M, N, Kd, res, act in shapes:
        x, w, b = rnd((M, Kd)), rnd((N, Kd), Kd ** -0.5), rnd((N,))
        r = rnd((M, N)) if res else None
        rows = []
        for tile in range(1, 9):
            for st in range(1, 6):
                try:
                    tmin, tmed = timeit(lambda: ops.linear(x, w, bias=b, residual=r, act=act, tile=tile, staging=st), iters=8, warm=1)
                except RuntimeError:
                    continue
                rows.append((tmin, tmed, L.TILE_NAMES[tile], st))
        rows.sort()
        flops = 2.0 * M * N * Kd
        emit = lambda rec: print(json.dumps(rec), flush=True)  # noqa: E731
        emit({"op": "gemm", "name": name, "best": [[t[2], t[3], round(t[0], 1)] for t in rows[:4]],
              "w8": [[t[2], t[3], round(t[0], 1)] for t in rows if t[2] == "128x128w8"],
              "best_tflops": round(flops / rows[0][0] / 1e6, 1)})
    for name, B, H, S, Skv in [("sdxl self 1024", 2, 20, 1024, 1024), ("sdxl self 4096", 2, 10, 4096, 4096),
                               ("sdxl cross 1024", 2, 20, 1024, 77), ("sdxl cross 4096", 2, 10, 4096, 77),
                               ("sd15 self 4096 b2h8 (d40->64)", 2, 8, 4096, 4096)]:
        D = 64
        inner = H * D
        sa = ((Skv + 15) // 16) * 16
        q, k, vt = rnd((B * S, inner)), rnd((B * sa, inner)), rnd((inner, B * sa))
        def run(qb):
            return ops.attention(q, k, vt, B=B, H=H, D=D, Sq=S, Skv=Skv, Skv_alloc=sa, q_row_stride=inner, k_row_stride=inner,
                                 q_batch_stride=S * inner, k_batch_stride=sa * inner, vt_ld=B * sa, vt_batch_stride=sa,
                                 q_block=qb)
        ref = run(128)
        rec = {"op": "attn_qblock", "name": name, "tflop": round(4.0 * B * H * S * Skv * D / 1e12, 4)}
        for qb in (128, 64, 0):
            same = bool(torch.equal(run(qb), ref))
            tmin, tmed = timeit(lambda: run(qb), iters=10, warm=2)
            rec[f"q{qb}_us"] = round(tmin, 1)
            rec[f"q{qb}_same"] = same
        print(json.dumps(rec), flush=True)


def attn_pv_delay():
    K.FLUSH = torch.empty(320 << 20, dtype=torch.uint8, device="cuda")
    for name, B, H, S, Skv, D in [("sdxl self 1024", 2, 20, 1024, 1024, 64), ("sdxl self 4096", 2, 10, 4096, 4096, 64),
                                  ("sdxl cross 1024", 2, 20, 1024, 77, 64), ("sdxl cross 4096", 2, 10, 4096, 77, 64),
                                  ("sd15 self 4096 (d40->64)", 2, 8, 4096, 4096, 64), ("flux joint", 1, 24, 4608, 4608, 128),
                                  ("wan slice", 1, 12, 8192, 8192, 128)]:
        inner = H * D
        sa = ((Skv + 15) // 16) * 16
        q, k, vt = rnd((B * S, inner)), rnd((B * sa, inner)), rnd((inner, B * sa))
        def run(pd):
            return ops.attention(q, k, vt, B=B, H=H, D=D, Sq=S, Skv=Skv, Skv_alloc=sa, q_row_stride=inner, k_row_stride=inner,
                                 q_batch_stride=S * inner, k_batch_stride=sa * inner, vt_ld=B * sa, vt_batch_stride=sa,
                                 pv_delay=pd)
        ref = run(-1)
        rec = {"op": "attn_pv_delay", "name": name, "tflop": round(4.0 * B * H * S * Skv * D / 1e12, 4),
               "same": bool(torch.equal(run(1), ref))}
        for rep in range(2):            # interleaved A/B
            for pd in (-1, 1):
                tmin, tmed = timeit(lambda: run(pd), iters=10, warm=2)
                key = "plain_us" if pd < 0 else "delayed_us"
                rec[key] = round(min(rec.get(key, 1e9), tmin), 1)
        rec["speedup"] = round(rec["plain_us"] / rec["delayed_us"], 3)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "attn":
        attn_pv_delay()
    else:
        main()
