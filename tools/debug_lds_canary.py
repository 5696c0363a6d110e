#!/usr/bin/env python
"""Round 6 debug (the open item of pipelines._exclusive_decode): does any kernel write LDS outside its workgroup's allocation?  Canary
workgroups of ANOTHER kernel (tools/lds_canary.hip: a few KiB of LDS filled with a pattern and re-read for a millisecond) are kept
resident on every CU from a second stream while the first stream runs (a) a positive control that aims plain ds_write / LDS-DMA at
offsets beyond its own 1 KiB, (b) eager VAE decodes, (c) eager U-Net steps, (d) the eight-phase GEMM tiles and D = 128 attention.
usage: debug_lds_canary.py [out.json]"""
import ctypes as C
import json
import subprocess
import sys
import threading
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from diffusers_amd import factory, init as dinit, ops  # noqa: E402
from diffusers_amd import _lib as L  # noqa: E402

SO = "/tmp/liblds_canary.so"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", str(ROOT / "tools" / "lds_canary.hip"), "-o", SO], check=True)
lib = C.CDLL(SO)
lib.lds_canary_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_void_p]
lib.lds_overflow_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_void_p]
dev = torch.device("cuda", 0)
bf16 = torch.bfloat16
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
TICK = 100_000   # wall_clock64 ticks per millisecond


def canary_loop(stop, bad, kib=4, blocks=2048, ms=1.0):
    n = 0
    with torch.cuda.stream(sb):
        while not stop.is_set():
            rc = lib.lds_canary_launch(bad.data_ptr(), blocks, kib, int(ms * TICK), sb.cuda_stream)
            assert rc == 0, rc
            n += 1
            if n % 8 == 0:
                sb.synchronize()       # bounded queue depth
        sb.synchronize()
    return n


def with_canary(name, work, kib=4, repeat=1):
    bad = torch.zeros(8, dtype=torch.int32, device=dev)
    stop = threading.Event()
    count = {}
    th = threading.Thread(target=lambda: count.setdefault("n", canary_loop(stop, bad, kib)))
    th.start()
    time.sleep(0.02)
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        for _ in range(repeat):
            work()
        sa.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    torch.cuda.synchronize()
    b = bad.cpu().tolist()
    rec = {"case": name, "canary_kib": kib, "canary_launches": count.get("n"), "seconds": round(dt, 3), "corrupted_words": b[0],
           "first": {"block": b[1], "word": b[2], "got": hex(b[3] & 0xffffffff), "want": hex(b[4] & 0xffffffff)} if b[0] else None}
    print("RESULT", json.dumps(rec), flush=True)
    return rec


def main():
    out = []
    out.append(with_canary("control: nothing on the other stream", lambda: time.sleep(0.05)))
    src = torch.full((4096,), 0x5a5a5a5a, dtype=torch.int32, device=dev)
    for dma in (0, 1):
        for off in (2048, 16384, 65536, 131072):
            out.append(with_canary(f"positive control: {'LDS-DMA' if dma else 'ds_write_b32'} at byte {off} of a 1 KiB allocation",
                                   lambda: lib.lds_overflow_launch(src.data_ptr(), 1024, off, dma, 20 * TICK, sa.cuda_stream)))
    vae, _ = factory.build_vae(dinit.SDXL_VAE, seed=1, device=dev, init_device=str(dev))
    z = torch.randn((1, 4, 128, 128), generator=torch.Generator("cpu").manual_seed(3)).to(bf16).to(dev)
    with torch.cuda.stream(sa):
        vae.decode(z, return_dict=False)
        sa.synchronize()
    for kib in (4, 12):
        out.append(with_canary("SDXL AutoencoderKL.decode 1024^2, eager", lambda: vae.decode(z, return_dict=False), kib=kib, repeat=4))
    del vae
    from diffusers_amd.pipelines import StableDiffusionXLPipeline
    from diffusers_amd.schedulers import EulerDiscreteScheduler
    unet, _ = factory.build_unet(dinit.SDXL_UNET, seed=0, device=dev, init_device=str(dev))
    pipe = StableDiffusionXLPipeline(vae=None, unet=unet, scheduler=EulerDiscreteScheduler(**factory.SDXL_SCHEDULER))
    inp = bench.synth_inputs(1, False, dev)
    pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], dim=0).contiguous()
    te = torch.cat([inp["negative_pooled"], inp["pooled"]], dim=0)
    ids = torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]], device=dev).repeat(2, 1)
    with torch.cuda.stream(sa):
        cond = unet.precompute_conditioning(pe, {"text_embeds": te, "time_ids": ids})
        pipe.scheduler.set_timesteps(50, device=dev)
        lat = inp["latents"].clone()
        pipe.scheduler.reset(0)
        pipe._step(lat, cond, bench.GUIDANCE, True)
        sa.synchronize()

    def steps():
        pipe.scheduler.reset(0)
        for _ in range(3):
            pipe._step(lat, cond, bench.GUIDANCE, True)
    for kib in (4, 12):
        out.append(with_canary("SDXL U-Net denoising steps, eager", steps, kib=kib, repeat=2))
    del unet, pipe
    # the eight-phase tiles and D = 128 attention (Flux / Wan shapes)
    g = torch.Generator("cpu").manual_seed(5)
    x, w = (torch.randn((4608, 3072), generator=g) * 1.0).to(bf16).to(dev), (torch.randn((12288, 3072), generator=g) * 0.02).to(bf16).to(dev)
    out.append(with_canary("k3:256x256 4608 x 12288 x 3072 (+ tanh-GELU)",
                           lambda: ops.linear(x, w, act=L.ACT_GELU_TANH, tile=L.TILE_K3_256x256, staging=L.STAGE_LDS_DIRECT), repeat=20))
    B, H, S, D = 1, 24, 4608, 128
    q = torch.randn((B, S, H * D), generator=g).to(bf16).to(dev)
    k = torch.randn((B, S, H * D), generator=g).to(bf16).to(dev)
    vt = torch.randn((H * D, B * S), generator=g).to(bf16).to(dev)
    out.append(with_canary("flash attention D 128, S 4608, 24 heads", lambda: ops.attention(q, k, vt, B=B, H=H, D=D, Sq=S, Skv=S, Skv_alloc=S, q_row_stride=H * D,
                                                                                         k_row_stride=H * D, q_batch_stride=S * H * D, k_batch_stride=S * H * D,
                                                                                         vt_ld=B * S, vt_batch_stride=S), repeat=20))
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text("\n".join(json.dumps(r) for r in out) + "\n")


if __name__ == "__main__":
    main()
