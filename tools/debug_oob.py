#!/usr/bin/env python
"""Debug: does any launch of AutoencoderKL.decode (or of an SDXL U-Net step with --unet) write outside the tensors it was given?
Every `torch.empty` / `torch.zeros` of the run is carved out of one arena with 64 KiB canary gaps either side; after the run the
gaps are checked and the allocations next to a damaged gap are named (shape, dtype, the ops.* frame that made them)."""
import sys
import traceback
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import factory, init as dinit  # noqa: E402

dev = torch.device("cuda", 0)
GAP = 64 << 10
ARENA_BYTES = (int(sys.argv[sys.argv.index("--gb") + 1]) if "--gb" in sys.argv else 48) << 30
PATTERN = 0x5A
arena = None
allocs = []     # (offset, nbytes, shape, dtype, where)
cursor = [GAP]
real_empty, real_zeros = torch.empty, torch.zeros


def carve(shape, dtype, zero):
    if isinstance(shape, int):
        shape = (shape,)
    shape = tuple(int(s) for s in shape)
    n = 1
    for s in shape:
        n *= s
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    off = (cursor[0] + 255) & ~255
    if off + nbytes + GAP > ARENA_BYTES:
        raise RuntimeError("arena exhausted")
    cursor[0] = off + nbytes + GAP
    fr = [f for f in traceback.extract_stack(limit=8) if "diffusers_amd" in f.filename]
    where = f"{Path(fr[-1].filename).name}:{fr[-1].lineno} {fr[-1].name}" if fr else "?"
    allocs.append((off, nbytes, shape, dtype, where))
    t = arena[off:off + nbytes].view(dtype).view(shape) if nbytes else real_empty(shape, dtype=dtype, device=dev)
    if zero and nbytes:
        t.zero_()
    return t


def patched(zero):
    def f(*size, **kw):
        device = kw.get("device")
        if arena is None or device is None or torch.device(device).type != "cuda":
            return (real_zeros if zero else real_empty)(*size, **kw)
        shape = size[0] if len(size) == 1 and not isinstance(size[0], int) else size
        return carve(shape, kw.get("dtype", torch.float32), zero)
    return f


def check(tag):
    torch.cuda.synchronize()
    bad = 0
    edges = [(0, GAP, None, 0)] + [(off + nb, GAP, i, i + 1) for i, (off, nb, *_rest) in enumerate(allocs)]
    for start, length, before, after in edges:
        start = (start + 0) if before is None else start
        end = min(start + length, (allocs[after][0] if after is not None and after < len(allocs) else ARENA_BYTES))
        if end <= start:
            continue
        seg = arena[start:end]
        nz = (seg != PATTERN).nonzero()
        if nz.numel():
            bad += 1
            lo, hi = int(nz.min()), int(nz.max())
            b = allocs[before] if before is not None else None
            a = allocs[after] if after is not None and after < len(allocs) else None
            print(f"OOB {tag}: gap after alloc #{before} damaged: bytes [{lo}, {hi}] of the gap ({int(nz.numel())} bytes changed)\n"
                  f"      alloc before: {b[2:] if b else None}\n      alloc after:  {a[2:] if a else None}", flush=True)
    print(f"RESULT {tag}: {len(allocs)} allocations, {cursor[0] / 2**30:.2f} GiB carved, damaged gaps: {bad}", flush=True)


arena = real_empty(ARENA_BYTES, dtype=torch.uint8, device=dev)
arena.fill_(PATTERN)
torch.empty, torch.zeros = patched(False), patched(True)
try:
    if "--unet" in sys.argv:
        import bench
        from diffusers_amd.pipelines import StableDiffusionXLPipeline
        from diffusers_amd.schedulers import EulerDiscreteScheduler
        unet, _ = factory.build_unet(dinit.SDXL_UNET, seed=0, device=dev, init_device=str(dev))
        allocs.clear()
        pipe = StableDiffusionXLPipeline(vae=None, unet=unet, scheduler=EulerDiscreteScheduler(**factory.SDXL_SCHEDULER))
        inp = bench.synth_inputs(1, False, dev)
        pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], dim=0).contiguous()
        te = torch.cat([inp["negative_pooled"], inp["pooled"]], dim=0)
        ids = torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]], device=dev).repeat(2, 1)
        cond = unet.precompute_conditioning(pe, {"text_embeds": te, "time_ids": ids})
        pipe.scheduler.set_timesteps(50, device=dev)
        lat = inp["latents"].clone()
        pipe.scheduler.reset(0)
        pipe._step(lat, cond, bench.GUIDANCE, True)
        check("SDXL U-Net step (eager)")
    else:
        vae, _ = factory.build_vae(dinit.SDXL_VAE, seed=1, device=dev, init_device=str(dev))
        torch.cuda.synchronize()
        # (the weights were carved too; what follows is the decode's own allocations)
        n0 = len(allocs)
        g = torch.Generator("cpu").manual_seed(3)
        z = torch.randn((1, 4, 128, 128), generator=g).to(torch.bfloat16).to(dev)
        vae.decode(z, return_dict=False, latents_div=0.13025, postprocess="pt")
        print("decode allocations:", len(allocs) - n0)
        check("AutoencoderKL.decode 1024x1024 (eager)")
finally:
    torch.empty, torch.zeros = real_empty, real_zeros
