#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc traffic passes: ONE process, eager launches, a few SDXL denoising steps at the bench
configuration (batch 2, 128x128 latents) and nothing else -- no HIP graph, no tuning, no decode.  (The full bench.py under
--pmc dies inside rocprofv3 on this image, with or without graph replay; this keeps the dispatch count small.)"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    import bench
    from diffusers_amd import factory, init as dinit
    from diffusers_amd.pipelines import StableDiffusionXLPipeline
    from diffusers_amd.schedulers import EulerDiscreteScheduler
    dev = torch.device("cuda", 0)
    unet, _ = factory.build_unet(dinit.SDXL_UNET, seed=0, device=dev, init_device=str(dev))
    pipe = StableDiffusionXLPipeline(vae=None, unet=unet, scheduler=EulerDiscreteScheduler(**factory.SDXL_SCHEDULER))
    inp = bench.synth_inputs(1, False, dev)
    pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], dim=0).contiguous()
    te = torch.cat([inp["negative_pooled"], inp["pooled"]], dim=0)
    ids = torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]], device=dev).repeat(2, 1)
    cond = unet.precompute_conditioning(pe, {"text_embeds": te, "time_ids": ids})
    pipe.scheduler.set_timesteps(50, device=dev)
    lat = inp["latents"].clone()
    pipe.scheduler.reset(0)
    for _ in range(steps):
        pipe._step(lat, cond, bench.GUIDANCE, True)
    torch.cuda.synchronize()
    print("pmc_one_step: done", steps, flush=True)


if __name__ == "__main__":
    main()
