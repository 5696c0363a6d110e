#!/usr/bin/env python
"""Debug: where does the two-in-flight leg lose bit-identity?  Phases: warm-up (capture) -> sequential x2 -> concurrent x3, each compared
with the warm-up images; per pipeline max |diff| and the share of differing elements."""
import sys
import threading
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from diffusers_amd import factory, init as dinit, pipelines as P  # noqa: E402
from diffusers_amd.schedulers import EulerDiscreteScheduler  # noqa: E402

dev = torch.device("cuda", 0)
unet, _ = factory.build_unet(dinit.SDXL_UNET, seed=0, device=dev, init_device=str(dev))
vae, _ = factory.build_vae(dinit.SDXL_VAE, seed=1, device=dev, init_device=str(dev))
if "--headline-first" in sys.argv:      # as bench.py: a third pipeline of the default domain has run before
    p0 = P.StableDiffusionXLPipeline(vae=vae, unet=unet, scheduler=EulerDiscreteScheduler(**factory.SDXL_SCHEDULER))
    i0 = bench.synth_inputs(1, False, dev)
    p0(prompt_embeds=i0["prompt_embeds"], negative_prompt_embeds=i0["negative_prompt_embeds"], pooled_prompt_embeds=i0["pooled"],
       negative_pooled_prompt_embeds=i0["negative_pooled"], latents=i0["latents"].clone(), num_inference_steps=50, guidance_scale=bench.GUIDANCE,
       height=1024, width=1024, output_type="pt")
pipes = [P.StableDiffusionXLPipeline(vae=vae, unet=unet, scheduler=EulerDiscreteScheduler(**factory.SDXL_SCHEDULER)) for _ in range(2)]
inp = bench.synth_inputs(2, False, dev)
streams = [torch.cuda.Stream() for _ in range(2)]
outs = {}
otype = "latent" if "--latent" in sys.argv else "pt"


def work(i, count):
    P.STREAM_DOMAIN.tag = 101 + i
    with torch.cuda.stream(streams[i]):
        for _ in range(count):
            outs[i] = pipes[i](prompt_embeds=inp["prompt_embeds"][i:i + 1], negative_prompt_embeds=inp["negative_prompt_embeds"][i:i + 1],
                               pooled_prompt_embeds=inp["pooled"][i:i + 1], negative_pooled_prompt_embeds=inp["negative_pooled"][i:i + 1],
                               latents=inp["latents"][i:i + 1].clone(), num_inference_steps=50, guidance_scale=bench.GUIDANCE, height=1024,
                               width=1024, output_type=otype).images.clone()
        streams[i].synchronize()


def run(concurrent, count):
    th = [threading.Thread(target=work, args=(i, count)) for i in range(2)]
    if concurrent:
        [t.start() for t in th]
        [t.join() for t in th]
    else:
        for t in th:
            t.start()
            t.join()
    torch.cuda.synchronize()


def report(tag, ref):
    for i in range(2):
        d = (outs[i].float() - ref[i].float()).abs()
        print(f"{tag}: pipeline {i}: max |diff| {float(d.max()):.3e}, differing {100 * float((d > 0).float().mean()):.3f} %", flush=True)


run(False, 1)
ref = {i: outs[i].clone() for i in range(2)}
for k in range(2):
    run(False, 1)
    report(f"sequential {k}", ref)
for k in range(3):
    run(True, 2)
    report(f"concurrent {k}", ref)
