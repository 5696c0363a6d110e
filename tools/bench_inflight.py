#!/usr/bin/env python
"""Round 6 experiment: how much of an SDXL image is launch-latency bubbles?  TWO independent batch-1 pipelines (same U-Net / VAE
weights, own schedulers, own step graphs, own streams and -- through pipelines.STREAM_DOMAIN -- own workspaces) generate images
concurrently from two host threads, against the same two images generated one after the other.  Every image is the same bits either
way (checked).  This is NOT the headline's protocol (one prompt per GPU, one image at a time); it measures what a serving host that
keeps two requests in flight per GPU would get.  usage: bench_inflight.py out.json [images per pipeline]"""
import json
import sys
import threading
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from diffusers_amd import factory, init as dinit, pipelines as P  # noqa: E402
from diffusers_amd.schedulers import EulerDiscreteScheduler  # noqa: E402


def main():
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda", 0)
    unet, _ = factory.build_unet(dinit.SDXL_UNET, seed=0, device=dev, init_device=str(dev))
    vae, _ = factory.build_vae(dinit.SDXL_VAE, seed=1, device=dev, init_device=str(dev))
    pipes = [P.StableDiffusionXLPipeline(vae=vae, unet=unet, scheduler=EulerDiscreteScheduler(**factory.SDXL_SCHEDULER)) for _ in range(2)]
    inp = bench.synth_inputs(2, False, dev)
    streams = [torch.cuda.Stream() for _ in range(2)]

    def kw(i):
        return dict(prompt_embeds=inp["prompt_embeds"][i:i + 1], negative_prompt_embeds=inp["negative_prompt_embeds"][i:i + 1],
                    pooled_prompt_embeds=inp["pooled"][i:i + 1], negative_pooled_prompt_embeds=inp["negative_pooled"][i:i + 1],
                    num_inference_steps=50, guidance_scale=bench.GUIDANCE, height=1024, width=1024, output_type="pt")
    outs = [None, None]

    def work(i, count):
        P.STREAM_DOMAIN.tag = i + 1
        with torch.cuda.stream(streams[i]):
            for _ in range(count):
                outs[i] = pipes[i](latents=inp["latents"][i:i + 1].clone(), **kw(i)).images
            streams[i].synchronize()

    def run_threads(count):
        th = [threading.Thread(target=work, args=(i, count)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    # warm-up / capture, one pipeline at a time (each in its own stream domain)
    for i in range(2):
        t = threading.Thread(target=work, args=(i, 1))
        t.start()
        t.join()
    torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    res = {}
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2):                      # one after the other (each still from its own thread / stream)
            t = threading.Thread(target=work, args=(i, n))
            t.start()
            t.join()
        torch.cuda.synchronize()
        seq = time.perf_counter() - t0
        same_seq = all(torch.equal(o, r) for o, r in zip(outs, ref))
        t0 = time.perf_counter()
        run_threads(n)
        torch.cuda.synchronize()
        con = time.perf_counter() - t0
        same_con = all(torch.equal(o, r) for o, r in zip(outs, ref))
        res[f"pass{rep}"] = {"sequential_images_per_s": 2 * n / seq, "two_in_flight_images_per_s": 2 * n / con, "ratio": seq / con,
                             "bit_identical_sequential": same_seq, "bit_identical_concurrent": same_con}
        print(json.dumps(res[f"pass{rep}"]), flush=True)
    rec = {"what": "SDXL-base 1024x1024, 50 Euler steps, CFG 5, bf16: two batch-1 pipelines on one MI355X, sequential vs concurrent "
                   "(two host threads, two streams, separate step graphs and workspaces, shared weights)", "images_per_pipeline": n, **res}
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(json.dumps(rec, indent=1) + "\n")


if __name__ == "__main__":
    main()
