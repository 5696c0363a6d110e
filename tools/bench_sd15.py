#!/usr/bin/env python
"""stable-diffusion-v1-5 (BASELINE config 2): 512x512, 50-step DDIM, CFG 7.5, bf16, one MI355X; seeded random weights.
82.84 TFLOP per image (50 x 1.6065 + 2.5145)."""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import factory  # noqa: E402

bf16 = torch.bfloat16


def main():
    dev = torch.device("cuda", 0)
    pipe = factory.build_sd15_pipeline(device=dev, tiny=False, seed=0)
    g = torch.Generator("cpu").manual_seed(1234)
    pe = torch.randn((1, 77, 768), generator=g).to(bf16).to(dev)
    ne = torch.randn((1, 77, 768), generator=g).to(bf16).to(dev)
    lat = torch.randn((1, 4, 64, 64), generator=g).to(bf16).to(dev)
    res = []
    img = None
    for i in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        img = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat.clone(), num_inference_steps=50,
                   guidance_scale=7.5, output_type="raw").images
        torch.cuda.synchronize()
        res.append(time.perf_counter() - t0)
    best = min(res[1:])
    rec = {"op": "sd15_image_512_50step_ddim_cfg7.5", "s_per_image": round(best, 4), "images_per_s": round(1 / best, 3),
           "first_s": round(res[0], 2), "tflops": round(82.84 / best, 1), "finite": bool(torch.isfinite(img.float()).all()),
           "shape": list(img.shape)}
    print(json.dumps(rec), flush=True)
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "sd15.jsonl").write_text(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
