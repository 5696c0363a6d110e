#!/usr/bin/env python
"""Wan2.1-T2V-1.3B (BASELINE config 5) at full size on one MI355X: 832x480x81 frames -> latents (16, 21, 60, 104) =
32 760 tokens; one CFG step = one batch-2 transformer call = 2 x 283.0 TFLOP.  Seeded random weights."""
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
    pipe = factory.build_wan_pipeline(device=dev, tiny=False, seed=9)
    g = torch.Generator("cpu").manual_seed(1234)
    pe = torch.randn((1, 512, 4096), generator=g).to(bf16).to(dev)
    ne = torch.randn((1, 512, 4096), generator=g).to(bf16).to(dev)
    lat = torch.randn((1, 16, 21, 60, 104), generator=g).to(bf16).to(dev)
    steps = 3
    res = []
    out = None
    for i in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, num_inference_steps=steps,
                   guidance_scale=5.0, height=480, width=832, num_frames=81).images
        torch.cuda.synchronize()
        res.append(time.perf_counter() - t0)
    per_step = min(res[1:]) / steps
    rec = {"op": "wan13_cfg_step_832x480x81", "s_per_step": round(per_step, 4), "tflops": round(2 * 283.0018 / per_step, 1),
           "s_per_video_50_steps_extrapolated": round(50 * per_step, 1), "first_call_s": round(res[0], 2),
           "finite": bool(torch.isfinite(out.float()).all()), "shape": list(out.shape),
           "mem_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}
    print(json.dumps(rec), flush=True)
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "wan.jsonl").write_text(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
