#!/usr/bin/env python
"""google/ddpm-cat-256 (BASELINE config 1): UNet2DModel 256x256 + DDPMScheduler, 50 ancestral steps, batch 1, on one
MI355X in bf16 (the reference runs this config on CPU in fp32: 2.24 s per forward on 8 vCPUs).
24.85 TFLOP per image.  Seeded random weights."""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import factory  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    pipe = factory.build_ddpm_pipeline(device=dev, tiny=False, seed=0)
    res = []
    img = None
    for i in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        img = pipe(batch_size=1, generator=torch.Generator().manual_seed(0), num_inference_steps=50, output_type="pt").images
        torch.cuda.synchronize()
        res.append(time.perf_counter() - t0)
    best = min(res[1:])
    rec = {"op": "ddpm_cat_256_50step", "s_per_image": round(best, 4), "images_per_s": round(1 / best, 3),
           "ms_per_step": round(best / 50 * 1e3, 2), "tflops": round(24.85 / best, 1), "first_s": round(res[0], 2),
           "finite": bool(torch.isfinite(img).all()), "shape": list(img.shape)}
    print(json.dumps(rec), flush=True)
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "ddpm.jsonl").write_text(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
