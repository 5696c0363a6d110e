#!/usr/bin/env python
"""FLUX.1-schnell (BASELINE config 4) at full size on one MI355X: transformer forward time (74.38 TFLOP per call at
4096 image + 512 text tokens) and a 4-step 1024x1024 image incl. the 16-channel VAE decode.  Seeded random weights."""
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
    t0 = time.perf_counter()
    pipe = factory.build_flux_pipeline(device=dev, tiny=False, seed=5)
    torch.cuda.synchronize()
    print(f"built FLUX.1-schnell in {time.perf_counter() - t0:.1f} s, {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
    g = torch.Generator("cpu").manual_seed(1234)
    pe = torch.randn((1, 512, 4096), generator=g).to(bf16).to(dev)
    pooled = torch.randn((1, 768), generator=g).to(bf16).to(dev)
    lat = torch.randn((1, 4096, 64), generator=g).to(bf16).to(dev)
    tr = pipe.transformer
    img_ids = pipe._prepare_latent_image_ids(64, 64)
    txt_ids = torch.zeros(512, 3)
    cond = tr.precompute_conditioning(pooled, img_ids, txt_ids)
    ts = torch.tensor([0.5])
    out = None
    times = []
    for i in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = tr(lat, encoder_hidden_states=pe, timestep=ts, conditioning=cond, return_dict=False)[0]
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    fwd = min(times[1:])
    rec = {"op": "flux_forward_eager", "ms": round(fwd * 1e3, 2), "tflops": round(74.3846 / fwd, 1),
           "first_call_s": round(times[0], 2), "finite": bool(torch.isfinite(out.float()).all()),
           "out_rms": float(out.float().pow(2).mean().sqrt())}
    print(json.dumps(rec), flush=True)
    res = []
    for i in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        img = pipe(prompt_embeds=pe, pooled_prompt_embeds=pooled, latents=lat, num_inference_steps=4, guidance_scale=0.0,
                   height=1024, width=1024, output_type="raw").images
        torch.cuda.synchronize()
        res.append(time.perf_counter() - t0)
    rec2 = {"op": "flux_schnell_image_1024_4step", "s_per_image": round(min(res[1:]), 4), "first_s": round(res[0], 2),
            "images_per_s": round(1.0 / min(res[1:]), 3), "tflops": round((4 * 74.3846 + 10.5) / min(res[1:]), 1),
            "finite": bool(torch.isfinite(img.float()).all()), "shape": list(img.shape)}
    print(json.dumps(rec2), flush=True)
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "flux.jsonl").write_text(json.dumps(rec) + "\n" + json.dumps(rec2) + "\n")


if __name__ == "__main__":
    main()
