#!/usr/bin/env python
"""One SDXL-shaped denoising step (scale_model_input, U-Net forward on the CFG pair, CFG combine + Euler step, step counter) and
one VAE decode, recorded as a launch plan (diffusers_amd/plan.py) and written for examples/abi_demo.cpp:

    python tools/make_plan_demo.py step.daplan && ./abi_demo step.daplan

The model is the small SDXL-shaped configuration of the test-suite (seeded random weights; the full model's regions would make a
5 GB file -- the mechanism is the same).  Needs an MI355X.  `build()` is what tests/test_plan_gpu.py checks bit for bit against
the Python-driven step."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
bf16 = torch.bfloat16


def build(device="cuda", steps=4, guidance=5.0):
    """-> dict(pipe, cond, latents, latents0, run (the recorded callable), reset (back to the state the step starts from))."""
    from diffusers_amd import factory
    pipe = factory.build_sdxl_pipeline(device=device, tiny=True, seed=0, init_device="cpu")
    g = torch.Generator("cpu").manual_seed(1234)
    pe = torch.randn((2, 77, 64), generator=g).to(bf16).to(device)
    te = torch.randn((2, 64), generator=g).to(bf16).to(device)
    latents0 = torch.randn((1, 4, 16, 16), generator=g).to(bf16).to(device)
    ids = torch.tensor([[64., 64., 0., 0., 64., 64.]], device=device).repeat(2, 1)
    cond = pipe.unet.precompute_conditioning(pe.contiguous(), {"text_embeds": te, "time_ids": ids})
    sch = pipe.scheduler
    sch.set_timesteps(steps, device=device)
    latents = latents0.clone()
    div = float(pipe.vae.config.scaling_factor)

    def reset():
        sch.reset(0)
        latents.copy_(latents0)

    def run():
        pipe._step(latents, cond, guidance, True)
        return pipe.vae.decode(latents, return_dict=False, latents_div=div)[0]

    return {"pipe": pipe, "cond": cond, "latents": latents, "latents0": latents0, "run": run, "reset": reset,
            "keep": [pe, te, ids, cond, latents, latents0, pipe]}


def main():
    from diffusers_amd import plan as P
    out = sys.argv[1] if len(sys.argv) > 1 else "step.daplan"
    st = build()
    st["reset"]()
    st["run"]()                       # un-recorded first pass: tunes shapes, builds lazy caches
    st["reset"]()
    pl, image = P.record(st["run"], keep=st["keep"])
    st["reset"]()
    info = pl.save(out, outputs=[st["latents"], image])
    print(f"{out}: {info}")
    names = {}
    for n in pl.names:
        names[n] = names.get(n, 0) + 1
    print("launches by entry point:", dict(sorted(names.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
