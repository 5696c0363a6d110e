"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Golden vectors for the Wan row from the REAL reference
(huggingface/diffusers imported from /root/reference/src), CPU fp32, seeded weights / inputs.  Build container only:

    PYTHONPATH=/root/reference/src python oracle/make_golden_wan.py

The denoising loop follows pipelines/wan/pipeline_wan.py:590-640 literally (two transformer calls per step, CFG combine,
scheduler.step) with FlowMatchEulerDiscreteScheduler(shift=3.0), the scheduler SURVEY.md 8d fixes for rows (a)-(e)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, "/root/reference/src")
from diffusers import FlowMatchEulerDiscreteScheduler, WanTransformer3DModel  # noqa: E402

from diffusers_amd import init as dinit  # noqa: E402

GOLD = ROOT / "tests" / "golden"
bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731


def main():
    torch.set_grad_enabled(False)
    cfg = dinit.TINY_WAN
    tr = WanTransformer3DModel(**cfg).eval()
    shapes = dinit.wan_param_shapes(cfg)
    assert {k: tuple(v.shape) for k, v in tr.state_dict().items()} == dict(shapes)
    sd = dinit.random_state_dict(shapes, seed=9)
    tr.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    g = torch.Generator().manual_seed(77)
    hs = bf(torch.randn((2, 16, 3, 8, 8), generator=g))
    ehs = bf(torch.randn((2, 16, 64), generator=g))
    t = torch.tensor([801, 201])
    y = tr(hidden_states=hs, timestep=t, encoder_hidden_states=ehs, return_dict=False)[0]
    np.savez_compressed(GOLD / "tiny_wan.npz", hidden_states=hs.numpy(), encoder_hidden_states=ehs.numpy(),
                        timestep=t.numpy(), out=y.numpy())
    print("tiny_wan out rms", float(y.pow(2).mean().sqrt()))

    sch = FlowMatchEulerDiscreteScheduler(shift=3.0)
    sch.set_timesteps(3)
    lat = bf(torch.randn((1, 16, 3, 8, 8), generator=g))
    pe, ne = ehs[:1].clone(), ehs[1:].clone()
    x = lat.clone()
    gs = 5.0
    sch.set_begin_index(0)
    for tt in sch.timesteps:
        ts = tt.expand(1)
        cond = tr(hidden_states=x, timestep=ts, encoder_hidden_states=pe, return_dict=False)[0]
        unc = tr(hidden_states=x, timestep=ts, encoder_hidden_states=ne, return_dict=False)[0]
        x = sch.step(unc + gs * (cond - unc), tt, x, return_dict=False)[0]
    np.savez_compressed(GOLD / "tiny_wan_pipeline.npz", prompt_embeds=pe.numpy(), negative_prompt_embeds=ne.numpy(),
                        latents=lat.numpy(), final_latents=x.numpy(), timesteps=sch.timesteps.numpy(),
                        sigmas=sch.sigmas.numpy(), guidance_scale=np.float32(gs))
    print("tiny_wan_pipeline latents rms", float(x.pow(2).mean().sqrt()), "timesteps", sch.timesteps.tolist())


if __name__ == "__main__":
    main()
