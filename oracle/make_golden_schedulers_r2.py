"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Generates tests/golden/schedulers_r2.npz from the REAL reference schedulers
(imported from /root/reference/src; build container only):

    python oracle/make_golden_schedulers_r2.py

Cases added in round 2: v_prediction / sample prediction types (DDIM, Euler, DDPM), DDIM eta > 0 with given variance
noise, DDPM on 'linspace' / 'trailing' spacings with step counts that do not divide 1000 (previous_timestep follows the
schedule, scheduling_ddpm.py:648-668), FlowMatch-Euler with an fp32 sample and a bf16 model output (the reference's Wan
hand-over), the no-CFG Euler trajectory.  fp32 and bf16 trajectories of 6 steps on seeded random tensors.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, "/root/reference/src")

from diffusers import (DDIMScheduler, DDPMScheduler, EulerDiscreteScheduler,  # noqa: E402
                       FlowMatchEulerDiscreteScheduler)

GOLD = ROOT / "tests" / "golden"
N = 6
SHAPE = (2, 4, 8, 8)
SD = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")


def main():
    out = {}
    g = torch.Generator("cpu").manual_seed(77)
    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        x0 = torch.randn(SHAPE, generator=g).to(dt)
        eps = [torch.randn(SHAPE, generator=g).to(dt) for _ in range(N)]
        noise = [torch.randn(SHAPE, generator=g).to(dt) for _ in range(N)]
        out[f"x0_{dt_name}"] = x0.float().numpy()
        out[f"eps_{dt_name}"] = torch.stack(eps).float().numpy()
        out[f"noise_{dt_name}"] = torch.stack(noise).float().numpy()

        for pred in ("v_prediction", "sample"):
            d = DDIMScheduler(clip_sample=False, set_alpha_to_one=False, steps_offset=1, prediction_type=pred, **SD)
            d.set_timesteps(N)
            x, traj = x0.clone(), []
            for i, t in enumerate(d.timesteps):
                x = d.step(eps[i], t, x).prev_sample
                traj.append(x.float())
            out[f"ddim_{pred}_{dt_name}"] = torch.stack(traj).numpy()

            e = EulerDiscreteScheduler(steps_offset=1, timestep_spacing="leading", prediction_type=pred, **SD)
            e.set_timesteps(N)
            x = (x0 * e.init_noise_sigma).to(dt)
            out[f"euler_{pred}_start_{dt_name}"] = x.float().numpy()
            traj = []
            for i, t in enumerate(e.timesteps):
                e.scale_model_input(x, t)
                x = e.step(eps[i], t, x).prev_sample
                traj.append(x.float())
            out[f"euler_{pred}_{dt_name}"] = torch.stack(traj).numpy()

            p = DDPMScheduler(prediction_type=pred, clip_sample=True)
            p.set_timesteps(N)
            x, traj = x0.clone(), []
            for i, t in enumerate(p.timesteps):
                # the reference draws inside step(); feeding the draw through a generator keeps its code path: use a
                # fresh generator per step that reproduces noise[i]?  Not possible -- so take variance noise from the
                # generator stream below and RECORD it
                gg = torch.Generator("cpu").manual_seed(1000 + i)
                x = p.step(eps[i], t, x, generator=gg).prev_sample
                traj.append(x.float())
            out[f"ddpm_{pred}_{dt_name}"] = torch.stack(traj).numpy()

        # DDIM eta = 0.6 with explicit variance noise (epsilon and v_prediction)
        for pred in ("epsilon", "v_prediction"):
            d = DDIMScheduler(clip_sample=False, set_alpha_to_one=False, steps_offset=1, prediction_type=pred, **SD)
            d.set_timesteps(N)
            x, traj = x0.clone(), []
            for i, t in enumerate(d.timesteps):
                x = d.step(eps[i], t, x, eta=0.6, variance_noise=noise[i]).prev_sample
                traj.append(x.float())
            out[f"ddim_eta_{pred}_{dt_name}"] = torch.stack(traj).numpy()

        # DDPM spacings whose previous timestep is NOT t - 1000 // n
        for spacing, n in (("linspace", 7), ("trailing", 7), ("leading", 7)):
            p = DDPMScheduler(timestep_spacing=spacing, clip_sample=True)
            p.set_timesteps(n)
            out[f"ddpm_{spacing}{n}_timesteps"] = p.timesteps.numpy()
            x, traj = x0.clone(), []
            for i, t in enumerate(p.timesteps[:N]):
                gg = torch.Generator("cpu").manual_seed(1000 + i)
                x = p.step(eps[i], t, x, generator=gg).prev_sample
                traj.append(x.float())
            out[f"ddpm_{spacing}{n}_{dt_name}"] = torch.stack(traj).numpy()

    # the DDPM variance draws used above (seed 1000 + i, drawn in the model-output dtype like randn_tensor does)
    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        out[f"ddpm_draws_{dt_name}"] = torch.stack([
            torch.randn(SHAPE, generator=torch.Generator("cpu").manual_seed(1000 + i), dtype=dt).float()
            for i in range(N)]).numpy()

    # FlowMatch: fp32 sample, bf16 model output on every step (the result comes back in bf16 and is re-upcast by the
    # caller, as WanPipeline does with `latents` on FlowMatchEuler); plus the plain dtype-matched runs exist in r1
    f = FlowMatchEulerDiscreteScheduler(shift=3.0)
    f.set_timesteps(N)
    x = torch.from_numpy(out["x0_f32"]).clone()
    v = torch.from_numpy(out["eps_bf16"]).to(torch.bfloat16)
    traj = []
    for i, t in enumerate(f.timesteps):
        y = f.step(v[i], t, x).prev_sample
        assert y.dtype == torch.bfloat16
        traj.append(y.float())
        x = y.float()
    out["flow_mixed"] = torch.stack(traj).numpy()
    out["flow_mixed_sigmas"] = f.sigmas.numpy()

    np.savez_compressed(GOLD / "schedulers_r2.npz", **out)
    print("wrote", GOLD / "schedulers_r2.npz", len(out), "arrays")


if __name__ == "__main__":
    torch.set_grad_enabled(False)
    main()
