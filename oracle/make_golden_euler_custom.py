"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Sigma / timestep tables of the REAL reference EulerDiscreteScheduler under CUSTOM schedules
(`set_timesteps(timesteps=...)` / `set_timesteps(sigmas=...)`, scheduling_euler_discrete.py:378-407; the "align your steps" ladders
of the reference docs are the typical use), and the SDXL pipeline's `denoising_end` truncation
(pipeline_stable_diffusion_xl.py:1164-1183).  Build container only:

    PYTHONPATH=/root/reference/src python oracle/make_golden_euler_custom.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, "/root/reference/src")
from diffusers import EulerDiscreteScheduler  # noqa: E402

BASE = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1, timestep_spacing="leading")
TIMESTEPS = {"ays_sdxl": [999, 845, 730, 587, 443, 310, 193, 116, 53, 13], "three": [901, 501, 101]}
SIGMAS = {"ays_sdxl": [14.615, 6.315, 3.771, 2.181, 1.342, 0.862, 0.555, 0.380, 0.234, 0.113, 0.0], "short": [10.0, 2.5, 0.7, 0.0]}


def main():
    out = {}
    for name, ts in TIMESTEPS.items():
        for spacing in ("leading", "trailing"):
            s = EulerDiscreteScheduler(**dict(BASE, timestep_spacing=spacing))
            s.set_timesteps(timesteps=ts)
            out[f"timesteps_{name}_{spacing}_sigmas"] = s.sigmas.numpy()
            out[f"timesteps_{name}_{spacing}_timesteps"] = s.timesteps.numpy()
            out[f"timesteps_{name}_{spacing}_init_noise_sigma"] = np.float32(float(s.init_noise_sigma))
    for name, sg in SIGMAS.items():
        s = EulerDiscreteScheduler(**BASE)
        s.set_timesteps(sigmas=sg)
        out[f"sigmas_{name}_sigmas"] = s.sigmas.numpy()
        out[f"sigmas_{name}_timesteps"] = s.timesteps.numpy()
        out[f"sigmas_{name}_init_noise_sigma"] = np.float32(float(s.init_noise_sigma))
    # denoising_end: how many of the 50 / 30 leading-spaced steps survive the cut-off (the pipeline's arithmetic, :1170-1183)
    for n in (50, 30):
        s = EulerDiscreteScheduler(**BASE)
        s.set_timesteps(n)
        for frac in (0.8, 0.5, 0.25):
            cutoff = int(round(s.config.num_train_timesteps - frac * s.config.num_train_timesteps))
            out[f"denoising_end_{n}_{frac}"] = np.int64(len([t for t in s.timesteps if t >= cutoff]))
    np.savez_compressed(ROOT / "tests" / "golden" / "euler_custom.npz", **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
