"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Sigma / timestep tables of the REAL reference EulerDiscreteScheduler with the
Karras and exponential sigma ladders (scheduling_euler_discrete.py:446-452, :483-585), SDXL's beta schedule.  Build
container only:

    PYTHONPATH=/root/reference/src python oracle/make_golden_euler_karras.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, "/root/reference/src")
from diffusers import EulerDiscreteScheduler  # noqa: E402

BASE = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1, timestep_spacing="leading")
CASES = {
    "karras": dict(use_karras_sigmas=True),
    "exponential": dict(use_exponential_sigmas=True),
    "karras_minmax_trailing": dict(use_karras_sigmas=True, sigma_min=0.05, sigma_max=10.0, timestep_spacing="trailing"),
    "karras_sigma_min_last": dict(use_karras_sigmas=True, final_sigmas_type="sigma_min"),
}


def main():
    out = {}
    for name, extra in CASES.items():
        for n in (4, 25, 50):
            s = EulerDiscreteScheduler(**dict(BASE, **extra))
            s.set_timesteps(n)
            out[f"{name}_{n}_sigmas"] = s.sigmas.numpy()
            out[f"{name}_{n}_timesteps"] = s.timesteps.numpy()
            out[f"{name}_{n}_init_noise_sigma"] = np.float32(float(s.init_noise_sigma))
    np.savez_compressed(ROOT / "tests" / "golden" / "euler_karras.npz", **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
