"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Golden trajectories of the reference UniPCMultistepScheduler in Wan 2.1's shipped
configuration (flow_prediction, use_flow_sigmas, flow_shift=3.0, order 2, bh2; pipeline_wan.py:52-59), CPU.  Build
container only:   PYTHONPATH=/root/reference/src python oracle/make_golden_unipc.py"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, "/root/reference/src")
from diffusers import UniPCMultistepScheduler  # noqa: E402


def main():
    out = {}
    g = torch.Generator().manual_seed(21)
    shape = (1, 4, 3, 8, 8)
    n = 7
    sch = UniPCMultistepScheduler(prediction_type="flow_prediction", use_flow_sigmas=True, flow_shift=3.0)
    sch.set_timesteps(n)
    out["timesteps"] = sch.timesteps.numpy()
    out["sigmas"] = sch.sigmas.numpy()
    x0 = torch.randn(shape, generator=g)
    vs = [torch.randn(shape, generator=g) for _ in range(n)]
    out["x0"] = x0.numpy()
    out["v"] = torch.stack(vs).numpy()
    for name, xd, vd in (("f32", torch.float32, torch.float32), ("mixed", torch.float32, torch.bfloat16),
                         ("bf16", torch.bfloat16, torch.bfloat16)):
        sch.set_timesteps(n)
        x = x0.to(xd)
        traj = []
        for i, t in enumerate(sch.timesteps):
            x = sch.step(vs[i].to(vd), t, x).prev_sample
            assert x.dtype == xd
            traj.append(x.float())
        out[f"traj_{name}"] = torch.stack(traj).numpy()
    np.savez_compressed(ROOT / "tests" / "golden" / "unipc.npz", **out)
    print("unipc ok", out["timesteps"], [float(np.sqrt((out[f'traj_{k}'][-1] ** 2).mean())) for k in ("f32", "mixed", "bf16")])


if __name__ == "__main__":
    main()
