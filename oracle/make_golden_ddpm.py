"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Golden vectors for the DDPM row (BASELINE config 1) from the
REAL reference UNet2DModel / DDPMScheduler / DDPMPipeline, CPU fp32, seeded weights.  Build container only:

    PYTHONPATH=/root/reference/src python oracle/make_golden_ddpm.py"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, "/root/reference/src")
from diffusers import DDPMPipeline, DDPMScheduler, UNet2DModel  # noqa: E402

from diffusers_amd import init as dinit  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    cfg = dinit.TINY_DDPM
    ref = UNet2DModel(**cfg).eval()
    shapes = dinit.unet2d_param_shapes(dict(ref.config))
    assert {k: tuple(v.shape) for k, v in ref.state_dict().items()} == dict(shapes)
    sd = dinit.random_state_dict(shapes, seed=11)
    ref.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 3, 32, 32), generator=g).to(torch.bfloat16).float()
    y = ref(x, 601).sample
    pipe = DDPMPipeline(unet=ref, scheduler=DDPMScheduler(**dinit.DDPM_SCHEDULER))
    pipe.set_progress_bar_config(disable=True)
    img = pipe(batch_size=1, generator=torch.Generator().manual_seed(0), num_inference_steps=5, output_type="np").images
    np.savez_compressed(ROOT / "tests" / "golden" / "tiny_ddpm.npz", sample=x.numpy(), t=np.float32(601), out=y.numpy(),
                        pipeline_image=img.astype(np.float32))
    print("tiny_ddpm out rms", float(y.pow(2).mean().sqrt()), "pipeline image", img.shape, float(img.mean()))


if __name__ == "__main__":
    main()
