"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Golden vector for SD1.5's attention head geometry (8 heads of 40 / 80 / 160
channels, unet_2d_condition.py:248-254) from the REAL reference UNet2DConditionModel, CPU fp32.  Build container only:

    PYTHONPATH=/root/reference/src python oracle/make_golden_sd15.py"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, "/root/reference/src")
from diffusers import UNet2DConditionModel  # noqa: E402

from diffusers_amd import init as dinit  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    cfg = dinit.SMALL_SD15_UNET
    ref = UNet2DConditionModel(**cfg).eval()
    shapes = dinit.unet_param_shapes(dict(ref.config))
    assert {k: tuple(v.shape) for k, v in ref.state_dict().items()} == dict(shapes)
    heads = [blk.attentions[0].transformer_blocks[0].attn1.heads for blk in ref.down_blocks]
    dims = [blk.attentions[0].transformer_blocks[0].attn1.inner_dim // h for blk, h in zip(ref.down_blocks, heads)]
    assert heads == [8, 8, 8] and dims == [40, 80, 160], (heads, dims)
    sd = dinit.random_state_dict(shapes, seed=7)
    ref.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    g = torch.Generator().manual_seed(99)
    bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    sample = bf(torch.randn((2, 4, 16, 16), generator=g))
    ehs = bf(torch.randn((2, 7, 64), generator=g))
    y = ref(sample, torch.tensor(601.0), ehs).sample
    np.savez_compressed(ROOT / "tests" / "golden" / "small_unet_sd15_heads.npz", sample=sample.numpy(), ehs=ehs.numpy(),
                        t=np.float32(601.0), out=y.numpy())
    print("small_unet_sd15_heads out rms", float(y.pow(2).mean().sqrt()), "params",
          sum(v.numel() for v in sd.values()) / 1e6, "M")


if __name__ == "__main__":
    main()
