"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Golden vectors for AutoencoderKLWan.decode from the REAL
reference (huggingface/diffusers imported from /root/reference/src), CPU fp32, seeded weights / inputs.  Build container:

    PYTHONPATH=/root/reference/src python oracle/make_golden_wan_vae.py

The reference decodes frame by frame through its feature cache (autoencoder_kl_wan.py:1197-1205); the oracle and the
engine run the whole sequence at once, so this fixture is what pins that restatement.  5 latent frames -> 17 video
frames: covers the "Rep" first chunk, the zero-history second chunk and the steady state of both temporal upsamplers.
The latents are de-normalised as the pipeline does (pipeline_wan.py:653-661) before decode."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, "/root/reference/src")
from diffusers import AutoencoderKLWan  # noqa: E402

from diffusers_amd import init as dinit  # noqa: E402

GOLD = ROOT / "tests" / "golden"
bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731


def main():
    torch.set_grad_enabled(False)
    cfg = dinit.TINY_WAN_VAE
    vae = AutoencoderKLWan(**{k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}).eval()
    shapes = dinit.wan_vae_decoder_param_shapes(cfg)
    sd = dinit.random_state_dict(shapes, seed=21)
    missing, unexpected = vae.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.startswith(("encoder.", "quant_conv.")) for k in missing)
    g = torch.Generator().manual_seed(123)
    lat = bf(torch.randn((1, 16, 5, 6, 10), generator=g))
    mean = torch.tensor(vae.config.latents_mean).view(1, 16, 1, 1, 1)
    inv_std = 1.0 / torch.tensor(vae.config.latents_std).view(1, 16, 1, 1, 1)
    z = lat / inv_std + mean
    video = vae.decode(z, return_dict=False)[0]
    assert tuple(video.shape) == (1, 3, 17, 48, 80)
    np.savez_compressed(GOLD / "tiny_wan_vae.npz", latents=lat.numpy(), z=z.numpy(), video=video.numpy())
    print("tiny_wan_vae video rms", float(video.pow(2).mean().sqrt()), "clamped frac",
          float((video.abs() >= 1.0).float().mean()))


if __name__ == "__main__":
    main()
