"""HOST logic of every engine model class on torch-CPU stand-ins of the kernels (tests/ops_emulation.py): weight packing
(fused Q/K, V^T products, GEGLU interleave, head / channel zero padding, folded biases), call sequencing and strides are
checked against the reference's golden outputs without a GPU.  bf16 activations vs the fp32 reference: the same 2.5e-2
relative-RMS bound the GPU parity tests use.  The kernels themselves are only ever tested on the GPU."""
import pytest
import torch

import ops_emulation
from diffusers_amd import init as dinit, ops

bf16 = torch.bfloat16
TOL = 2.5e-2


def _t(g, k, dtype=bf16):
    return torch.from_numpy(g[k]).to(dtype)


def _rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


@pytest.fixture(autouse=True)
def _emulated_kernels(monkeypatch):
    ops_emulation.install(monkeypatch, ops)
    monkeypatch.setattr(ops, "TUNING", False)


@pytest.mark.parametrize("name,added", [("tiny_unet_sdxl", True), ("tiny_unet_sd15", False)])
def test_unet2d_condition(golden, name, added):
    from diffusers_amd.unet_2d_condition import UNet2DConditionModel
    cfg = dinit.TINY_SDXL_UNET if added else dinit.TINY_SD15_UNET
    g = golden(name)
    unet = UNet2DConditionModel(**cfg)
    unet.load_state_dict(dinit.random_state_dict(dinit.unet_param_shapes(unet.config), seed=0), device="cpu")
    kw = {}
    if added:
        kw["added_cond_kwargs"] = {"text_embeds": _t(g, "text_embeds"), "time_ids": _t(g, "time_ids", torch.float32)}
    y = unet(_t(g, "sample"), torch.tensor(float(g["t"])), _t(g, "ehs"), **kw).sample
    rr = _rel(y, torch.from_numpy(g["out"]))
    print(f"[host] {name}: rel rms vs reference fp32 = {rr:.3e}")
    assert y.shape == g["out"].shape and rr < TOL


def test_unet_sd15_head_padding(golden):
    """SD1.5's 8 heads of 40 / 80 / 160 channels: zero-padded to the flash kernel's 64 / 96 / 160."""
    from diffusers_amd.unet_2d_condition import UNet2DConditionModel
    g = golden("small_unet_sd15_heads")
    unet = UNet2DConditionModel(**dinit.SMALL_SD15_UNET)
    unet.load_state_dict(dinit.random_state_dict(dinit.unet_param_shapes(unet.config), seed=7), device="cpu")
    assert sorted({tr.blocks[0].attn1.kdim for tr in unet._transformers()}) == [64, 96, 160]
    y = unet(_t(g, "sample"), torch.tensor(float(g["t"])), _t(g, "ehs")).sample
    rr = _rel(y, torch.from_numpy(g["out"]))
    print(f"[host] SD1.5 head geometry: rel rms = {rr:.3e}")
    assert rr < TOL


@pytest.mark.parametrize("gemm_attn", [False, True])
def test_autoencoder_kl(golden, gemm_attn):
    from diffusers_amd.autoencoder_kl import AutoencoderKL
    g = golden("tiny_vae")
    vae = AutoencoderKL(**dinit.TINY_VAE)
    vae.load_state_dict(dinit.random_state_dict(dinit.vae_decoder_param_shapes(vae.config), seed=1), device="cpu")
    vae.mid_attn.force_gemm_path = gemm_attn
    img = vae.decode(_t(g, "z")).sample
    rr = _rel(img, torch.from_numpy(g["out"]))
    print(f"[host] tiny_vae (gemm_attn={gemm_attn}): rel rms = {rr:.3e}")
    assert rr < TOL


def test_flux_transformer(golden):
    from diffusers_amd.transformer_flux import FluxTransformer2DModel
    g = golden("tiny_flux")
    tr = FluxTransformer2DModel(**dinit.TINY_FLUX)
    tr.load_state_dict(dinit.random_state_dict(dinit.flux_param_shapes(tr.config), seed=5), device="cpu")
    y = tr(hidden_states=_t(g, "hidden_states"), encoder_hidden_states=_t(g, "encoder_hidden_states"),
           pooled_projections=_t(g, "pooled"), timestep=_t(g, "timestep", torch.float32),
           img_ids=torch.from_numpy(g["img_ids"]), txt_ids=torch.from_numpy(g["txt_ids"])).sample
    rr = _rel(y, torch.from_numpy(g["out"]))
    print(f"[host] tiny_flux: rel rms = {rr:.3e}")
    assert rr < TOL


def test_wan_transformer(golden):
    from diffusers_amd.transformer_wan import WanTransformer3DModel
    g = golden("tiny_wan")
    tr = WanTransformer3DModel(**dinit.TINY_WAN)
    tr.load_state_dict(dinit.random_state_dict(dinit.wan_param_shapes(tr.config), seed=9), device="cpu")
    y = tr(hidden_states=_t(g, "hidden_states"), timestep=torch.from_numpy(g["timestep"]),
           encoder_hidden_states=_t(g, "encoder_hidden_states")).sample
    rr = _rel(y, torch.from_numpy(g["out"]))
    print(f"[host] tiny_wan: rel rms = {rr:.3e}")
    assert rr < TOL


def test_unet2d_ddpm(golden):
    from diffusers_amd.unet_2d import UNet2DModel
    g = golden("tiny_ddpm")
    unet = UNet2DModel(**dinit.TINY_DDPM)
    unet.load_state_dict(dinit.random_state_dict(dinit.unet2d_param_shapes(dict(unet.config)), seed=11), device="cpu")
    y = unet(_t(g, "sample"), float(g["t"])).sample
    rr = _rel(y, torch.from_numpy(g["out"]))
    print(f"[host] tiny UNet2DModel: rel rms = {rr:.3e}")
    assert rr < TOL
