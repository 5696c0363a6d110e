"""Checkpoint-side set-up: ``from_pretrained`` over the reference's on-disk layout, the pack-once cache,
LoRA adapters fused into the packed weights in place.  Packing is host arithmetic, so those tests run on CPU; the `gpu`-marked test
at the end runs the loaded / fused / unfused model's forward on hardware against the oracle on the same (merged) weights."""
import json
import sys
from pathlib import Path

import pytest
import torch

from diffusers_amd import init as dinit, loading, packed_cache
from diffusers_amd.autoencoder_kl import AutoencoderKL
from diffusers_amd.unet_2d_condition import UNet2DConditionModel

bf16 = torch.bfloat16
REF_SRC = Path("/root/reference/src")


def _tiny_unet_checkpoint(tmp_path, seed=0, variant=None, shards=1):
    cfg = dict(dinit.TINY_SDXL_UNET)
    sd = dinit.random_state_dict(dinit.unet_param_shapes(UNet2DConditionModel(**cfg).config), seed=seed)
    d = tmp_path / "unet"
    if shards == 1:
        loading.save_reference_checkpoint(sd, dict(cfg, _class_name="UNet2DConditionModel", _diffusers_version="0.40.0"), d,
                                          variant=variant)
    else:   # sharded layout with an index (modeling_utils.py: *.safetensors.index.json)
        from safetensors.torch import save_file
        d.mkdir(parents=True)
        (d / "config.json").write_text(json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}))
        keys = sorted(sd)
        wm = {}
        for i in range(shards):
            part = {k: sd[k].contiguous() for k in keys[i::shards]}
            name = f"diffusion_pytorch_model-{i + 1:05d}-of-{shards:05d}.safetensors"
            save_file(part, str(d / name))
            wm.update({k: name for k in part})
        (d / "diffusion_pytorch_model.safetensors.index.json").write_text(json.dumps({"metadata": {}, "weight_map": wm}))
    return d, cfg, sd


def _same_packed(a, b):
    ta, tb = packed_cache.packed_tensors(a), packed_cache.packed_tensors(b)
    assert list(ta) == list(tb)
    for k in ta:
        assert torch.equal(ta[k], tb[k]), k


@pytest.mark.parametrize("variant,shards", [(None, 1), ("fp16", 1), (None, 3)])
def test_from_pretrained_equals_load_state_dict_and_caches(tmp_path, monkeypatch, variant, shards):
    d, cfg, sd = _tiny_unet_checkpoint(tmp_path, variant=variant, shards=shards)
    want = UNet2DConditionModel(**cfg)
    want.load_state_dict(sd, device="cpu")
    got = UNet2DConditionModel.from_pretrained(d, variant=variant, device="cpu")
    _same_packed(got, want)
    cached = list((d / loading.PACKED_DIR).glob("UNet2DConditionModel-*.safetensors"))
    assert len(cached) == 1
    # second start: served from the packed file -- the checkpoint's tensors are never read again
    monkeypatch.setattr(loading.LazyCheckpoint, "__getitem__", lambda self, k: (_ for _ in ()).throw(AssertionError("re-read")))
    again = UNet2DConditionModel.from_pretrained(d, variant=variant, device="cpu")
    _same_packed(again, want)
    monkeypatch.undo()
    # a changed checkpoint invalidates the cache (new fingerprint -> new cache file, packed again)
    f = sorted(d.glob("diffusion_pytorch_model*.safetensors"))[0]
    from safetensors.torch import load_file, save_file
    t = load_file(str(f))
    k0 = sorted(t)[0]
    t[k0] = t[k0] + 1
    save_file(t, str(f))
    changed = UNet2DConditionModel.from_pretrained(d, variant=variant, device="cpu")
    assert len(list((d / loading.PACKED_DIR).glob("*.safetensors"))) == 2
    assert any(not torch.equal(a, b) for a, b in zip(packed_cache.packed_tensors(changed).values(),
                                                     packed_cache.packed_tensors(want).values()))


def test_from_pretrained_refuses_what_it_cannot_honour(tmp_path):
    d, cfg, sd = _tiny_unet_checkpoint(tmp_path)
    with pytest.raises(ValueError):
        UNet2DConditionModel.from_pretrained(d, torch_dtype=torch.float16, device="cpu")
    with pytest.raises(ValueError):
        UNet2DConditionModel.from_pretrained(d, use_safetensors=False, device="cpu")
    with pytest.raises(OSError):
        UNet2DConditionModel.from_pretrained(d, variant="nope", device="cpu")
    with pytest.raises(OSError):
        UNet2DConditionModel.from_pretrained(tmp_path / "missing", device="cpu")
    (d / "diffusion_pytorch_model.safetensors").rename(d / "diffusion_pytorch_model.bin")
    with pytest.raises(OSError, match="pickle"):
        UNet2DConditionModel.from_pretrained(d, device="cpu")


def _lora_for(sd, names, r=4, seed=3, fmt="peft", prefix="unet."):
    g = torch.Generator().manual_seed(seed)
    out, dense = {}, {}
    for n in names:
        w = sd[n + ".weight"]
        if w.dim() == 4:
            a = torch.randn((r, w.shape[1], w.shape[2], w.shape[3]), generator=g) * 0.05
            b = torch.randn((w.shape[0], r, 1, 1), generator=g) * 0.05
            delta = (b.reshape(w.shape[0], r) @ a.reshape(r, -1)).reshape(w.shape)
        else:
            a = torch.randn((r, w.shape[1]), generator=g) * 0.05
            b = torch.randn((w.shape[0], r), generator=g) * 0.05
            delta = b @ a
        alpha = 8.0
        dense[n] = delta * (alpha / r)
        if fmt == "peft":
            out[f"{prefix}{n}.lora_A.weight"], out[f"{prefix}{n}.lora_B.weight"] = a, b
        else:   # legacy diffusers attention-processor naming
            out[f"{prefix}{n}.lora.down.weight"], out[f"{prefix}{n}.lora.up.weight"] = a, b
        out[f"{prefix}{n}.alpha"] = torch.tensor(alpha)
    return out, dense


@pytest.mark.parametrize("fmt", ["peft", "legacy"])
def test_fuse_lora_repacks_in_place(tmp_path, fmt):
    d, cfg, sd = _tiny_unet_checkpoint(tmp_path)
    model = UNet2DConditionModel.from_pretrained(d, device="cpu")
    base_snapshot = {k: v.clone() for k, v in packed_cache.packed_tensors(model).items()}
    ptrs = {k: v.data_ptr() for k, v in packed_cache.packed_tensors(model).items()}
    targets = ["down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q",
               "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_v",
               "down_blocks.1.attentions.0.transformer_blocks.0.attn2.to_q",     # also feeds the folded LayerNorm vectors
               "down_blocks.1.attentions.0.transformer_blocks.0.ff.net.0.proj",
               "mid_block.attentions.0.proj_in", "down_blocks.0.resnets.0.conv1",
               "up_blocks.0.resnets.1.time_emb_proj"]       # one block's rows of the model's stacked time projections
    lora, dense = _lora_for(sd, targets, fmt=fmt)
    model.fuse_lora(lora, lora_scale=0.7)
    fused_sd = dict(sd)
    for n, dl in dense.items():
        fused_sd[n + ".weight"] = (sd[n + ".weight"].float() + 0.7 * dl).to(sd[n + ".weight"].dtype)
    want = UNet2DConditionModel(**cfg)
    want.load_state_dict(fused_sd, device="cpu")
    _same_packed(model, want)
    assert {k: v.data_ptr() for k, v in packed_cache.packed_tensors(model).items()} == ptrs, "re-pack moved a packed tensor"
    changed = [k for k, v in packed_cache.packed_tensors(model).items() if not torch.equal(v, base_snapshot[k])]
    assert changed and len(changed) < len(base_snapshot) // 4, "only the targeted layers' packed tensors may change"
    model.unfuse_lora()
    for k, v in packed_cache.packed_tensors(model).items():
        assert torch.equal(v, base_snapshot[k]), f"unfuse did not restore {k}"
    with pytest.raises(KeyError):
        model.fuse_lora({"unet.no_such_module.lora_A.weight": torch.zeros(2, 4), "unet.no_such_module.lora_B.weight": torch.zeros(4, 2)})
    with pytest.raises(ValueError):
        model.fuse_lora({"unet.mid_block.attentions.0.proj_in.lora_A.weight": torch.zeros(2, 4)})


def test_lora_files_with_other_components_and_legacy_processor_names(tmp_path):
    """(a) a pipeline-level LoRA file also carries `text_encoder.*` keys -- `load_lora_adapter` keeps only the
    keys under the model's prefix (loaders/peft.py:194-195), it does not raise; (b) the legacy attention-processor naming
    `<attn>.processor.to_out_lora.down.weight` targets `<attn>.to_out.0`; (c) no blanket `_lora` removal from module names."""
    from diffusers_amd.loading import _lora_pairs
    d, cfg, sd = _tiny_unet_checkpoint(tmp_path)
    model = UNet2DConditionModel.from_pretrained(d, device="cpu")
    attn = "down_blocks.1.attentions.0.transformer_blocks.0.attn1"
    lora, dense = _lora_for(sd, [attn + ".to_q", attn + ".to_out.0"])
    legacy = {}
    for k, v in lora.items():                       # PEFT keys -> LoRAAttnProcessor keys
        k = k.replace(".to_q.lora_A.weight", ".processor.to_q_lora.down.weight").replace(".to_q.lora_B.weight", ".processor.to_q_lora.up.weight")
        k = k.replace(".to_out.0.lora_A.weight", ".processor.to_out_lora.down.weight").replace(".to_out.0.lora_B.weight", ".processor.to_out_lora.up.weight")
        legacy[k] = v
    legacy = {k: v for k, v in legacy.items() if not k.endswith(".alpha")}
    legacy["text_encoder.text_model.encoder.layers.0.self_attn.q_proj.lora_A.weight"] = torch.zeros(4, 8)
    legacy["text_encoder.text_model.encoder.layers.0.self_attn.q_proj.lora_B.weight"] = torch.zeros(8, 4)
    pairs = _lora_pairs(legacy)
    assert sorted(pairs) == sorted([attn + ".to_q", attn + ".to_out.0"])
    model.fuse_lora(legacy, lora_scale=1.0)
    fused_sd = dict(sd)
    for n, dl in dense.items():                    # no alpha in the legacy file: scale = 1 (alpha / r was 2 in `dense`)
        fused_sd[n + ".weight"] = (sd[n + ".weight"].float() + 0.5 * dl).to(sd[n + ".weight"].dtype)
    want = UNet2DConditionModel(**cfg)
    want.load_state_dict(fused_sd, device="cpu")
    _same_packed(model, want)
    # a module whose own name contains "_lora" is not renamed
    assert list(_lora_pairs({"unet.my_lora_block.proj.lora_A.weight": torch.zeros(2, 4),
                             "unet.my_lora_block.proj.lora_B.weight": torch.zeros(4, 2)})) == ["my_lora_block.proj"]
    # a file for other components only: warn and leave the model alone, as loaders/peft.py:359-366 does (strict: raise)
    te_only = {"text_encoder.x.lora_A.weight": torch.zeros(2, 4), "text_encoder.x.lora_B.weight": torch.zeros(4, 2)}
    with pytest.warns(UserWarning, match="No LoRA keys"):
        assert _lora_pairs(te_only) == {}
    with pytest.raises(ValueError, match="No LoRA keys"):
        _lora_pairs(te_only, strict=True)
    before = {k: t.clone() for k, t in packed_cache.packed_tensors(model).items()}
    state = model._lora
    with pytest.warns(UserWarning, match="No LoRA keys"):
        model.load_lora_adapter(te_only)
    assert model._lora == state and all(torch.equal(t, before[k]) for k, t in packed_cache.packed_tensors(model).items())


def test_in_memory_models_need_a_base_for_lora():
    cfg = dict(dinit.TINY_SDXL_UNET)
    sd = dinit.random_state_dict(dinit.unet_param_shapes(UNet2DConditionModel(**cfg).config), seed=0)
    m = UNet2DConditionModel(**cfg)
    m.load_state_dict(sd, device="cpu")
    lora, dense = _lora_for(sd, ["mid_block.attentions.0.proj_out"])
    with pytest.raises(RuntimeError, match="base_state_dict"):
        m.fuse_lora(lora)
    m.fuse_lora(lora, base_state_dict=sd)
    assert m._lora == {"scale": 1.0, "modules": 1}


@pytest.mark.skipif(not REF_SRC.exists(), reason="needs the reference checkout (build container only)")
def test_loads_what_the_reference_save_pretrained_writes(tmp_path):
    """The REAL reference classes write the directory (ModelMixin.save_pretrained, modeling_utils.py:629-884); the engine
    reads it back -- U-Net (config round trip incl. private keys) and the VAE (decoder half of a full AutoencoderKL)."""
    sys.path.insert(0, str(REF_SRC))
    try:
        import diffusers as ref
        u = ref.UNet2DConditionModel(**dinit.TINY_SDXL_UNET)
        u.save_pretrained(tmp_path / "pipe" / "unet")
        got = UNet2DConditionModel.from_pretrained(tmp_path / "pipe", subfolder="unet", device="cpu")
        want = UNet2DConditionModel(**dinit.TINY_SDXL_UNET)
        want.load_state_dict({k: v.to(bf16) for k, v in u.state_dict().items()}, device="cpu")
        _same_packed(got, want)
        v = ref.AutoencoderKL(**dinit.TINY_VAE)
        v.save_pretrained(tmp_path / "pipe" / "vae")
        gv = AutoencoderKL.from_pretrained(tmp_path / "pipe", subfolder="vae", device="cpu")
        wv = AutoencoderKL(**dinit.TINY_VAE)
        wv.load_state_dict({k: t.to(bf16) for k, t in v.state_dict().items()}, device="cpu")
        _same_packed(gv, wv)
        assert gv.config.scaling_factor == v.config.scaling_factor
    finally:
        sys.path.remove(str(REF_SRC))


# ----------------------------------------------------------------------------------------------------------------------
# on hardware: the loaded / fused / unfused model's FORWARD against the oracle
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["peft", "legacy"])
def test_from_pretrained_and_lora_forward_on_gpu(tmp_path, fmt):
    """`from_pretrained` of a checkpoint directory -> forward vs the fp32 oracle on the checkpoint's weights; `fuse_lora(scale)` ->
    forward vs the oracle on the MERGED weights W + scale (alpha / r) B A (the reference's merge, loaders/lora_base.py:544 ->
    peft's `merge`); `unfuse_lora` -> bit-identical to the base forward; a second adapter replaces the first (adapters do not
    stack, documented in loading.py).  Key styles covered: PEFT (`lora_A.weight` / `lora_B.weight` + `.alpha`) and the legacy
    diffusers naming (`lora.down.weight` / `lora.up.weight`), both with the pipeline-level `unet.` prefix."""
    from diffusers_amd.unet_2d_condition import _DEFAULTS as UD
    from oracle import reference_math as R
    d, cfg, sd = _tiny_unet_checkpoint(tmp_path)
    model = UNet2DConditionModel.from_pretrained(d, device="cuda")
    full = dict(UD)
    full.update(cfg)
    g = torch.Generator().manual_seed(11)
    sample = torch.randn((2, 4, 16, 16), generator=g).to(bf16)
    ehs = torch.randn((2, 7, 64), generator=g).to(bf16)
    te = torch.randn((2, 64), generator=g).to(bf16)
    ids = torch.tensor([[128., 128., 0., 0., 128., 128.]]).repeat(2, 1)

    def engine():
        return model(sample.cuda(), torch.tensor(401.0), ehs.cuda(),
                     added_cond_kwargs={"text_embeds": te.cuda(), "time_ids": ids.cuda()}).sample.float().cpu()

    def oracle(weights):
        with torch.no_grad():
            return R.unet_forward({k: v.float() for k, v in weights.items()}, full, sample.float(), 401.0, ehs.float(),
                                  {"text_embeds": te.float(), "time_ids": ids})

    def rel(a, b):
        return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
    base = engine()
    want_base = oracle(sd)
    assert rel(base, want_base) < 2.5e-2
    targets = ["down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q",
               "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_out.0",
               "down_blocks.1.attentions.0.transformer_blocks.0.attn2.to_k",
               "down_blocks.1.attentions.0.transformer_blocks.0.ff.net.0.proj",
               "mid_block.attentions.0.proj_in", "up_blocks.0.resnets.0.conv1", "down_blocks.0.resnets.0.conv1"]
    lora, dense = _lora_for(sd, targets, fmt=fmt, r=4, seed=5)
    # make the adapter matter: scale the deltas up so the fused forward is measurably different from the base one
    lora = {k: (v * 6.0 if k.endswith(("lora_B.weight", "lora.up.weight")) else v) for k, v in lora.items()}
    dense = {k: v * 6.0 for k, v in dense.items()}
    model.fuse_lora(lora, lora_scale=0.7)
    merged = dict(sd)
    for n, dl in dense.items():
        merged[n + ".weight"] = (sd[n + ".weight"].float() + 0.7 * dl).to(sd[n + ".weight"].dtype)
    fused = engine()
    want_fused = oracle(merged)
    r_f, moved = rel(fused, want_fused), rel(want_fused, want_base)
    print(f"[parity] LoRA-fused tiny U-Net ({fmt} keys) on the GPU: rel-rms vs the oracle on merged weights {r_f:.3e} "
          f"(the adapter moves the output by {moved:.3e})")
    assert r_f < 2.5e-2 and moved > 4 * r_f, "the fused forward must follow the merged weights, not the base ones"
    model.unfuse_lora()
    assert torch.equal(engine(), base), "unfuse_lora must restore the base forward bit for bit"
    # a second adapter replaces the first
    lora2, dense2 = _lora_for(sd, targets[:2], fmt=fmt, r=2, seed=9)
    model.fuse_lora(lora, lora_scale=0.7)
    model.fuse_lora(lora2, lora_scale=1.0)
    merged2 = dict(sd)
    for n, dl in dense2.items():
        merged2[n + ".weight"] = (sd[n + ".weight"].float() + dl).to(sd[n + ".weight"].dtype)
    assert rel(engine(), oracle(merged2)) < 2.5e-2
