"""Boundaries B3 / B4 of SURVEY.md 8b through the reference's PUBLIC API, on REAL reference modules.

B4: ``register_backend()`` then ``with attention_backend("mi355x"):`` (models/attention_dispatch.py:370-389) and
    ``model.set_attention_backend("mi355x")`` (models/modeling_utils.py:598-660) on a real ``FluxTransformer2DModel`` /
    ``WanTransformer3DModel``: every ``dispatch_attention_fn`` call (transformer_flux.py:121-130) lands in
    ``da_attention_bf16`` (counted), and the bf16 forward agrees with the model's own fp32 run within 1.2 x the bf16 floor
    (the same model in bf16 on the reference's native backend).
B3: ``unet.set_attn_processor(MI355XAttnProcessor())`` (models/attention.py:64-96 / unets/unet_2d_condition.py) on a real
    reference ``UNet2DConditionModel`` -- contract of tests/models/unets/test_models_unet_2d_condition.py:760-825.

The same bodies run twice: on CPU here (reference from /root/reference/src, kernels = the torch stand-ins of
tests/ops_emulation.py -- checks the binding, the layouts and the strides) and on the GPU box (reference from
oracle/_ref/diffusers_ref.zip, the HIP kernels)."""
import importlib
import sys
from pathlib import Path

import pytest
import torch

from oracle import ref_runtime as RR

REF = Path("/root/reference/src")
bf16 = torch.bfloat16


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


def _load_ref():
    if REF.exists():
        sys.path.insert(0, str(REF))
        try:
            import diffusers
        finally:
            sys.path.remove(str(REF))
        return diffusers
    ref = RR.load_reference()
    if ref is None:
        pytest.skip("neither /root/reference/src nor oracle/_ref/diffusers_ref.zip is present")
    return ref


class _Counter:
    def __init__(self, monkeypatch):
        from diffusers_amd import ops
        self.n = 0
        self.shapes = []
        inner = ops.attention

        def counted(q, k, vt, **kw):
            self.n += 1
            self.shapes.append((kw["B"], kw["H"], kw["Sq"], kw["Skv"], kw["D"]))
            return inner(q, k, vt, **kw)
        monkeypatch.setattr(ops, "attention", counted)


def _env(monkeypatch, dev):
    """CPU: install the kernel stand-ins; GPU: the real library.  Returns the attention-launch counter."""
    from diffusers_amd import ops
    if dev == "cpu":
        import ops_emulation
        ops_emulation.install(monkeypatch, ops)
        monkeypatch.setattr(ops, "TUNING", False)
    return _Counter(monkeypatch)


def _as_lists(c):
    return {k: (list(v) if isinstance(v, tuple) else v) for k, v in c.items()}


def _flux(ref, cfg, dev, seed=0):
    torch.manual_seed(seed)
    m = ref.FluxTransformer2DModel(**_as_lists(cfg)).eval()
    return m.to(dev)


def _flux_inputs(cfg, dev, s_img, s_txt, seed=1):
    g = torch.Generator().manual_seed(seed)
    B = 1
    hs = torch.randn((B, s_img, cfg["in_channels"]), generator=g)
    ehs = torch.randn((B, s_txt, cfg["joint_attention_dim"]), generator=g)
    pooled = torch.randn((B, cfg["pooled_projection_dim"]), generator=g)
    side = int(s_img ** 0.5)
    img_ids = torch.zeros((s_img, 3))
    img_ids[:, 1] = torch.arange(side).repeat_interleave(side)[:s_img]
    img_ids[:, 2] = torch.arange(side).repeat(side)[:s_img]
    return dict(hidden_states=hs.to(dev), encoder_hidden_states=ehs.to(dev), pooled_projections=pooled.to(dev),
                timestep=torch.tensor([0.7]).to(dev), img_ids=img_ids.to(dev), txt_ids=torch.zeros((s_txt, 3)).to(dev))


def _cast(kw, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() and k not in ("img_ids", "txt_ids") else v) for k, v in kw.items()}


def _run_flux_backend(monkeypatch, dev, cfg, s_img, s_txt, gate):
    from diffusers_amd.attention_backend import BACKEND_NAME, register_backend
    ref = _load_ref()
    cnt = _env(monkeypatch, dev)
    from_ref = importlib.import_module(ref.__name__ + ".models.attention_dispatch")
    name = register_backend()
    assert name.value == BACKEND_NAME and from_ref.AttentionBackendName(BACKEND_NAME) is name
    assert BACKEND_NAME in {m.value for m in from_ref.AttentionBackendName.__members__.values()}
    model = _flux(ref, cfg, dev)
    kw = _flux_inputs(cfg, dev, s_img, s_txt)
    with torch.no_grad():
        want = model(**kw, return_dict=False)[0]                                  # the model's own fp32 run
        model.to(bf16)
        kwb = _cast(kw, bf16)
        floor = model(**kwb, return_dict=False)[0]                                # bf16, the reference's native backend
        assert cnt.n == 0
        with from_ref.attention_backend(BACKEND_NAME):                            # the public context manager
            got_ctx = model(**kwb, return_dict=False)[0]
        layers = cfg["num_layers"] + cfg["num_single_layers"]
        assert cnt.n == layers, f"{cnt.n} flash launches for {layers} attention layers"
        assert all(s == (1, cfg["num_attention_heads"], s_img + s_txt, s_img + s_txt, cfg["attention_head_dim"]) for s in cnt.shapes)
        model.set_attention_backend(BACKEND_NAME)                                 # the public model method
        try:
            got_set = model(**kwb, return_dict=False)[0]
        finally:
            model.reset_attention_backend()
            from_ref._AttentionBackendRegistry.set_active_backend(from_ref.AttentionBackendName.NATIVE)
        assert cnt.n == 2 * layers
        after = model(**kwb, return_dict=False)[0]                                # reset: native again, no launches
        assert cnt.n == 2 * layers and torch.equal(after, floor)
    rf, rc, rs = _rel(floor, want), _rel(got_ctx, want), _rel(got_set, want)
    print(f"[B4] reference FluxTransformer2DModel (S {s_img}+{s_txt}, H {cfg['num_attention_heads']}, D {cfg['attention_head_dim']}) under "
          f"attention_backend('{BACKEND_NAME}'): rel-rms vs its fp32 run {rc:.3e} (set_attention_backend: {rs:.3e}; native bf16 floor {rf:.3e})")
    assert torch.equal(got_ctx, got_set)
    assert rc <= gate(rf), (rc, rf)


TINY_FLUX = dict(patch_size=1, in_channels=64, out_channels=None, num_layers=2, num_single_layers=2, attention_head_dim=64,
                 num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=64, guidance_embeds=False,
                 axes_dims_rope=(8, 28, 28))


@pytest.mark.skipif(not REF.exists(), reason="reference sources not present")
def test_backend_binds_through_the_public_api_cpu(monkeypatch):
    _run_flux_backend(monkeypatch, "cpu", TINY_FLUX, s_img=64, s_txt=16, gate=lambda f: max(1.5 * f, 2.5e-2))


@pytest.mark.gpu
def test_backend_binds_through_the_public_api_tiny(monkeypatch):
    _run_flux_backend(monkeypatch, "cuda", TINY_FLUX, s_img=64, s_txt=16, gate=lambda f: max(1.2 * f, 1.5e-2))


@pytest.mark.gpu
def test_backend_under_a_full_width_flux_block(monkeypatch):
    """One double + one single block at FLUX.1 width (24 heads of 128, joint dim 4096) and the full 4096 + 512 sequence."""
    cfg = dict(patch_size=1, in_channels=64, out_channels=None, num_layers=1, num_single_layers=1, attention_head_dim=128,
               num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=False,
               axes_dims_rope=(16, 56, 56))
    _run_flux_backend(monkeypatch, "cuda", cfg, s_img=4096, s_txt=512, gate=lambda f: 1.2 * f)


def _run_wan_backend(monkeypatch, dev, gate):
    """The same binding under a real WanTransformer3DModel (transformer_wan.py:143-155: self-attention over the video tokens and
    cross-attention over the text tokens both go through dispatch_attention_fn)."""
    from diffusers_amd import init as dinit
    from diffusers_amd.attention_backend import BACKEND_NAME, register_backend
    ref = _load_ref()
    cnt = _env(monkeypatch, dev)
    ad = importlib.import_module(ref.__name__ + ".models.attention_dispatch")
    register_backend()
    cfg = dinit.TINY_WAN
    torch.manual_seed(0)
    model = ref.WanTransformer3DModel(**_as_lists(cfg)).eval().to(dev)
    g = torch.Generator().manual_seed(2)
    hs = torch.randn((1, cfg["in_channels"], 3, 16, 16), generator=g).to(dev)            # 3 x 8 x 8 = 192 tokens
    ehs = torch.randn((1, 16, cfg["text_dim"]), generator=g).to(dev)
    ts = torch.tensor([500]).to(dev)
    with torch.no_grad():
        want = model(hidden_states=hs, timestep=ts, encoder_hidden_states=ehs, return_dict=False)[0]
        model.to(bf16)
        floor = model(hidden_states=hs.to(bf16), timestep=ts, encoder_hidden_states=ehs.to(bf16), return_dict=False)[0]
        assert cnt.n == 0
        with ad.attention_backend(BACKEND_NAME):
            got = model(hidden_states=hs.to(bf16), timestep=ts, encoder_hidden_states=ehs.to(bf16), return_dict=False)[0]
    assert cnt.n == 2 * cfg["num_layers"], f"{cnt.n} flash launches for {cfg['num_layers']} blocks (self + cross each)"
    assert {s[3] for s in cnt.shapes} == {192, 16}                                         # key counts: video tokens, text tokens
    rf, rg = _rel(floor, want), _rel(got, want)
    print(f"[B4] reference WanTransformer3DModel under attention_backend('{BACKEND_NAME}'): rel-rms vs its fp32 run {rg:.3e} (native bf16 floor {rf:.3e})")
    assert rg <= gate(rf), (rg, rf)


@pytest.mark.skipif(not REF.exists(), reason="reference sources not present")
def test_backend_under_the_reference_wan_model_cpu(monkeypatch):
    _run_wan_backend(monkeypatch, "cpu", gate=lambda f: max(1.5 * f, 2.5e-2))


@pytest.mark.gpu
def test_backend_under_the_reference_wan_model(monkeypatch):
    _run_wan_backend(monkeypatch, "cuda", gate=lambda f: max(1.2 * f, 1.5e-2))


def _run_backend_gates(monkeypatch, dev):
    """What the reference's public API does with the slots: gated slots are refused by register_backend (the API would raise
    before dispatching); an ungated slot can be taken over; unsupported arguments raise from the backend itself."""
    from diffusers_amd.attention_backend import UNGATED_SLOTS, mi355x_flash_attention, register_backend
    ref = _load_ref()
    _env(monkeypatch, dev)
    ad = importlib.import_module(ref.__name__ + ".models.attention_dispatch")
    with pytest.raises(ValueError, match="gated"):
        register_backend(slot="aiter_fa2_hub")
    saved = {k: dict(getattr(ad._AttentionBackendRegistry, k)) for k in ("_backends", "_constraints", "_supported_arg_names")}
    try:
        name = register_backend(slot=UNGATED_SLOTS[0])
        assert ad._AttentionBackendRegistry._backends[name] is mi355x_flash_attention
        q = torch.randn((1, 64, 2, 64), generator=torch.Generator().manual_seed(0)).to(bf16).to(dev)
        with ad.attention_backend(UNGATED_SLOTS[0]):
            o = ad.dispatch_attention_fn(q, q, q)
            with pytest.raises(ValueError, match="not supported"):
                ad.dispatch_attention_fn(q, q, q, is_causal=True)
        want = torch.nn.functional.scaled_dot_product_attention(*(q.float().cpu().transpose(1, 2),) * 3).transpose(1, 2)
        assert _rel(o, want) < 1e-2
    finally:
        for k, v in saved.items():
            getattr(ad._AttentionBackendRegistry, k).clear()
            getattr(ad._AttentionBackendRegistry, k).update(v)


@pytest.mark.skipif(not REF.exists(), reason="reference sources not present")
def test_backend_slots_and_refusals_cpu(monkeypatch):
    _run_backend_gates(monkeypatch, "cpu")


@pytest.mark.gpu
def test_backend_slots_and_refusals(monkeypatch):
    _run_backend_gates(monkeypatch, "cuda")


def _run_backend_layouts(monkeypatch, dev):
    """Strided inputs reach the kernel without copies: q / k / v as column blocks of one fused [B][S][3*H*D] projection, V
    handed over as a view of a V^T buffer, a ragged key count, batch 2."""
    from diffusers_amd import attention_backend as AB
    cnt = _env(monkeypatch, dev)
    g = torch.Generator().manual_seed(3)
    B, S, H, D = 2, 72, 2, 64
    qkv = torch.randn((B, S, 3 * H * D), generator=g).to(bf16).to(dev)
    q, k, v = (qkv[..., i * H * D:(i + 1) * H * D].unflatten(-1, (H, D)) for i in range(3))

    def sdpa(q_, k_, v_):
        f = lambda t: t.float().cpu().transpose(1, 2)     # noqa: E731
        return torch.nn.functional.scaled_dot_product_attention(f(q_), f(k_), f(v_)).transpose(1, 2)
    copies = []
    orig = torch.Tensor.contiguous
    monkeypatch.setattr(torch.Tensor, "contiguous", lambda t, *a, **k_: (copies.append(t.is_contiguous()), orig(t, *a, **k_))[1])
    o = AB.mi355x_flash_attention(q, k, v)
    assert copies.count(False) == 0, "a fused-projection slice was copied"
    monkeypatch.setattr(torch.Tensor, "contiguous", orig)
    assert _rel(o, sdpa(q, k, v)) < 1e-2
    vt = torch.zeros((H * D, B * S), dtype=bf16, device=dev)          # the caller already holds V^T
    vt.copy_(v.reshape(B * S, H * D).t())
    v_view = vt.view(H, D, B, S).permute(2, 3, 0, 1)
    assert AB._is_vt_layout(v_view) and torch.equal(v_view, v)
    o2 = AB.mi355x_flash_attention(q, k, v_view)
    assert torch.equal(o2, o)
    kr, vr = (torch.randn((B, 77, H, D), generator=g).to(bf16).to(dev) for _ in range(2))    # ragged key count
    assert _rel(AB.mi355x_flash_attention(q, kr, vr), sdpa(q, kr, vr)) < 1e-2
    assert cnt.n == 3


@pytest.mark.skipif(not REF.exists(), reason="reference sources not present")
def test_backend_layouts_cpu(monkeypatch):
    _run_backend_layouts(monkeypatch, "cpu")


@pytest.mark.gpu
def test_backend_layouts(monkeypatch):
    _run_backend_layouts(monkeypatch, "cuda")


# ---- B3 ------------------------------------------------------------------------------------------------------------------------
def _run_processor(monkeypatch, dev, cfg, hw, gate, cross_tokens=77):
    from diffusers_amd.attention_backend import MI355XAttnProcessor
    ref = _load_ref()
    cnt = _env(monkeypatch, dev)
    torch.manual_seed(0)
    unet = ref.UNet2DConditionModel(**_as_lists(cfg)).eval().to(dev)
    g = torch.Generator().manual_seed(5)
    cd = cfg["cross_attention_dim"]
    sample = torch.randn((2, 4, hw, hw), generator=g).to(dev)
    ehs = torch.randn((2, cross_tokens, cd), generator=g).to(dev)
    added = None
    if cfg.get("addition_embed_type") == "text_time":
        nt = cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"]
        added = {"text_embeds": torch.randn((2, nt), generator=g).to(dev),
                 "time_ids": torch.tensor([[64., 64., 0., 0., 64., 64.]]).repeat(2, 1).to(dev)}

    def fwd(dtype):
        ak = None if added is None else {"text_embeds": added["text_embeds"].to(dtype), "time_ids": added["time_ids"].to(dtype)}
        return unet(sample.to(dtype), torch.tensor(481.0).to(dev), encoder_hidden_states=ehs.to(dtype), added_cond_kwargs=ak,
                    return_dict=False)[0]
    with torch.no_grad():
        want = fwd(torch.float32)
        unet.to(bf16)
        floor = fwd(bf16)
        assert cnt.n == 0
        n_attn = len(unet.attn_processors)
        proc = MI355XAttnProcessor()
        unet.set_attn_processor(proc)                                            # the public call (one shared processor)
        assert all(p is proc for p in unet.attn_processors.values())
        got = fwd(bf16)
        assert cnt.n == n_attn, f"{cnt.n} flash launches for {n_attn} attention layers"
        # cross-attention K / V^T are kept per module while the caller passes the same text-embedding tensor
        from diffusers_amd import ops
        lin = []
        orig = ops.linear
        monkeypatch.setattr(ops, "linear", lambda *a, **k_: (lin.append(a[0].shape), orig(*a, **k_))[1])
        ehs_b = ehs.to(bf16)
        a1 = unet(sample.to(bf16), torch.tensor(481.0).to(dev), encoder_hidden_states=ehs_b, return_dict=False,
                  added_cond_kwargs=None if added is None else {k: v.to(bf16) for k, v in added.items()})[0]
        n1 = len(lin)
        lin.clear()
        a2 = unet(sample.to(bf16), torch.tensor(481.0).to(dev), encoder_hidden_states=ehs_b, return_dict=False,
                  added_cond_kwargs=None if added is None else {k: v.to(bf16) for k, v in added.items()})[0]
        assert len(lin) == n1 - 2 * (n_attn // 2)                                # second call: no K / V^T launches
        # (whole-model equality is asserted on CPU only: on the GPU the reference's OWN convolutions are not reproducible from call to
        # call -- MIOpen's find mode settles on another solver as its database warms, seen as ~3e-3 rel-rms between two identical
        # forwards -- so there the module-level check below carries the exactness claim and the model-level one is the fp32 gate)
        assert torch.equal(a1, a2) if dev == "cpu" else _rel(a1, want) <= gate(rf := _rel(floor, want))
        repeat_ours = _rel(a1, a2)
        ehs_b.mul_(0.5)                                                          # in-place edit: version bump -> recomputed
        lin.clear()
        a3 = unet(sample.to(bf16), torch.tensor(481.0).to(dev), encoder_hidden_states=ehs_b, return_dict=False,
                  added_cond_kwargs=None if added is None else {k: v.to(bf16) for k, v in added.items()})[0]
        assert len(lin) == n1 and not torch.equal(a3, a2)
        monkeypatch.setattr(ops, "linear", orig)
        # one cross-attention module on its own: cache miss, cache hit and a fresh processor agree bit for bit
        blk = next(m for n, m in unet.named_modules() if n.endswith("transformer_blocks.0"))
        xs = torch.randn((2, 64, blk.attn2.to_q.weight.shape[1]), generator=torch.Generator().manual_seed(9)).to(bf16).to(dev)
        e2 = ehs.to(bf16)
        y0 = blk.attn2(xs, encoder_hidden_states=e2)
        y0 = blk.attn2(xs, encoder_hidden_states=e2)                          # (variant choices settled)
        blk.attn2.set_processor(MI355XAttnProcessor())
        y1 = blk.attn2(xs, encoder_hidden_states=e2)                          # miss
        y2 = blk.attn2(xs, encoder_hidden_states=e2)                          # hit
        assert torch.equal(y0, y1) and torch.equal(y1, y2)
        s0 = blk.attn1(xs)
        assert torch.equal(s0, blk.attn1(xs))
        unet.set_attn_processor(importlib.import_module(ref.__name__ + ".models.attention_processor").AttnProcessor2_0())
        back = fwd(bf16)
        back2 = fwd(bf16)
        print(f"[B3] two identical bf16 forwards: rel-rms {repeat_ours:.3e} with MI355XAttnProcessor, {_rel(back, back2):.3e} with the reference's "
              f"AttnProcessor2_0 (its convolutions / norms are PyTorch-ROCm's)")
        assert torch.equal(back, floor) if dev == "cpu" else _rel(back, want) <= 1.2 * _rel(floor, want)
    rf, rg = _rel(floor, want), _rel(got, want)
    print(f"[B3] reference UNet2DConditionModel ({n_attn} attention layers) after set_attn_processor(MI355XAttnProcessor()): rel-rms vs its "
          f"fp32 run {rg:.3e} (AttnProcessor2_0 in bf16: {rf:.3e})")
    assert rg <= gate(rf), (rg, rf)


TINY_UNET = dict(sample_size=16, in_channels=4, out_channels=4, block_out_channels=(64, 128), layers_per_block=1,
                 cross_attention_dim=64, attention_head_dim=(1, 2), down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"),
                 up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 2),
                 use_linear_projection=True, addition_embed_type="text_time", addition_time_embed_dim=32,
                 projection_class_embeddings_input_dim=256)


@pytest.mark.skipif(not REF.exists(), reason="reference sources not present")
def test_processor_on_the_reference_unet_cpu(monkeypatch):
    _run_processor(monkeypatch, "cpu", TINY_UNET, hw=16, gate=lambda f: max(1.5 * f, 2.5e-2))


@pytest.mark.gpu
def test_processor_on_the_reference_unet_tiny(monkeypatch):
    _run_processor(monkeypatch, "cuda", TINY_UNET, hw=16, gate=lambda f: max(1.2 * f, 1.5e-2))


@pytest.mark.gpu
def test_processor_on_the_reference_unet_sdxl_width(monkeypatch):
    """SDXL's widths and head geometry (640 / 1280 channels, heads of 64, 2048-wide text states, 77 tokens) with one
    transformer layer per block, at 64 x 64 latents: S = 1024 and 256."""
    cfg = dict(sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=1,
               cross_attention_dim=2048, attention_head_dim=(5, 10, 20),
               down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
               up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 1, 1),
               use_linear_projection=True, addition_embed_type="text_time", addition_time_embed_dim=256,
               projection_class_embeddings_input_dim=2816)
    _run_processor(monkeypatch, "cuda", cfg, hw=64, gate=lambda f: 1.2 * f)


SD15_HEADS_UNET = dict(sample_size=16, in_channels=4, out_channels=4, block_out_channels=(320, 640), layers_per_block=1,
                       cross_attention_dim=768, attention_head_dim=8, down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                       up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D"))


@pytest.mark.skipif(not REF.exists(), reason="reference sources not present")
def test_processor_pads_the_sd15_head_sizes_cpu(monkeypatch):
    """SD1.5's geometry: 8 heads of 40 / 80 channels (`attention_head_dim=8` is the head COUNT there) -- not flash-kernel sizes: the
    processor zero-pads the projections to 64 / 96 per head once per module, as the engine's own model classes do at load time."""
    _run_processor(monkeypatch, "cpu", SD15_HEADS_UNET, hw=16, gate=lambda f: max(1.5 * f, 2.5e-2))


@pytest.mark.gpu
def test_processor_pads_the_sd15_head_sizes(monkeypatch):
    _run_processor(monkeypatch, "cuda", SD15_HEADS_UNET, hw=16, gate=lambda f: max(1.2 * f, 1.5e-2))


# ---- hardening of the two binding shims ------------------------------------------------------------
@pytest.mark.skipif(not REF.exists(), reason="reference sources not present")
def test_run_time_enum_member_is_visible_through_every_public_view(monkeypatch):
    """`_extend_enum` writes CPython's private Enum tables; every PUBLIC view must then agree: lookup by value and by name, attribute
    access (the documented `AttentionBackendName.MI355X`; Python >= 3.12 resolves it through the class dict), iteration and
    `__members__`."""
    from diffusers_amd.attention_backend import BACKEND_NAME, _enum_member_is_whole, register_backend
    ref = _load_ref()
    ad = importlib.import_module(ref.__name__ + ".models.attention_dispatch")
    name = register_backend()
    E = ad.AttentionBackendName
    assert name is E(BACKEND_NAME) is E["MI355X"] is E.MI355X
    assert name in list(E) and E.__members__["MI355X"] is name and isinstance(name, E) and name.value == BACKEND_NAME
    assert _enum_member_is_whole(E, "MI355X", BACKEND_NAME)
    assert register_backend() is name                              # idempotent


def test_enum_shim_falls_back_to_an_ungated_slot_when_the_internals_moved():
    """An Enum whose private tables are not the ones the shim knows (simulated: `_member_names_` is not a list any more) must not
    be half-extended silently: `_extend_enum` returns None and `register_backend` registers under an ungated slot with a warning."""
    import enum
    from diffusers_amd import attention_backend as ab

    class Name(str, enum.Enum):
        NATIVE = "native"
        _NATIVE_FLASH = "_native_flash"

    class Broken(str, enum.Enum):
        NATIVE = "native"
    Broken._member_names_ = tuple(Broken._member_names_)          # .append() raises: the shim must notice, not half-register
    assert ab._extend_enum(Broken, "MI355X", "mi355x") is None
    m = ab._extend_enum(Name, "MI355X", "mi355x")
    assert m is not None and Name("mi355x") is m and Name.MI355X is m and m in list(Name)

    class Registry:
        _backends, _constraints, _supported_arg_names = {}, {}, {}
    import types
    fake = types.ModuleType("diffusers.models.attention_dispatch")
    fake.AttentionBackendName, fake._AttentionBackendRegistry = Name, Registry
    saved = {k: sys.modules.get(k) for k in ("diffusers", "diffusers.models", "diffusers.models.attention_dispatch")}
    pkg, models = types.ModuleType("diffusers"), types.ModuleType("diffusers.models")
    pkg.models, models.attention_dispatch = models, fake
    sys.modules.update({"diffusers": pkg, "diffusers.models": models, "diffusers.models.attention_dispatch": fake})
    orig = ab._extend_enum
    try:
        ab._extend_enum = lambda *a, **k: None
        with pytest.warns(RuntimeWarning, match="ungated slot"):
            name = ab.register_backend()
        assert name is Name(ab.UNGATED_SLOTS[0]) and Registry._backends[name] is ab.mi355x_flash_attention
    finally:
        ab._extend_enum = orig
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _attn_module(ref, dev, dim=64, heads=2, cross=None):
    ap = importlib.import_module(ref.__name__ + ".models.attention_processor")
    torch.manual_seed(3)
    return ap.Attention(query_dim=dim, cross_attention_dim=cross, heads=heads, dim_head=dim // heads, bias=False).eval().to(dev).to(bf16)


def _run_processor_cache_rules(monkeypatch, dev):
    """(1) the pack cache is keyed by a weak reference: it empties when the module dies; (2) tensors made under
    `torch.inference_mode()` (no version counter) work and are cached by identity; (3) a cached K / V^T is never used with another
    batch count -- the batch check runs on every call; (4) a self-attention token count that is not a multiple of 8 takes the
    padded-key path instead of raising."""
    import gc
    from diffusers_amd.attention_backend import MI355XAttnProcessor
    ref = _load_ref()
    _env(monkeypatch, dev)
    proc = MI355XAttnProcessor()
    a = _attn_module(ref, dev, cross=64)
    a.set_processor(proc)
    g = torch.Generator().manual_seed(1)
    x = torch.randn((2, 16, 64), generator=g).to(bf16).to(dev)
    e = torch.randn((2, 8, 64), generator=g).to(bf16).to(dev)
    with torch.no_grad():
        y = a(x, encoder_hidden_states=e)
        assert len(proc._packs) == 1
        with pytest.raises(ValueError, match="batch does not match"):       # cache HIT on `e`, other batch count
            a(x[:1], encoder_hidden_states=e)
    with torch.inference_mode():
        xi, ei = x.clone(), e.clone()
        assert ei.is_inference()
        yi = a(xi, encoder_hidden_states=ei)
        yi2 = a(xi, encoder_hidden_states=ei)
    assert torch.equal(yi, yi2) and torch.equal(yi.clone(), y)
    del a
    gc.collect()
    assert len(proc._packs) == 0, "the pack of a freed module is still cached"
    s = _attn_module(ref, dev)
    native = _attn_module(ref, dev)
    native.load_state_dict(s.state_dict())
    s.set_processor(proc)
    xo = torch.randn((2, 13, 64), generator=g).to(bf16).to(dev)              # 13 tokens: not a multiple of 8
    with torch.no_grad():
        got, want = s(xo), native(xo)
    assert got.shape == want.shape and _rel(got, want) < 2e-2


@pytest.mark.skipif(not REF.exists(), reason="reference sources not present")
def test_processor_cache_rules_cpu(monkeypatch):
    _run_processor_cache_rules(monkeypatch, "cpu")


@pytest.mark.gpu
def test_processor_cache_rules(monkeypatch):
    _run_processor_cache_rules(monkeypatch, "cuda")
