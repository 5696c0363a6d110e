"""Text encoders on the engine's kernels vs the ``transformers`` modules they replace (the reference pipelines' text_encoder
slots).  The oracle is transformers itself (the reference's pinned third-party dependency, installed in this image) run in
fp32 on CPU on tiny random-weight configs with the real head size (64); the engine runs in bf16 -- on the torch stand-ins of
the kernels here (host logic: weight fusion, padding, bias tables, EOS pooling), on the HIP kernels in the -m gpu twin.
Tolerance: relative rms <= 2.5e-2 vs fp32, the bound of every model test in this suite."""
import pytest
import torch

import ops_emulation
from diffusers_amd import ops
from diffusers_amd import text_encoders as TE

bf16 = torch.bfloat16
TOL = 2.5e-2


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


def _bf16_sd(model):
    return {k: v.detach().to(bf16) for k, v in model.state_dict().items()}


def _fp32_of_bf16(model):
    """Round the oracle's weights to bf16 so both sides hold the same parameters (the engine stores bf16)."""
    with torch.no_grad():
        for p_ in model.parameters():
            p_.copy_(p_.to(bf16).float())
    return model.eval()


def clip_pair(act="quick_gelu", proj=False, seed=0):
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    torch.manual_seed(seed)
    cfg = CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                         max_position_embeddings=77, projection_dim=64, hidden_act=act, eos_token_id=999, bos_token_id=998,
                         pad_token_id=0)
    ref = _fp32_of_bf16((CLIPTextModelWithProjection if proj else CLIPTextModel)(cfg))
    return cfg, ref


def t5_pair(umt5=False, seed=1):
    from transformers import T5Config, T5EncoderModel, UMT5Config, UMT5EncoderModel
    torch.manual_seed(seed)
    kw = dict(vocab_size=500, d_model=128, d_kv=64, d_ff=256, num_layers=3, num_heads=2, feed_forward_proj="gated-gelu",
              relative_attention_num_buckets=16, relative_attention_max_distance=32)
    cfg = (UMT5Config if umt5 else T5Config)(**kw)
    ref = _fp32_of_bf16((UMT5EncoderModel if umt5 else T5EncoderModel)(cfg))
    return cfg, ref


def clip_ids(B, S, cfg, seed=3):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, 900, (B, S), generator=g)
    ids[:, 0] = cfg.bos_token_id
    for b in range(B):
        e = 5 + 7 * b
        ids[b, e] = cfg.eos_token_id
        ids[b, e + 1:] = cfg.pad_token_id
    return ids


def check_clip(dev, act, proj):
    cfg, ref = clip_pair(act, proj)
    eng = (TE.CLIPTextModelWithProjection if proj else TE.CLIPTextModel)(cfg)
    eng.load_state_dict(_bf16_sd(ref), device=dev)
    ids = clip_ids(2, 77, cfg)
    with torch.no_grad():
        want = ref(ids, output_hidden_states=True)
    got = eng(ids.to(dev), output_hidden_states=True)
    assert len(got.hidden_states) == len(want.hidden_states) == cfg.num_hidden_layers + 1
    r_pen = _rel(got.hidden_states[-2], want.hidden_states[-2])        # what the SD / SDXL pipelines read
    r_last = _rel(got.last_hidden_state, want.last_hidden_state)
    first = _rel(got[0], want[0])
    print(f"[parity] CLIP text ({act}, projection={proj}) on {dev}: hidden[-2] {r_pen:.3e}, last {r_last:.3e}, [0] {first:.3e}")
    assert got[0].shape == want[0].shape and got[0].ndim == (2 if proj else 3)
    assert max(r_pen, r_last, first) < TOL
    if proj:
        assert _rel(got.text_embeds, want.text_embeds) < TOL
    else:
        assert _rel(got.pooler_output, want.pooler_output) < TOL
    short = eng(ids[:1, :20].to(dev))                                   # a shorter (non-multiple-of-16) sequence
    with torch.no_grad():
        assert _rel(short.last_hidden_state, ref(ids[:1, :20]).last_hidden_state) < TOL


def check_t5(dev, umt5, masked):
    cfg, ref = t5_pair(umt5)
    eng = (TE.UMT5EncoderModel if umt5 else TE.T5EncoderModel)(cfg)
    eng.load_state_dict(_bf16_sd(ref), device=dev)
    g = torch.Generator().manual_seed(5)
    B, S = 2, 48
    ids = torch.randint(1, 500, (B, S), generator=g)
    mask = torch.ones((B, S), dtype=torch.long)
    if masked:
        mask[0, 30:] = 0
        mask[1, 41:] = 0
        ids = ids * mask
    with torch.no_grad():
        want = ref(ids, attention_mask=mask if masked else None).last_hidden_state
    got = eng(ids.to(dev), attention_mask=mask.to(dev) if masked else None)
    assert got[0] is got.last_hidden_state and got[0].shape == want.shape
    valid = mask.bool()
    rr = _rel(got.last_hidden_state.cpu()[valid], want[valid])          # padded positions are trimmed by the pipelines
    print(f"[parity] {'UMT5' if umt5 else 'T5'} encoder (mask={masked}) on {dev}: last_hidden_state rel_rms {rr:.3e}")
    assert rr < TOL
    # output_hidden_states: T5Stack's tuple = (embeddings, each block's output ..., the final-normed last state)
    with torch.no_grad():
        wh = ref(ids, attention_mask=mask if masked else None, output_hidden_states=True).hidden_states
    gh = eng(ids.to(dev), attention_mask=mask.to(dev) if masked else None, output_hidden_states=True).hidden_states
    assert len(gh) == len(wh) == cfg.num_layers + 1
    for a, b in zip(gh, wh):
        assert _rel(a.cpu()[valid], b[valid]) < TOL


@pytest.fixture
def emulated(monkeypatch):
    ops_emulation.install(monkeypatch, ops)
    monkeypatch.setattr(ops, "TUNING", False)


@pytest.mark.parametrize("act,proj", [("quick_gelu", False), ("gelu", True)])
def test_clip_text_encoders_host_logic(emulated, act, proj):
    check_clip("cpu", act, proj)


@pytest.mark.parametrize("umt5,masked", [(False, False), (True, True), (False, True)])
def test_t5_encoders_host_logic(emulated, umt5, masked):
    check_t5("cpu", umt5, masked)


def test_relative_position_bucket_matches_transformers():
    from transformers.models.t5.modeling_t5 import T5Attention
    rp = torch.arange(-300, 300)[None, :] - torch.arange(0, 7)[:, None]
    for nb, md in ((32, 128), (16, 32)):
        assert torch.equal(TE.relative_position_bucket(rp, nb, md),
                           T5Attention._relative_position_bucket(rp, bidirectional=True, num_buckets=nb, max_distance=md))


def test_refusals():
    from transformers import CLIPTextConfig, T5Config
    with pytest.raises(ValueError):
        TE.CLIPTextModel(CLIPTextConfig(hidden_size=96, num_attention_heads=2))          # head size 48
    with pytest.raises(ValueError):
        TE.T5EncoderModel(T5Config(d_kv=64, feed_forward_proj="relu"))                   # original (non-gated) T5
    cfg, ref = clip_pair()
    eng = TE.CLIPTextModel(cfg)
    with pytest.raises(RuntimeError):
        eng(torch.zeros((1, 8), dtype=torch.long))                                       # not loaded
    with pytest.raises(ValueError):
        eng.to(torch.float16)


@pytest.mark.gpu
@pytest.mark.parametrize("act,proj", [("quick_gelu", False), ("gelu", True)])
def test_clip_text_encoders_on_gpu(act, proj):
    check_clip("cuda", act, proj)


@pytest.mark.gpu
@pytest.mark.parametrize("umt5,masked", [(False, False), (True, True)])
def test_t5_encoders_on_gpu(umt5, masked):
    check_t5("cuda", umt5, masked)


def test_sdxl_encode_prompt_with_engine_encoders(emulated):
    """The SDXL text front end end to end: tokenizers -> BOTH encoders on the engine (CLIP-L-like quick_gelu model +
    projection model) -> penultimate hidden states concatenated + pooled projection, CFG negatives included, through the same
    ``encode_prompt_sdxl`` the pipelines use -- against the transformers encoders in the same slots."""
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    from diffusers_amd.text_encoding import encode_prompt_sdxl
    from test_text_encoding import _tokenizer
    tok, nv = _tokenizer(max_len=16)
    torch.manual_seed(11)
    mk = lambda cls, act: _fp32_of_bf16(cls(CLIPTextConfig(  # noqa: E731
        vocab_size=nv, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
        max_position_embeddings=16, projection_dim=64, hidden_act=act, bos_token_id=nv - 2, eos_token_id=nv - 1,
        pad_token_id=nv - 1)))
    r1, r2 = mk(CLIPTextModel, "quick_gelu"), mk(CLIPTextModelWithProjection, "gelu")
    e1, e2 = TE.CLIPTextModel(r1.config), TE.CLIPTextModelWithProjection(r2.config)
    e1.load_state_dict(_bf16_sd(r1), device="cpu")
    e2.load_state_dict(_bf16_sd(r2), device="cpu")
    kw = dict(prompt=["hello a cat", "cat"], negative_prompt=["hello", ""], device="cpu", num_images_per_prompt=2,
              do_classifier_free_guidance=True, force_zeros_for_empty_prompt=False)
    want = encode_prompt_sdxl([tok, tok], [r1, r2], dtype=torch.float32, **kw)
    got = encode_prompt_sdxl([tok, tok], [e1, e2], **kw)
    for name, g_, w_ in zip(("prompt_embeds", "negative_prompt_embeds", "pooled", "negative_pooled"), got, want):
        assert g_.shape == w_.shape and g_.dtype == bf16
        rr = _rel(g_, w_)
        print(f"[parity] SDXL encode_prompt on engine encoders: {name} rel_rms {rr:.3e}")
        assert rr < TOL
