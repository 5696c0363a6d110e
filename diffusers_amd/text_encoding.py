"""Text-encoder front end of the CLIP-conditioned pipelines (the step in front of the hot path).

The encoders themselves are the caller's ``transformers`` modules (CLIPTextModel / CLIPTextModelWithProjection) running
on PyTorch-ROCm -- they are not re-implemented here; this module is the host logic around them that the reference keeps
inside its pipelines: tokenise with max-length padding, pick the hidden state the model family conditions on, build the
classifier-free-guidance negatives, duplicate per image, and hand bf16 tensors to the denoiser.  Mirrors

  * ``StableDiffusionPipeline.encode_prompt``     pipelines/stable_diffusion/pipeline_stable_diffusion.py:332-513
  * ``StableDiffusionXLPipeline.encode_prompt``   pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl.py:283-518

(same arguments, same errors; LoRA scaling and textual inversion are loader features outside this engine).  In a
multi-GPU job rank 0 encodes and the embeddings are broadcast (``distributed.broadcast_tensors``), so the text encoders
are loaded once per node, not once per GPU.
"""
from __future__ import annotations

import logging
from typing import List, Optional, Sequence, Tuple, Union

import torch

logger = logging.getLogger(__name__)
Prompt = Union[str, List[str]]


def _as_list(p: Prompt) -> List[str]:
    return [p] if isinstance(p, str) else list(p)


def _tokenize(tokenizer, prompt: Sequence[str], max_length: Optional[int] = None):
    ids = tokenizer(list(prompt), padding="max_length", max_length=max_length or tokenizer.model_max_length,
                    truncation=True, return_tensors="pt")
    if max_length is None:
        untruncated = tokenizer(list(prompt), padding="longest", return_tensors="pt").input_ids
        if untruncated.shape[-1] >= ids.input_ids.shape[-1] and not torch.equal(ids.input_ids, untruncated):
            removed = tokenizer.batch_decode(untruncated[:, tokenizer.model_max_length - 1: -1])
            logger.warning("The following part of your input was truncated because CLIP can only handle sequences up to"
                           f" {tokenizer.model_max_length} tokens: {removed}")
    return ids


def _check_negative(prompt: Optional[Prompt], negative_prompt: Prompt, batch_size: int) -> None:
    if prompt is not None and type(prompt) is not type(negative_prompt):
        raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} !="
                        f" {type(prompt)}.")
    if not isinstance(negative_prompt, str) and batch_size != len(negative_prompt):
        raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`:"
                         f" {prompt} has batch size {batch_size}. Please make sure that passed `negative_prompt` matches"
                         " the batch size of `prompt`.")


def _repeat(t: torch.Tensor, n: int) -> torch.Tensor:
    """[B][...] -> [B * n][...], each prompt's copies adjacent (the reference's repeat + view)."""
    if n == 1:
        return t
    return t.repeat_interleave(n, dim=0)


@torch.no_grad()
def encode_prompt_sd(tokenizer, text_encoder, prompt: Prompt, device, num_images_per_prompt: int = 1,
                     do_classifier_free_guidance: bool = True, negative_prompt: Optional[Prompt] = None,
                     clip_skip: Optional[int] = None, dtype: torch.dtype = torch.bfloat16
                     ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """SD 1.x / 2.x: one CLIP text encoder, last hidden state (or, with ``clip_skip``, an earlier one passed through the
    final layer norm).  Returns (prompt_embeds, negative_prompt_embeds) in ``dtype`` on ``device``."""
    batch_size = 1 if isinstance(prompt, str) else len(prompt)
    use_mask = bool(getattr(text_encoder.config, "use_attention_mask", False))

    def run(texts, max_length=None, skip=None):
        tok = _tokenize(tokenizer, texts, max_length)
        mask = tok.attention_mask.to(device) if use_mask else None
        if skip is None:
            return text_encoder(tok.input_ids.to(device), attention_mask=mask)[0]
        out = text_encoder(tok.input_ids.to(device), attention_mask=mask, output_hidden_states=True)
        # (transformers >= 5 flattened CLIPTextModel: the norm sits on the model itself, not on `.text_model`)
        return getattr(text_encoder, "text_model", text_encoder).final_layer_norm(out[-1][-(skip + 1)])

    pe = run(_as_list(prompt), skip=clip_skip)
    ne = None
    if do_classifier_free_guidance:
        if negative_prompt is None:
            uncond = [""] * batch_size
        else:
            _check_negative(prompt, negative_prompt, batch_size)
            uncond = _as_list(negative_prompt)
        ne = run(uncond, max_length=pe.shape[1])
        ne = _repeat(ne.to(device=device, dtype=dtype), num_images_per_prompt)
    return _repeat(pe.to(device=device, dtype=dtype), num_images_per_prompt), ne


@torch.no_grad()
def encode_prompt_sdxl(tokenizers: Sequence, text_encoders: Sequence, prompt: Prompt, prompt_2: Optional[Prompt] = None,
                       device=None, num_images_per_prompt: int = 1, do_classifier_free_guidance: bool = True,
                       negative_prompt: Optional[Prompt] = None, negative_prompt_2: Optional[Prompt] = None,
                       force_zeros_for_empty_prompt: bool = True, clip_skip: Optional[int] = None,
                       dtype: torch.dtype = torch.bfloat16):
    """SDXL: the penultimate hidden states of both encoders concatenated on the feature axis, and the pooled output of
    the LAST encoder (the one with the projection head).  ``tokenizers`` / ``text_encoders`` are [first, second], or
    [second] alone (the refiner layout).  Returns (prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds,
    negative_pooled_prompt_embeds)."""
    if len(tokenizers) != len(text_encoders) or not 1 <= len(tokenizers) <= 2:
        raise ValueError("encode_prompt_sdxl: pass one or two (tokenizer, text_encoder) pairs")
    prompt_l = _as_list(prompt)
    batch_size = len(prompt_l)
    # the reference zips [prompt, prompt_2] with the encoder list: a lone (second) encoder therefore reads `prompt`
    prompts = [prompt_l, _as_list(prompt_2) if prompt_2 else prompt_l][:len(tokenizers)]

    def run(texts_per_encoder, max_length=None, skip=None):
        hidden, pooled = [], None
        for texts, tok, enc in zip(texts_per_encoder, tokenizers, text_encoders):
            ids = _tokenize(tok, texts, max_length).input_ids.to(device)
            out = enc(ids, output_hidden_states=True)
            if pooled is None and out[0].ndim == 2:       # only the projection model returns a pooled [B][D] first
                pooled = out[0]
            hidden.append(out.hidden_states[-2] if skip is None else out.hidden_states[-(skip + 2)])
        return torch.concat(hidden, dim=-1), pooled

    pe, pooled = run(prompts, skip=clip_skip)
    ne = npooled = None
    if do_classifier_free_guidance:
        if negative_prompt is None and force_zeros_for_empty_prompt:
            ne, npooled = torch.zeros_like(pe), torch.zeros_like(pooled)
        else:
            neg = negative_prompt or ""
            neg2 = negative_prompt_2 or neg
            # the SDXL reference normalises BOTH sides to lists before its type check
            # (pipeline_stable_diffusion_xl.py:360, :431-446): a str negative prompt broadcasts over a list of prompts
            neg_l = batch_size * [neg] if isinstance(neg, str) else neg
            neg2_l = batch_size * [neg2] if isinstance(neg2, str) else neg2
            _check_negative(prompt_l, neg_l, batch_size)
            neg_l, neg2_l = list(neg_l), list(neg2_l)
            uncond = [neg_l, neg2_l][:len(tokenizers)]
            ne, npooled = run(uncond, max_length=pe.shape[1])
        ne = _repeat(ne.to(device=device, dtype=dtype), num_images_per_prompt)
        npooled = _repeat(npooled.to(device=device, dtype=dtype), num_images_per_prompt)
    pe = _repeat(pe.to(device=device, dtype=dtype), num_images_per_prompt)
    pooled = _repeat(pooled.to(device=device, dtype=dtype), num_images_per_prompt)
    return pe, ne, pooled, npooled


# ----------------------------------------------------------------------------------------------------------------------
# T5-conditioned families
# ----------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def encode_prompt_flux(tokenizer, text_encoder, tokenizer_2, text_encoder_2, prompt: Prompt, prompt_2: Optional[Prompt] = None,
                       device=None, num_images_per_prompt: int = 1, max_sequence_length: int = 512,
                       dtype: torch.dtype = torch.bfloat16):
    """``FluxPipeline.encode_prompt`` (pipelines/flux/pipeline_flux.py:219-399): the CLIP encoder's pooled output on
    ``prompt`` and the T5 encoder's last hidden state on ``prompt_2`` (max_length padding, no attention mask), plus the
    all-zero text position ids.  Returns (prompt_embeds, pooled_prompt_embeds, text_ids)."""
    p1 = _as_list(prompt)
    p2 = _as_list(prompt_2) if prompt_2 else p1
    ids = tokenizer(p1, padding="max_length", max_length=tokenizer.model_max_length, truncation=True,
                    return_overflowing_tokens=False, return_length=False, return_tensors="pt").input_ids
    pooled = text_encoder(ids.to(device), output_hidden_states=False).pooler_output
    pooled = _repeat(pooled.to(device=device, dtype=dtype), num_images_per_prompt)
    ids2 = tokenizer_2(p2, padding="max_length", max_length=max_sequence_length, truncation=True, return_length=False,
                       return_overflowing_tokens=False, return_tensors="pt").input_ids
    untruncated = tokenizer_2(p2, padding="longest", return_tensors="pt").input_ids
    if untruncated.shape[-1] >= ids2.shape[-1] and not torch.equal(ids2, untruncated):
        logger.warning(f"part of the prompt was truncated because `max_sequence_length` is {max_sequence_length} tokens")
    pe = text_encoder_2(ids2.to(device), output_hidden_states=False)[0]
    pe = _repeat(pe.to(device=device, dtype=dtype), num_images_per_prompt)
    text_ids = torch.zeros(pe.shape[1], 3, device=device, dtype=dtype)
    return pe, pooled, text_ids


def prompt_clean(text: str) -> str:
    """pipelines/wan/pipeline_wan.py:78-93: ftfy (when installed), double html-unescape, whitespace collapse."""
    import html
    import re
    try:
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip()


@torch.no_grad()
def encode_prompt_wan(tokenizer, text_encoder, prompt: Prompt, negative_prompt: Optional[Prompt] = None,
                      do_classifier_free_guidance: bool = True, num_videos_per_prompt: int = 1,
                      max_sequence_length: int = 226, device=None, dtype: torch.dtype = torch.bfloat16):
    """``WanPipeline.encode_prompt`` (pipelines/wan/pipeline_wan.py:158-279): UMT5 last hidden state WITH the attention
    mask, positions past each prompt's length zeroed (trim + re-pad, :187-190).  Returns (prompt_embeds,
    negative_prompt_embeds)."""
    def t5(texts):
        texts = [prompt_clean(u) for u in texts]
        tok = tokenizer(texts, padding="max_length", max_length=max_sequence_length, truncation=True, add_special_tokens=True,
                        return_attention_mask=True, return_tensors="pt")
        mask = tok.attention_mask
        h = text_encoder(tok.input_ids.to(device), mask.to(device)).last_hidden_state.to(device=device, dtype=dtype)
        keep = (torch.arange(h.shape[1])[None, :] < mask.gt(0).sum(dim=1)[:, None]).to(h.device)
        return _repeat(h * keep[..., None].to(h.dtype), num_videos_per_prompt)

    prompt_l = _as_list(prompt)
    pe = t5(prompt_l)
    ne = None
    if do_classifier_free_guidance:
        neg = negative_prompt or ""
        neg_l = len(prompt_l) * [neg] if isinstance(neg, str) else list(neg)
        # (the reference compares against the LIST form of `prompt`, so a str negative with a list prompt is accepted)
        if not isinstance(neg_l, list):
            raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(neg_l)} != {list}.")
        if len(neg_l) != len(prompt_l):
            raise ValueError(f"`negative_prompt`: {neg_l} has batch size {len(neg_l)}, but `prompt`: {prompt_l} has batch "
                             f"size {len(prompt_l)}. Please make sure that passed `negative_prompt` matches the batch size "
                             "of `prompt`.")
        ne = t5(neg_l)
    return pe, ne
