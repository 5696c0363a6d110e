#!/usr/bin/env python
"""Experiment: the CFG pair of one SDXL denoising step as ONE batch-2 U-Net forward on one stream (what the pipeline
does) versus TWO batch-1 forwards on two streams inside one HIP graph (fork / join).  The second form halves every
launch's grid but lets one branch's launch ramp / prologue / epilogue / inter-kernel gap hide under the other's main loop.
Prints one JSON line."""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from diffusers_amd import factory, init as dinit  # noqa: E402


def replay_ms(g, n=20):
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


def main():
    dev = torch.device("cuda:0")
    unet, _ = factory.build_unet(dinit.SDXL_UNET, seed=0, device=dev, init_device="cuda:0")
    inp = bench.synth_inputs(1, False, dev)
    pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]]).contiguous()
    te = torch.cat([inp["negative_pooled"], inp["pooled"]]).contiguous()
    ids = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * 2, dtype=torch.float32, device=dev)
    lat = torch.cat([inp["latents"], inp["latents"]]).contiguous()
    cond2 = unet.precompute_conditioning(pe, {"text_embeds": te, "time_ids": ids})
    cond1 = [unet.precompute_conditioning(pe[i:i + 1].contiguous(), {"text_embeds": te[i:i + 1].contiguous(), "time_ids": ids[i:i + 1].contiguous()})
             for i in range(2)]
    lat1 = [lat[i:i + 1].contiguous() for i in range(2)]
    t = torch.tensor([500.0], device=dev)

    def fwd2():
        return unet(lat, t, None, conditioning=cond2, return_dict=False)[0]

    def fwd1(i):
        return unet(lat1[i], t, None, conditioning=cond1[i], return_dict=False)[0]

    ref = fwd2()          # eager: tunes unseen shapes
    a, b = fwd1(0), fwd1(1)
    torch.cuda.synchronize()
    same = bool(torch.equal(torch.cat([a, b]), ref))
    err = float((torch.cat([a, b]).float() - ref.float()).abs().max())

    s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s_main):
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=s_main):
            o2 = fwd2()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=s_main):
            s_side.wait_stream(s_main)
            oa = fwd1(0)
            with torch.cuda.stream(s_side):
                ob = fwd1(1)
            s_main.wait_stream(s_side)
        gs = torch.cuda.CUDAGraph()   # the two batch-1 forwards back to back on ONE stream (isolates the overlap)
        with torch.cuda.graph(gs, stream=s_main):
            oc, od = fwd1(0), fwd1(1)
        ms2, ms1, mss = replay_ms(g2), replay_ms(g1), replay_ms(gs)
        ms2b, ms1b = replay_ms(g2), replay_ms(g1)
        # variant B: the batch-1 launches use the variant the table holds for the batch-2 shape (tuned with the GPU to
        # itself, a half-size launch prefers small tiles to fill 256 CUs alone; next to its twin it does not have to)
        import re
        from diffusers_amd import tuning
        tab, swapped = tuning.table(), 0
        solo = dict(tab)
        for k in list(tab):
            m = re.search(r":M(\d+):", k)
            if not m:
                continue
            k2 = k.replace(f":M{m.group(1)}:", f":M{2 * int(m.group(1))}:")
            if k2 in tab and k2 != k and tab[k2][:2] != tab[k][:2] and int(m.group(1)) in (1024, 4096, 16384):
                tab[k] = tab[k2]
                swapped += 1
        g1b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1b, stream=s_main):
            s_side.wait_stream(s_main)
            oe = fwd1(0)
            with torch.cuda.stream(s_side):
                of = fwd1(1)
            s_main.wait_stream(s_side)
        ms1c = min(replay_ms(g1b), replay_ms(g1b))
        same_b = bool(torch.equal(torch.cat([oe, of]), o2))
        # variant C: branch B starts only after branch A's k-th GEMM / attention launch, so the twins run out of phase
        # (one's GEMM main loop next to the other's attention / norm instead of two copies of the same kernel)
        tab.update(solo)
        from diffusers_amd import ops as _ops
        stagger = {}
        for kth in (2, 5, 9, 14, 20, 40):
            cnt = {"n": 0}
            ev = torch.cuda.Event()
            orig_lin, orig_att = _ops.linear, _ops.attention

            def hook(fn):
                def w(*a, **k):
                    out = fn(*a, **k)
                    cnt["n"] += 1
                    if cnt["n"] == kth:
                        ev.record(torch.cuda.current_stream())
                    return out
                return w
            gk = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gk, stream=s_main):
                _ops.linear, _ops.attention = hook(orig_lin), hook(orig_att)
                try:
                    og = fwd1(0)
                finally:
                    _ops.linear, _ops.attention = orig_lin, orig_att
                s_side.wait_event(ev)
                with torch.cuda.stream(s_side):
                    oh = fwd1(1)
                s_main.wait_stream(s_side)
            stagger[kth] = [round(min(replay_ms(gk), replay_ms(gk)), 3), bool(torch.equal(torch.cat([og, oh]), o2))]
    print(json.dumps({"op": "cfg_two_stream_stagger", "ms_by_offset_launches": stagger}))
    print(json.dumps({"op": "cfg_two_stream", "batch2_one_stream_ms": round(min(ms2, ms2b), 3),
                      "batch1_x2_two_streams_ms": round(min(ms1, ms1b), 3), "batch1_x2_one_stream_ms": round(mss, 3),
                      "speedup": round(min(ms2, ms2b) / min(ms1, ms1b), 4), "bit_identical_outputs": same,
                      "max_abs_diff": err,
                      "graph_outputs_match": bool(torch.equal(torch.cat([oa, ob]), o2)),
                      "two_streams_with_batch2_variants_ms": round(ms1c, 3), "variants_swapped": swapped,
                      "variant_b_outputs_match": same_b}))


if __name__ == "__main__":
    main()
