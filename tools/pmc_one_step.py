#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc traffic passes: ONE process, eager launches, a few SDXL denoising steps at the bench
configuration (batch 2, 128x128 latents) and nothing else -- no HIP graph, no tuning, no decode.  (The full bench.py under
--pmc dies inside rocprofv3 on this image, with or without graph replay; this keeps the dispatch count small.)"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    import bench
    from diffusers_amd import factory, init as dinit
    from diffusers_amd.pipelines import StableDiffusionXLPipeline
    from diffusers_amd.schedulers import EulerDiscreteScheduler
    dev = torch.device("cuda", 0)
    unet, _ = factory.build_unet(dinit.SDXL_UNET, seed=0, device=dev, init_device=str(dev))
    pipe = StableDiffusionXLPipeline(vae=None, unet=unet, scheduler=EulerDiscreteScheduler(**factory.SDXL_SCHEDULER))
    inp = bench.synth_inputs(1, False, dev)
    pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], dim=0).contiguous()
    te = torch.cat([inp["negative_pooled"], inp["pooled"]], dim=0)
    ids = torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]], device=dev).repeat(2, 1)
    cond = unet.precompute_conditioning(pe, {"text_embeds": te, "time_ids": ids})
    pipe.scheduler.set_timesteps(50, device=dev)
    lat = inp["latents"].clone()
    pipe.scheduler.reset(0)
    log = _launch_log() if len(sys.argv) > 2 else None
    for _ in range(steps):
        pipe._step(lat, cond, bench.GUIDANCE, True)
    torch.cuda.synchronize()
    if log is not None:
        import json
        Path(sys.argv[2]).write_text(json.dumps(log["rows"]))
    print("pmc_one_step: done", steps, flush=True)


def _launch_log():
    """Round 6: note every implicit-GEMM launch of the steps in issue order -- (population label, algorithmic bytes = each operand
    and the output once, flop) -- so that tools/pmc_traffic.py can give the PMC rows (which carry only kernel names) one table row
    per launch POPULATION (shape x role) instead of one per kernel family.  Same wrappers as bench.instrumented_pass, no events."""
    from diffusers_amd import ops
    rows = []
    orig = {n: getattr(ops, n) for n in ("linear", "linear_pair", "conv2d_nhwc", "linear_qkv")}

    def lin(*a, **k):
        out = orig["linear"](*a, **k)
        x, w = (a[0] if a else k["x"]), (a[1] if len(a) > 1 else k["w"])
        if x.shape[0] > 8:
            act = k.get("act")
            geglu = act in (ops.L.ACT_GEGLU, ops.L.ACT_GEGLU_TANH) if act is not None else False
            n_out = out.shape[1]
            extra = sum(t.numel() for t in (k.get("residual"),) if t is not None)
            rows.append({"pop": f"{'geglu' if geglu else 'lin'} M{x.shape[0]} N{w.shape[0]} K{x.shape[1]}"
                                + (" +res" if k.get("residual") is not None else ""),
                         "bytes": 2.0 * (x.numel() + w.numel() + out.shape[0] * n_out + extra),
                         "flop": 2.0 * x.shape[0] * w.shape[0] * x.shape[1]})
        return out

    def pair(*a, **k):
        n0 = len(rows)
        out = orig["linear_pair"](*a, **k)
        if len(rows) == n0 and all(pr["x"].shape[0] > 8 for pr in a):     # (a fallback to two linear() launches logged itself)
            rows.append({"pop": "pair " + " | ".join(f"M{pr['x'].shape[0]} N{pr['w'].shape[0]} K{pr['x'].shape[1]}" for pr in a),
                         "bytes": sum(2.0 * (pr["x"].numel() + pr["w"].numel() + o.numel()) for pr, o in zip(a, out)),
                         "flop": sum(2.0 * pr["x"].shape[0] * pr["w"].shape[0] * pr["x"].shape[1] for pr in a)})
        return out

    def qkv(*a, **k):
        n0 = len(rows)
        out = orig["linear_qkv"](*a, **k)
        del rows[n0:]                                   # (linear_qkv may go through ops.linear: one launch, one row)
        x, w = (a[0] if a else k["x"]), (a[1] if len(a) > 1 else k["w"])
        rows.append({"pop": f"qkv M{x.shape[0]} N{w.shape[0]} K{x.shape[1]}", "bytes": 2.0 * (x.numel() + w.numel() + x.shape[0] * w.shape[0]),
                     "flop": 2.0 * x.shape[0] * w.shape[0] * x.shape[1]})
        return out

    def conv(*a, **k):
        out = orig["conv2d_nhwc"](*a, **k)
        x, w = a[0], a[1]
        x2 = k.get("x2")
        m = out.shape[0] * out.shape[1] * out.shape[2]
        kk = w.numel() // w.shape[0]
        extra = sum(t.numel() for t in (k.get("residual"),) if t is not None)
        ks = k.get("ksize", 3)
        rows.append({"pop": f"conv{ks}x{ks} M{m} N{w.shape[0]} K{kk}" + (" up" if k.get("up") else "")
                            + (" s2" if k.get("stride", 1) == 2 else ""),
                     "bytes": 2.0 * (x.numel() + (x2.numel() if x2 is not None else 0) + w.numel() + out.numel() + extra),
                     "flop": 2.0 * m * w.shape[0] * kk})
        return out
    ops.linear, ops.linear_pair, ops.conv2d_nhwc = lin, pair, conv
    if "linear_qkv" in orig:
        ops.linear_qkv = qkv
    return {"rows": rows}


if __name__ == "__main__":
    main()
