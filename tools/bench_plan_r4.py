#!/usr/bin/env python
"""One full-size SDXL denoising step (CFG pair, 128x128 latents) three ways on the same buffers: the Python-driven eager step, the
same launches replayed by da_plan_launch (csrc/plan.hip: a C loop over the recorded C-ABI calls, what a host without Python
gets), and the captured HIP graph the pipelines replay.  Milliseconds per step over back-to-back steps, and the host time
da_plan_launch itself takes (it returns when the last launch is queued).  One JSON line (argv[1]: appended to that file)."""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
bf16 = torch.bfloat16


def main():
    from diffusers_amd import factory, ops, plan as P
    from diffusers_amd.pipelines import _pf
    dev = "cuda"
    pipe = factory.build_sdxl_pipeline(device=dev, tiny=False, seed=0, init_device=dev)
    g = torch.Generator("cpu").manual_seed(1234)
    pe = torch.randn((2, 77, 2048), generator=g).to(bf16).to(dev)
    te = torch.randn((2, 1280), generator=g).to(bf16).to(dev)
    lat0 = torch.randn((1, 4, 128, 128), generator=g).to(bf16).to(dev)
    ids = torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]], device=dev).repeat(2, 1)
    cond = pipe.unet.precompute_conditioning(pe.contiguous(), {"text_embeds": te, "time_ids": ids})
    sch = pipe.scheduler
    sch.set_timesteps(50, device=dev)
    lat = lat0.clone()
    pf = _pf(pipe)

    def reset():
        sch.reset(0)
        lat.copy_(lat0)

    def step(mode="apply"):
        with ops.weight_prefetch(pf, mode):
            pipe._step(lat, cond, 5.0, True)

    reset(); step("record"); step(); reset()
    pl, _ = P.record(step, keep=[pipe, cond, lat, lat0, pe, te, ids])
    torch.cuda.synchronize()
    want = lat.clone()
    reset(); pl.launch(); torch.cuda.synchronize()
    same = bool(torch.equal(lat, want))
    # the HIP graph of the same step
    reset()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr), ops.weight_prefetch(pf, "apply"):
        pipe._step(lat, cond, 5.0, True)
    reset(); gr.replay(); torch.cuda.synchronize()
    same_graph = bool(torch.equal(lat, want))

    def timed(fn, n=20):
        reset(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            fn()
        t_host = (time.perf_counter() - t0) / n
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n, t_host * 1e3

    res = {}
    for name, fn in (("eager python", step), ("da_plan_launch", pl.launch), ("hip graph replay", gr.replay),
                     ("da_plan_launch (2)", pl.launch), ("hip graph replay (2)", gr.replay)):
        ms, host = timed(fn)
        res[name] = {"ms_per_step": round(ms, 3), "host_ms_per_call": round(host, 3)}
    names = {}
    for n in pl.names:
        names[n] = names.get(n, 0) + 1
    rec = {"op": "sdxl step three ways", "launches": len(pl), "by_entry_point": names, "plan_bit_identical_to_eager": same,
           "graph_bit_identical_to_eager": same_graph, **res}
    line = json.dumps(rec)
    print(line)
    if len(sys.argv) > 1:
        open(sys.argv[1], "a").write(line + "\n")


if __name__ == "__main__":
    main()
