#!/usr/bin/env python
"""Debug: two threads REPLAY captured HIP graphs of AutoencoderKL.decode concurrently on two streams (each thread its own graph, its
own capture stream and static buffers) and compare with the one-at-a-time replays."""
import sys
import threading
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import factory, init as dinit  # noqa: E402

dev = torch.device("cuda", 0)
vae, _ = factory.build_vae(dinit.SDXL_VAE, seed=1, device=dev, init_device=str(dev))
g = torch.Generator("cpu").manual_seed(3)
zs = [torch.randn((1, 4, 128, 128), generator=g).to(torch.bfloat16).to(dev) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
graphs, statics = [], []
for i in range(2):
    cs = torch.cuda.Stream()
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        vae.decode(zs[i], return_dict=False, latents_div=0.13025, postprocess="pt")        # warm
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=cs):
            out = vae.decode(zs[i], return_dict=False, latents_div=0.13025, postprocess="pt")[0]
    torch.cuda.synchronize()
    graphs.append(gr)
    statics.append(out)
res = {}


def work(i, n):
    outs = []
    with torch.cuda.stream(streams[i]):
        for _ in range(n):
            graphs[i].replay()
            outs.append(statics[i].clone())
        streams[i].synchronize()
    res[i] = outs


for i in range(2):
    t = threading.Thread(target=work, args=(i, 2))
    t.start()
    t.join()
torch.cuda.synchronize()
ref = {i: res[i][0].clone() for i in range(2)}
print("sequential repeat identical:", all(torch.equal(o, ref[i]) for i in range(2) for o in res[i]))
bad = 0
for k in range(4):
    th = [threading.Thread(target=work, args=(i, 6)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    b = sum(int(not torch.equal(o, ref[i])) for i in range(2) for o in res[i])
    bad += b
    print(f"round {k}: {b} of 12 concurrent graph-replayed decodes differ", flush=True)
print("RESULT graph-replayed decodes differing:", bad, "of 48")
