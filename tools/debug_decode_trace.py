#!/usr/bin/env python
"""Debug: trace per-op output checksums of AutoencoderKL.decode under two concurrent threads and report the FIRST op whose
output differs from the sequential run (its inputs were still the sequential run's bits)."""
import sys
import threading
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import factory, init as dinit, ops  # noqa: E402

dev = torch.device("cuda", 0)
vae, _ = factory.build_vae(dinit.SDXL_VAE, seed=1, device=dev, init_device=str(dev))
g = torch.Generator("cpu").manual_seed(3)
zs = [torch.randn((1, 4, 128, 128), generator=g).to(torch.bfloat16).to(dev) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
tls = threading.local()
NAMES = ("conv2d_nhwc", "group_norm_nhwc", "linear", "linear_pair", "conv_thin_in", "conv_thin_out", "softmax_rows", "attention")
orig = {n: getattr(ops, n) for n in NAMES}


def wrap(name):
    def f(*a, **k):
        out = orig[name](*a, **k)
        rec = getattr(tls, "rec", None)
        if rec is not None:
            outs = out if isinstance(out, (tuple, list)) else (out,)
            shape = tuple(outs[0].shape)
            cs = torch.stack([o.float().abs().double().sum() for o in outs]).sum()
            rec.append((name, shape, cs))
        return out
    return f


for n in NAMES:
    setattr(ops, n, wrap(n))
traces = {}


def work(i, count, key):
    with torch.cuda.stream(streams[i]):
        for c in range(count):
            tls.rec = []
            vae.decode(zs[i], return_dict=False, latents_div=0.13025, postprocess="pt")
            traces[(key, i, c)] = tls.rec
        streams[i].synchronize()


for i in range(2):
    t = threading.Thread(target=work, args=(i, 1, "seq"))
    t.start()
    t.join()
torch.cuda.synchronize()
for rnd in range(3):
    th = [threading.Thread(target=work, args=(i, 3, f"con{rnd}")) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    for i in range(2):
        ref = traces[("seq", i, 0)]
        for c in range(3):
            tr = traces[(f"con{rnd}", i, c)]
            first = next((j for j, (a, b) in enumerate(zip(ref, tr)) if float(a[2]) != float(b[2])), None)
            if first is None:
                print(f"RESULT round {rnd} thread {i} decode {c}: identical ({len(tr)} ops)")
            else:
                prev = ref[first - 1][:2] if first else None
                print(f"RESULT round {rnd} thread {i} decode {c}: FIRST differing op #{first} of {len(tr)}: {tr[first][0]} {tr[first][1]} "
                      f"(previous op: {prev}); ops differing in total: {sum(1 for a, b in zip(ref, tr) if float(a[2]) != float(b[2]))}", flush=True)
