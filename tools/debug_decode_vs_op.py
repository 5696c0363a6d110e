#!/usr/bin/env python
"""Debug: thread A decodes (eager) while thread B loops ONE op on another stream -- which op on B corrupts A's decode?"""
import sys
import threading
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.argv = [sys.argv[0]]
from diffusers_amd import factory, init as dinit, ops  # noqa: E402
import tools.debug_ops_concurrent as D  # noqa: E402  (runs its own test once; reused for its CASES)

dev = torch.device("cuda", 0)
vae, _ = factory.build_vae(dinit.SDXL_VAE, seed=1, device=dev, init_device=str(dev))
g = torch.Generator("cpu").manual_seed(3)
z = torch.randn((1, 4, 128, 128), generator=g).to(torch.bfloat16).to(dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def decode(n):
    out = None
    with torch.cuda.stream(sa):
        for _ in range(n):
            out = vae.decode(z, return_dict=False, latents_div=0.13025, postprocess="pt")[0].clone()
        sa.synchronize()
    return out


ref = decode(2)
junk = torch.randn(32 << 20, device=dev)
cases = dict(D.CASES)
cases["torch mul_ (128 MB)"] = lambda i: junk.mul_(1.0001)
cases["torch small add"] = lambda i: junk[:1024].add_(1.0)
for name, fn in cases.items():
    stop = threading.Event()
    res = {}

    def b_loop(fn=fn):
        with torch.cuda.stream(sb):
            n = 0
            while not stop.is_set():
                fn(1)
                n += 1
                if n % 8 == 0:
                    sb.synchronize()
            sb.synchronize()
        res["n"] = n
    tb = threading.Thread(target=b_loop)
    tb.start()
    bad = 0
    for _ in range(3):
        o = decode(1)
        bad += int(not torch.equal(o, ref))
    stop.set()
    tb.join()
    torch.cuda.synchronize()
    print(f"RESULT B loops [{name}] ({res.get('n')} launches): {bad} of 3 decodes on A differ", flush=True)
