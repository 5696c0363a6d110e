#!/usr/bin/env python
"""Debug: two threads decode (AutoencoderKL.decode, eager) concurrently on two streams; compare with the sequential result."""
import os
import sys
import threading
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import factory, init as dinit, ops, tuning  # noqa: E402

if "--launchlock" in sys.argv:
    # serialise the HOST side of every launch of this library across threads (the GPU side stays concurrent)
    from diffusers_amd import _lib as L
    real = L.load()
    lock = threading.Lock()

    class Proxy:
        def __getattr__(self, name):
            fn = getattr(real, name)
            if not name.startswith("da_"):
                return fn

            def locked(*a, **k):
                with lock:
                    return fn(*a, **k)
            return locked
    prox = Proxy()
    L.load = lambda: prox
    print("launch lock installed")
dev = torch.device("cuda", 0)
vae, _ = factory.build_vae(dinit.SDXL_VAE, seed=1, device=dev, init_device=str(dev))
if "--nosplitk" in sys.argv:
    t = tuning.table()
    n = 0
    for k, v in list(t.items()):
        if len(v) > 3 and v[3] > 1:
            t[k] = tuple(v[:3]) + (1,) + tuple(v[4:])
            n += 1
    print("split-K entries switched off:", n)
g = torch.Generator("cpu").manual_seed(3)
zs = [torch.randn((1, 4, 128, 128), generator=g).to(torch.bfloat16).to(dev) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
outs = {}
mode = "pt" if "--raw" not in sys.argv else None


ARENA = None
if "--arena" in sys.argv:
    # every torch.empty / zeros of a decode comes out of a per-thread bump arena (no reuse inside a decode, no caching allocator)
    ARENA = [torch.empty(12 << 30, dtype=torch.uint8, device=dev) for _ in range(2)]
    tl = threading.local()
    real_empty, real_zeros = torch.empty, torch.zeros

    def carve(shape, dtype, zero):
        if isinstance(shape, int):
            shape = (shape,)
        shape = tuple(int(x) for x in shape)
        n = 1
        for x in shape:
            n *= x
        nb = n * real_empty((), dtype=dtype).element_size()
        off = (tl.cur + 255) & ~255
        tl.cur = off + nb + 4096
        t = ARENA[tl.idx][off:off + nb].view(dtype).view(shape)
        return t.zero_() if zero else t

    def patched(zero):
        def f(*size, **kw):
            if getattr(tl, "idx", None) is None or kw.get("device") is None or torch.device(kw["device"]).type != "cuda":
                return (real_zeros if zero else real_empty)(*size, **kw)
            shape = size[0] if len(size) == 1 and not isinstance(size[0], int) else size
            return carve(shape, kw.get("dtype", torch.float32), zero)
        return f
    torch.empty, torch.zeros = patched(False), patched(True)
    print("per-thread arenas installed")


def work(i, count):
    with torch.cuda.stream(streams[i]):
        for _ in range(count):
            if ARENA is not None:
                tl.idx, tl.cur = i, 0
                streams[i].synchronize()          # (the arena is rewound: the previous decode must have drained)
            outs[i] = vae.decode(zs[i], return_dict=False, latents_div=0.13025, postprocess=mode)[0].clone()
        streams[i].synchronize()


def run(concurrent, count):
    th = [threading.Thread(target=work, args=(i, count)) for i in range(2)]
    if concurrent:
        [t.start() for t in th]
        [t.join() for t in th]
    else:
        for t in th:
            t.start()
            t.join()
    torch.cuda.synchronize()


run(False, 2)
ref = {i: outs[i].clone() for i in range(2)}
bad = 0
for k in range(4):
    run(True, 4)
    for i in range(2):
        d = (outs[i].float() - ref[i].float()).abs()
        nz = float((d > 0).float().mean())
        bad += nz > 0
        print(f"concurrent {k}: decode {i}: max |diff| {float(d.max()):.3e}, differing {100 * nz:.3f} %", flush=True)
print("RESULT", " ".join(sys.argv[1:]), {k: os.environ.get(k) for k in ("DIFFUSERS_AMD_PREFETCH", "DA_GN_FUSED", "DIFFUSERS_AMD_TUNE", "DA_CONV_IN_QUAD")},
      "mismatching decodes:", bad, "splitk_error:", ops.splitk_error())
