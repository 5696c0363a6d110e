#!/usr/bin/env python
"""Debug: which op misbehaves when two host threads launch it concurrently on two streams?  Each op runs on per-thread inputs; the
concurrent results are compared with the same thread's sequential result."""
import sys
import threading
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import ops  # noqa: E402

bf16 = torch.bfloat16
dev = "cuda"
g = torch.Generator("cpu").manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(s, generator=g) * scale).to(bf16).to(dev)  # noqa: E731


def case_conv(C, Co, H, up=False):
    x = [rnd(1, H, H, C) for _ in range(2)]
    w4 = rnd(Co, C, 3, 3, scale=0.05)
    w, b = ops.pack_conv_weight(w4), rnd(Co)
    return lambda i: ops.conv2d_nhwc(x[i], w, b, ksize=3, up=up)


def case_gn(B, HW, C):
    x = [rnd(B, HW, C) + 0.5 for _ in range(2)]
    gm, bt = rnd(C) * 0.1 + 1.0, rnd(C, scale=0.1)
    return lambda i: ops.group_norm_nhwc(x[i], gm, bt, 32, 1e-6, silu=True)


def case_linear(M, N, K, f32=False, res=False):
    x = [rnd(M, K) for _ in range(2)]
    w, b = rnd(N, K, scale=K ** -0.5), rnd(N)
    r = [rnd(M, N) for _ in range(2)] if res else [None, None]
    return lambda i: ops.linear(x[i], w, None if f32 else b, out_f32=f32, residual=r[i])


def case_pair(M, C):
    x = [rnd(M, C) for _ in range(2)]
    wqk, bqk, wv = rnd(2 * C, C, scale=C ** -0.5), rnd(2 * C), rnd(C, C, scale=C ** -0.5)
    return lambda i: torch.cat([t.reshape(-1)[:100000] for t in ops.linear_pair({"x": x[i], "w": wqk, "bias": bqk}, {"x": wv, "w": x[i]})])


def case_softmax(M, N):
    s = [torch.randn((M, N), generator=g).to(dev) for _ in range(2)]
    return lambda i: ops.softmax_rows(s[i])


def case_thin_out(C, H):
    x = [rnd(1, H, H, C) for _ in range(2)]
    w, b = ops.pack_conv_weight(rnd(3, C, 3, 3, scale=0.05)), rnd(3)
    return lambda i: ops.conv_thin_out(x[i], w, b, postprocess="pt")


def case_thin_in():
    z = [rnd(1, 4, 128, 128) for _ in range(2)]
    w, b = ops.pack_conv_weight(rnd(512, 4, 3, 3, scale=0.2)), rnd(512)
    return lambda i: ops.conv_thin_in(z[i], w, b, ksize=3, in_nchw=True, in_div=0.13025)


CASES = {
    "conv3x3 512->512 @128": case_conv(512, 512, 128), "conv3x3 512->512 @256 up": case_conv(512, 512, 128, up=True),
    "conv3x3 256->256 @512": case_conv(256, 256, 512), "conv3x3 128->128 @1024": case_conv(128, 128, 1024),
    "groupnorm (1,16384,512)": case_gn(1, 16384, 512), "groupnorm (1,262144,256)": case_gn(1, 262144, 256),
    "groupnorm (1,1048576,128)": case_gn(1, 1048576, 128),
    "linear 16384x512x512 +res": case_linear(16384, 512, 512, res=True), "linear scores f32 4096x16384x512": case_linear(4096, 16384, 512, f32=True),
    "linear PV 4096x512x16384": case_linear(4096, 512, 16384), "linear_pair qk|vt": case_pair(16384, 512),
    "softmax_rows 4096x16384": case_softmax(4096, 16384), "conv_thin_out 128->3 @1024": case_thin_out(128, 1024), "conv_thin_in 4->512": case_thin_in(),
}
streams = [torch.cuda.Stream() for _ in range(2)]
for name, fn in CASES.items():
    res = {}

    def work(i, n, fn=fn):
        with torch.cuda.stream(streams[i]):
            outs = [fn(i).clone() for _ in range(n)]
            streams[i].synchronize()
        res[i] = outs
    for i in range(2):                                  # sequential reference
        t = threading.Thread(target=work, args=(i, 2))
        t.start()
        t.join()
    torch.cuda.synchronize()
    ref = {i: res[i][0].clone() for i in range(2)}
    seq_ok = all(torch.equal(o, ref[i]) for i in range(2) for o in res[i])
    th = [threading.Thread(target=work, args=(i, 12)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    bad = sum(int(not torch.equal(o, ref[i])) for i in range(2) for o in res[i])
    print(f"RESULT {name}: sequential repeat identical {seq_ok}; concurrent: {bad} of 24 results differ", flush=True)
print("splitk_error", ops.splitk_error())
