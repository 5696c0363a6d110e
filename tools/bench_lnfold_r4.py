#!/usr/bin/env python
"""LayerNorm fold on the second kernel family, round 4: the launch chains of one SDXL BasicTransformerBlock half, with the norm as a
kernel and folded, timed as CHAINS (back-to-back launches on one stream, HIP events: what the step pays), at both transformer
levels.  Also each launch on its own (cold: caches evicted before every launch).  One JSON object per line (argv[1]: appended)."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import _lib as L, ops  # noqa: E402
from tools.ceiling_table import chain_us, rnd, timeit  # noqa: E402
import tools.ceiling_table as CT  # noqa: E402


def main():
    out = open(sys.argv[1], "a") if len(sys.argv) > 1 else None

    def emit(rec):
        line = json.dumps(rec)
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()

    CT.FLUSH = torch.empty(320 << 20, dtype=torch.uint8, device="cuda")
    for level, M, C in (("1280 (32x32 tokens, B2)", 2048, 1280), ("640 (64x64 tokens, B2)", 8192, 640)):
        a, res = rnd((M, C)), rnd((M, C))
        wo, bo = rnd((C, C), C ** -0.5), rnd((C,))
        gamma, beta = rnd((C,)) * 0.2 + 1, rnd((C,)) * 0.1
        wq = rnd((C, C), C ** -0.5)
        w1, b1 = rnd((8 * C, C), C ** -0.5), rnd((8 * C,))
        wql, foldq = ops.fold_layernorm(wq, gamma, beta, 1e-5)
        w1l, fold1 = ops.fold_layernorm(w1, gamma, beta, 1e-5)
        w1p, b1p = ops.pack_geglu(w1, b1)
        w1lp, _ = ops.pack_geglu(w1l, None)
        n2 = w1.shape[0] // 2
        idx = torch.arange(n2, device="cuda").view(n2 // 32, 32)
        order = torch.cat([idx, idx + n2], dim=1).reshape(-1)
        fold1p = ops.LNFold(fold1.s[order].contiguous(), fold1.c[order].contiguous(), fold1.eps)
        st = ops.RowStats(M, "cuda")

        # single launches, cold
        t_out, _ = timeit(lambda: ops.linear(a, wo, bo, residual=res))
        t_outs, _ = timeit(lambda: ops.linear(a, wo, bo, residual=res, stats_out=st))
        h = ops.linear(a, wo, bo, residual=res, stats_out=st)
        t_ln, _ = timeit(lambda: ops.layer_norm(h, gamma, beta, 1e-5))
        hn = ops.layer_norm(h, gamma, beta, 1e-5)
        t_q, _ = timeit(lambda: ops.linear(hn, wq))
        t_qf, _ = timeit(lambda: ops.linear(h, wql, ln=(st, foldq)))
        t_g, _ = timeit(lambda: ops.linear(hn, w1p, b1p, act=L.ACT_GEGLU))
        t_gf, _ = timeit(lambda: ops.linear(h, w1lp, b1p, act=L.ACT_GEGLU, ln=(st, fold1p)))
        emit({"op": "lnfold single launches (cold)", "level": level, "to_out": round(t_out, 1), "to_out+stats": round(t_outs, 1),
              "layernorm": round(t_ln, 1), "to_q": round(t_q, 1), "to_q folded": round(t_qf, 1), "geglu": round(t_g, 1),
              "geglu folded": round(t_gf, 1), "parts": st.parts})
        # accuracy of the folded consumers against the kernel path
        yq, yqf = ops.linear(hn, wq), ops.linear(h, wql, ln=(st, foldq))
        yg, ygf = ops.linear(hn, w1p, b1p, act=L.ACT_GEGLU), ops.linear(h, w1lp, b1p, act=L.ACT_GEGLU, ln=(st, fold1p))
        rr = lambda x, y: float(((x.float() - y.float()).pow(2).mean() / y.float().pow(2).mean()).sqrt())  # noqa: E731
        emit({"op": "lnfold folded vs kernel path", "level": level, "to_q relrms": rr(yqf, yq), "geglu relrms": rr(ygf, yg)})

        # chains: what half a block pays
        def chain_q_plain():
            x = ops.linear(a, wo, bo, residual=res)
            return ops.linear(ops.layer_norm(x, gamma, beta, 1e-5), wq)

        def chain_q_fold():
            x = ops.linear(a, wo, bo, residual=res, stats_out=st)
            return ops.linear(x, wql, ln=(st, foldq))

        def chain_g_plain():
            x = ops.linear(a, wo, bo, residual=res)
            return ops.linear(ops.layer_norm(x, gamma, beta, 1e-5), w1p, b1p, act=L.ACT_GEGLU)

        def chain_g_fold():
            x = ops.linear(a, wo, bo, residual=res, stats_out=st)
            return ops.linear(x, w1lp, b1p, act=L.ACT_GEGLU, ln=(st, fold1p))

        rec = {"op": "lnfold chains (us per chain)", "level": level}
        for name, fn in (("to_out>LN>to_q", chain_q_plain), ("to_out+stats>to_q folded", chain_q_fold),
                         ("to_out>LN>geglu", chain_g_plain), ("to_out+stats>geglu folded", chain_g_fold)):
            rec[name] = round(min(chain_us(fn, 30) for _ in range(3)), 1)
        emit(rec)


if __name__ == "__main__":
    main()
