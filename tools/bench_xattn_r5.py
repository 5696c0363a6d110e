"""Round 5: the cross-attention link of a BasicTransformerBlock (attention.py:1030-1042) as launches -- to_q (norm2 folded) ->
77-key attention -> to_out (+ residual) -- against the fused form (to_q + attention in ONE launch, da_gemm_params.xa_*) -> to_out.
Chains of back-to-back launches in a HIP graph (as the step replays them), cold-ish operands (every layer its own weights).
usage: python tools/bench_xattn_r5.py out.jsonl"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from diffusers_amd import _lib as L, layers, ops  # noqa: E402

bf16 = torch.bfloat16
DEV = "cuda"


def rnd(shape, seed, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(bf16).to(DEV)


def graph_time(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3   # us per chain


def main(out):
    recs = []
    for (B, seq, C, heads, nlayers) in ((2, 1024, 1280, 20, 12), (2, 4096, 640, 10, 6)):
        inner, M = heads * 64, B * seq
        pad, s, sa = layers.pad_encoder_states(rnd((B, 77, 2048), 5))
        Ls = []
        for i in range(nlayers):
            wq, wk, wv = rnd((inner, C), 10 * i + 1, C ** -0.5), rnd((inner, 2048), 10 * i + 2, 2048 ** -0.5), rnd((inner, 2048), 10 * i + 3, 2048 ** -0.5)
            wo, bo = rnd((C, inner), 10 * i + 4, inner ** -0.5), rnd((C,), 10 * i + 5)
            gamma, beta = rnd((C,), 10 * i + 6) * 0.3 + 1.0, rnd((C,), 10 * i + 7) * 0.2
            wl, fold = ops.fold_layernorm(wq, gamma, beta, 1e-5)
            Ls.append(dict(wl=wl, fold=fold, k=ops.linear(pad, wk), vt=ops.linear(wv, pad), wo=wo, bo=bo))
        x = rnd((M, C), 99)
        st = ops.RowStats(M, DEV)
        x = ops.linear(rnd((M, 192), 98), rnd((C, 192), 97, 192 ** -0.5), residual=x, stats_out=st)
        scale = 64 ** -0.5

        def chain(fused, stg=None, with_out=True):
            def run():
                h = x
                for l_ in Ls:
                    if fused:
                        xa = {"k": l_["k"], "vt": l_["vt"], "skv": s, "skv_alloc": sa, "seq": seq, "scale": scale}
                        o = ops.linear(h, l_["wl"], ln=(st, l_["fold"]), xattn=xa, tile=L.TILE_K2_128x128, staging=stg)
                    else:
                        q = ops.linear(h, l_["wl"], ln=(st, l_["fold"]))
                        o = ops.attention(q, l_["k"], l_["vt"], B=B, H=heads, D=64, Sq=seq, Skv=s, Skv_alloc=sa, q_row_stride=inner,
                                          k_row_stride=inner, q_batch_stride=seq * inner, k_batch_stride=sa * inner, vt_ld=B * sa,
                                          vt_batch_stride=sa, scale=scale)
                    if with_out:
                        o = ops.linear(o, l_["wo"], l_["bo"], residual=h)
                return o
            return run
        for with_out in (False, True):
            t_sep = graph_time(chain(False, with_out=with_out)) / nlayers
            t_pp = graph_time(chain(True, L.STAGE_PINGPONG, with_out)) / nlayers
            t_ip = graph_time(chain(True, L.STAGE_LDS_DIRECT, with_out)) / nlayers
            rec = {"op": "cross-attention link" + (" + to_out" if with_out else ""), "M": M, "C": C, "heads": heads, "keys": 77,
                   "layers_in_chain": nlayers, "us_per_layer_separate_launches": round(t_sep, 2),
                   "us_per_layer_fused_pingpong": round(t_pp, 2), "us_per_layer_fused_inphase": round(t_ip, 2)}
            print(json.dumps(rec), flush=True)
            recs.append(rec)
    Path(out).write_text("".join(json.dumps(r) + "\n" for r in recs))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else str(ROOT / "gpurun_out" / "xattn_r5.jsonl"))
