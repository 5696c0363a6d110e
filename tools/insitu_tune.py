#!/usr/bin/env python
"""Round 6, third session: tune the SDXL U-Net's table entries WHERE THEY RUN.

`da_gemm_tune` times a launch alone (cold weights, re-warmed activations); rounds 4-6 found several shapes whose isolated winner loses
inside the denoising step (FF-down, the fused Q|K|V projection, the re-tune check of profiles/r06j_sdxl_retune_check.txt: the re-tuned
table is 1 % slower in situ).  This tool makes the step itself the arbiter: one SDXL U-Net (bench configuration: batch 2, 128 x 128
latents), the denoising step captured into a HIP graph exactly as the pipeline does (`_denoise(use_graph=True)`), HIP events around
N replays.  For each table key of the step, by share of the step, every other (tile, staging) is put into the live table, the step is
re-captured and timed; a candidate is kept when it beats the incumbent by more than `margin`.  A candidate the library refuses for
the key's launches (`DA_ERR_UNSUPPORTED`: `ops._launch_gemm` would silently fall back to TILE_AUTO) is skipped, not timed.

usage: insitu_tune.py <out.json> [max_keys] [replays] [margin_percent] [cap_seconds] [earlier_pass.json | -] [sdxl | sd15 | ddpm | flux]
Writes {"changed": {key: [old, new, ms_before, ms_after]}, "baseline_ms": ..., "final_ms": ..., "log": [...]} and, next to it,
`table_insitu.json` (the shipped table with the winners) for a bench.py A/B via DIFFUSERS_AMD_TUNE_DB."""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


class Unsupported(RuntimeError):
    pass


def main():
    out = Path(sys.argv[1])
    max_keys = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    replays = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    margin = float(sys.argv[4]) / 100 if len(sys.argv) > 4 else 0.0015
    cap_s = float(sys.argv[5]) if len(sys.argv) > 5 else 1200.0        # stop exploring after this many seconds
    skip = set(json.loads(Path(sys.argv[6]).read_text())["keys"]) if len(sys.argv) > 6 and sys.argv[6] != "-" else set()   # keys an earlier pass explored
    config = sys.argv[7] if len(sys.argv) > 7 else "sdxl"
    import bench
    from diffusers_amd import _lib as L, ops, tuning
    dev = torch.device("cuda", 0)
    table = tuning.table()

    # ---- launches per key in one eager step, and a strict launcher for the key under test ----
    counts, state = {}, {"key": None, "count": True}
    lib = L.load()

    def launch(p, st, what):
        k = tuning.key_of(p)
        if state["count"]:
            # every launch counts, pinned ones too: the LayerNorm-fold producers / consumers and the fused Q|K|V projection read the table
            # themselves (ops.linear, ops.qkv_variant) and hand the variant over as a pinned one
            kk = "qkv:" + k if p.vt else k
            counts[kk] = counts.get(kk, 0) + 1
        rc = lib.da_gemm_bf16(C.byref(p), st)
        if rc == ops.DA_ERR_UNSUPPORTED and state["key"] in (k, "qkv:" + k):
            raise Unsupported(k)
        if rc == ops.DA_ERR_UNSUPPORTED and getattr(p, "_auto", False) and p.tile != L.TILE_AUTO:
            p.tile, p.staging, p.split_k = L.TILE_AUTO, L.STAGE_LDS_DIRECT, 1
            rc = lib.da_gemm_bf16(C.byref(p), st)
        L.check(rc, what)
    ops._launch_gemm = launch

    if config == "sdxl":
        from diffusers_amd import factory, init as dinit
        from diffusers_amd.pipelines import StableDiffusionXLPipeline
        from diffusers_amd.schedulers import EulerDiscreteScheduler
        unet, _ = factory.build_unet(dinit.SDXL_UNET, seed=0, device=dev, init_device=str(dev))
        pipe = StableDiffusionXLPipeline(vae=None, unet=unet, scheduler=EulerDiscreteScheduler(**factory.SDXL_SCHEDULER))
        inp = bench.synth_inputs(1, False, dev)
        pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], dim=0).contiguous()
        te = torch.cat([inp["negative_pooled"], inp["pooled"]], dim=0)
        ids = torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]], device=dev).repeat(2, 1)
        cond = unet.precompute_conditioning(pe, {"text_embeds": te, "time_ids": ids})
        pipe.scheduler.set_timesteps(50, device=dev)
        lat0 = inp["latents"].clone()
        lat = lat0.clone()
        pipe.scheduler.reset(0)
        pipe._step(lat, cond, bench.GUIDANCE, True)
        torch.cuda.synchronize()
        state["count"] = False

        def step_ms():
            """Re-capture the step with the live table, then time `replays` graph replays (two groups, the faster one)."""
            pipe._graph = None
            lat.copy_(lat0)
            pipe._denoise(lat, cond, 1, bench.GUIDANCE, True, True)       # warm-up eager step + capture + one replay
            best = 1e9
            for _ in range(2):
                lat.copy_(lat0)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                pipe.scheduler.reset(0)
                e0.record()
                for _ in range(replays):
                    pipe._graph.replay()
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / replays)
            return best
    else:
        # the other bench configurations (sd15 / ddpm / flux): the unit bench.py times (the engine pipeline's own call: `replays` sampler steps
        # + decode), the graph thrown away and captured again for every candidate; ms per sampler step, decode included
        pipe, unit = bench._build_other(config, dev, 0, False, False)
        unit(1, graph=False)()
        torch.cuda.synchronize()
        state["count"] = False

        def step_ms():
            pipe._graph, pipe._graph_key = None, None
            run = unit(replays)
            run()                                                         # warm-up eager step + capture + the unit once
            best = 1e9
            for _ in range(2):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run()
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / replays)
            return best

    base = [step_ms() for _ in range(3)]
    noise = (max(base) - min(base)) / min(base)
    cur = min(base)
    log = [{"baseline_ms": base, "noise": noise}]
    print(f"[insitu] baseline step {base} ms (spread {100 * noise:.2f} %), margin {100 * margin:.2f} %", flush=True)

    if config != "sdxl":      # the headline's entries are not this pass's to change
        skip |= set(json.loads((ROOT / "tests" / "golden" / "sdxl_gemm_shape_keys.json").read_text())["keys"])
    keys = [k for k in counts if k in table and k not in skip and not L.TILE_NAMES[table[k][0]].startswith("k3:") and table[k][3] <= 1]
    keys.sort(key=lambda k: -counts[k] * table[k][2])
    keys = keys[:max_keys]
    for k in keys:
        print(f"[insitu] key {k}: {counts[k]} launches x {table[k][2]:.1f} us = {counts[k] * table[k][2] / 1e3:.2f} ms", flush=True)
    stagings = (L.STAGE_LDS_DIRECT, L.STAGE_LDS_DIRECT3, L.STAGE_LDS_DIRECT4, L.STAGE_LDS_DIRECT6, L.STAGE_PINGPONG, L.STAGE_PINGPONG3)
    changed, shipped = {}, dict(table)
    t_start = time.time()
    for k in keys:
        if time.time() - t_start > cap_s:
            print(f"[insitu] time cap reached before {k}", flush=True)
            break
        old = table[k]
        best_ent, best_ms = old, cur
        for tile in range(1, len(L.TILE_NAMES)):
            if L.TILE_NAMES[tile].startswith("k3:"):
                continue
            for st in stagings:
                if (tile, st) == (old[0], old[1]):
                    continue
                table[k] = (tile, st, old[2], 1)
                state["key"] = k
                try:
                    ms = step_ms()
                except Unsupported:
                    continue
                except RuntimeError as e:
                    print(f"[insitu] {k} {L.TILE_NAMES[tile]}/st{st}: {str(e)[:80]}", flush=True)
                    continue
                finally:
                    state["key"] = None
                log.append({"key": k, "tile": L.TILE_NAMES[tile], "staging": st, "ms": ms})
                if ms < best_ms * (1 - margin):
                    # confirm against the incumbent back to back before adopting
                    table[k] = best_ent
                    ref = step_ms()
                    table[k] = (tile, st, old[2], 1)
                    again = step_ms()
                    print(f"[insitu] {k}: {L.TILE_NAMES[tile]}/st{st} {ms:.4f} / {again:.4f} ms vs incumbent {ref:.4f}", flush=True)
                    if max(ms, again) < ref * (1 - margin):
                        best_ent, best_ms = (tile, st, old[2], 1), max(ms, again)
        table[k] = best_ent
        if best_ent != old:
            changed[k] = [[L.TILE_NAMES[old[0]], old[1]], [L.TILE_NAMES[best_ent[0]], best_ent[1]], cur, best_ms]
            cur = best_ms
        print(f"[insitu] {k}: {'-> ' + L.TILE_NAMES[best_ent[0]] + '/st' + str(best_ent[1]) if best_ent != old else 'unchanged'}"
              f" (step {cur:.4f} ms, {time.time() - t_start:.0f} s)", flush=True)

    # ---- shipped vs final, alternating ----
    final = dict(table)
    ab = []
    for _ in range(3):
        table.clear(); table.update(shipped); a = step_ms()
        table.clear(); table.update(final); b = step_ms()
        ab.append([a, b])
    print(f"[insitu] shipped vs final (ms per step): {ab}", flush=True)
    raw = json.loads(Path(tuning.DB_PATH).read_text())                  # (DIFFUSERS_AMD_TUNE_DB chains passes: the table this pass started from)
    for k in changed:
        e = final[k]
        raw["entries"][k] = [e[0], e[1], round(e[2], 2)]
    (out.parent / ("table_insitu.json" if config == "sdxl" else f"table_insitu_{config}.json")).write_text(json.dumps(raw))
    out.write_text(json.dumps({"changed": changed, "baseline_ms": base, "shipped_vs_final_ms": ab, "keys": keys, "log": log}, indent=1))
    print(f"[insitu] changed {len(changed)} entries: {json.dumps(changed)}", flush=True)


if __name__ == "__main__":
    main()
