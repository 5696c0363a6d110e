#!/usr/bin/env python
"""Headline benchmark: images/sec @ SDXL-base 1024x1024, 50-step EulerDiscrete, CFG 5.0, bf16 (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...

A "step" is one complete image per GPU: 50 x (scale_model_input + CFG-batched U-Net + fused CFG/Euler step) + the VAE
decode, on synthetic prompt embeddings / latents that are resident in HBM before the timed region (weights are seeded
random, there are no checkpoints offline).  With N > 1 every rank owns one prompt (weak scaling, no collective in the
data path; rank 0 broadcasts the text embeddings once before the timed region).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

TFLOP_PER_IMAGE = 50 * 13.5225 + 10.4704   # 686.6 TFLOP per SDXL 1024^2 50-step image
UNET_TFLOP = 13.5225                        # one CFG-batched (B=2) U-Net forward at 128x128 latents
MFMA_PEAK_TFLOPS = 2500.0                   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed images per GPU")
    ap.add_argument("--warmup", type=int, default=1, help="untimed images per GPU")
    ap.add_argument("--denoise-steps", type=int, default=50)
    ap.add_argument("--tiny", action="store_true", help="tiny config (plumbing check only; not a valid bench number)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--save-tuning", default=None, help="write the GEMM variant table measured during warm-up here")
    return ap.parse_args()


def synth_inputs(n_prompts, tiny, device):
    """Fixed synthetic conditioning: per prompt seed 1234+i, N(0,1), bf16."""
    cd, pd, lat = (64, 64, 16) if tiny else (2048, 1280, 128)
    out = {k: [] for k in ("prompt_embeds", "negative_prompt_embeds", "pooled", "negative_pooled", "latents")}
    for i in range(n_prompts):
        g = torch.Generator("cpu").manual_seed(1234 + i)
        out["prompt_embeds"].append(torch.randn((77, cd), generator=g))
        out["negative_prompt_embeds"].append(torch.randn((77, cd), generator=g))
        out["pooled"].append(torch.randn((pd,), generator=g))
        out["negative_pooled"].append(torch.randn((pd,), generator=g))
        out["latents"].append(torch.randn((4, lat, lat), generator=g))
    return {k: torch.stack(v).to(torch.bfloat16).to(device) for k, v in out.items()}


def instrumented_gemm_pass(pipe, run_one_step):
    """Roofline leg: one eager denoising step with a HIP-event pair around every igemm launch (linear + conv), on the
    stream the kernels are launched on.  Returns (launches, total_ms, total_algorithmic_flops)."""
    from diffusers_amd import ops
    records = []
    orig_linear, orig_conv = ops.linear, ops.conv2d_nhwc

    def timed(fn, flops_of):
        def wrapper(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            records.append((e0, e1, flops_of(a, k, out)))
            return out
        return wrapper

    def lin_flops(a, k, out):
        x, w = a[0], a[1]
        if x.shape[0] <= 8:
            return None  # skinny path: not the igemm kernel
        return 2.0 * x.shape[0] * w.shape[0] * x.shape[1]

    def conv_flops(a, k, out):
        w = a[1]
        return 2.0 * out.shape[0] * out.shape[1] * out.shape[2] * w.shape[0] * w.shape[1]

    ops.linear, ops.conv2d_nhwc = timed(orig_linear, lin_flops), timed(orig_conv, conv_flops)
    try:
        run_one_step()
        torch.cuda.synchronize()
    finally:
        ops.linear, ops.conv2d_nhwc = orig_linear, orig_conv
    recs = [(e0.elapsed_time(e1), f) for e0, e1, f in records if f is not None]
    return len(recs), sum(r[0] for r in recs), sum(r[1] for r in recs)


# CFG-batched (B=2) SDXL U-Net forward at 32x32 latents, from the 128x128 op census of SURVEY.md 8a: conv 3.246/16,
# linear (8.709 - 0.105)/16 + 0.105 (the cross-attention K/V projections see 154 text rows at any resolution),
# attention 1.503/256 (self, quadratic in tokens) + 0.0646/16 (cross)
CPU_SAMPLE_TFLOP = 3.246 / 16 + (8.709 - 0.105) / 16 + 0.105 + 1.503 / 256 + 0.0646 / 16


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(unet_sd, cfg_full, budget_s=25.0):
    """CPU leg (rank 0, N=1): the oracle restatement of the U-Net forward (oracle/reference_math.py, kind "port") on the
    host cores this process may run on, on a bounded sample: CFG-batched forward of the FULL SDXL architecture at 32x32
    latents, fp32.  images/s is scaled by algorithmic FLOPs (CPU_SAMPLE_TFLOP per sample, 686.6 TFLOP per image)."""
    from oracle import reference_math as R
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    torch.set_num_threads(threads)
    sd = {k: v.detach().to("cpu", torch.float32) for k, v in unet_sd.items()}
    g = torch.Generator("cpu").manual_seed(7)
    sample = torch.randn((2, 4, 32, 32), generator=g)
    ehs = torch.randn((2, 77, 2048), generator=g)
    added = {"text_embeds": torch.randn((2, 1280), generator=g),
             "time_ids": torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]]).repeat(2, 1)}
    reps, tot = 0, 0.0
    with torch.no_grad():
        while True:
            t0 = time.perf_counter()
            R.unet_forward(sd, cfg_full, sample, 961.0, ehs, added)
            dt = time.perf_counter() - t0
            tot += dt
            reps += 1
            log(f"cpu_baseline: oracle forward {reps} took {dt:.1f} s on {threads} threads")
            if tot >= 10.0 or tot + dt > budget_s:
                break
    tflops = CPU_SAMPLE_TFLOP * reps / tot
    return {"value": tflops / TFLOP_PER_IMAGE, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{reps}x oracle fp32 CFG-batched SDXL U-Net forward at 32x32 latents ({CPU_SAMPLE_TFLOP:.3f} "
                      f"TFLOP each, {tot:.1f} s); scaled by FLOPs to 686.6 TFLOP/image",
            "cpu_tflops": tflops}


def main():
    args = parse()
    from diffusers_amd import distributed as D
    rank, world, local = D.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from diffusers_amd import factory, init as dinit
    from diffusers_amd.unet_2d_condition import _DEFAULTS as UD
    ucfg = dinit.TINY_SDXL_UNET if args.tiny else dinit.SDXL_UNET
    vcfg = dinit.TINY_VAE if args.tiny else dinit.SDXL_VAE
    log(f"rank {rank}/{world}: building models on {dev}")
    unet, unet_sd = factory.build_unet(ucfg, seed=0, device=dev, init_device=str(dev))
    vae, _ = factory.build_vae(vcfg, seed=1, device=dev, init_device=str(dev))
    from diffusers_amd.pipelines import StableDiffusionXLPipeline
    from diffusers_amd.schedulers import EulerDiscreteScheduler
    pipe = StableDiffusionXLPipeline(vae=vae, unet=unet, scheduler=EulerDiscreteScheduler(**factory.SDXL_SCHEDULER))

    # rank 0 owns the "prompts"; one broadcast of the embeddings over RCCL, then every rank keeps its shard
    inputs = synth_inputs(world, args.tiny, dev)
    if world > 1:
        if rank != 0:
            inputs = {k: torch.empty_like(v) for k, v in inputs.items()}
        D.broadcast_tensors(inputs, src=0)
    mine = D.select_shard(inputs, D.shard_indices(world, rank, world))
    hw = 128 if args.tiny else 1024

    state = {"graph": not args.no_graph}

    def one_image():
        return pipe(prompt_embeds=mine["prompt_embeds"], negative_prompt_embeds=mine["negative_prompt_embeds"],
                    pooled_prompt_embeds=mine["pooled"], negative_pooled_prompt_embeds=mine["negative_pooled"],
                    latents=mine["latents"].clone(), num_inference_steps=args.denoise_steps, guidance_scale=5.0,
                    height=hw, width=hw, output_type="raw", use_graph=state["graph"]).images

    log("warm-up (tunes GEMM variants for unseen shapes, captures the denoising-step HIP graph)")
    for _ in range(args.warmup):
        try:
            img = one_image()
        except RuntimeError as e:
            # a capture that another component of the process invalidated must not cost the measurement: the eager
            # launch path is the same kernels in the same order, and the JSON line says which one ran
            if not state["graph"]:
                raise
            log(f"HIP-graph capture failed ({e}); continuing with eager launches")
            torch.cuda.synchronize()
            state["graph"] = False
            pipe._graph = None
            img = one_image()
    torch.cuda.synchronize()
    log("timed region")
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img = one_image()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    finite = bool(torch.isfinite(img.float()).all())
    log(f"timed region done: {elapsed:.3f} s for {args.steps} image(s) per GPU")
    if args.save_tuning and rank == 0:
        from diffusers_amd import tuning
        tuning.save(args.save_tuning)

    result = {
        "metric": "images/sec @ SDXL-base 1024x1024 50-step EulerDiscrete CFG bf16",
        "value": world * args.steps / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic (seeded random weights, embeddings, latents)",
        "config": {"workload": "SDXL-base-1.0 U-Net (2567 M params) x 50 EulerDiscrete steps, CFG 5.0 (batch 2), "
                               "128x128 latents + AutoencoderKL decode to 1024x1024; 1 prompt per GPU"
                               if not args.tiny else "TINY plumbing config (not a benchmark)",
                   "global_batch": world, "parallelism": f"dp{world} (independent prompts, replicas)",
                   "denoise_steps": args.denoise_steps, "hip_graph": state["graph"], "output_finite": finite},
    }

    if rank == 0 and not args.tiny:
        if not args.no_roofline:
            # dominant kernel = igemm_bf16_kernel (all Linear + Conv2d 3x3/1x1): MFMA-bound.
            sch = pipe.scheduler
            cond = pipe._static.get("cond")
            lat = mine["latents"].clone()

            def one_step():
                sch.reset(0)
                pipe._step(lat, cond, 5.0, True)
            log("roofline leg: one eager denoising step with HIP events around every igemm launch")
            one_step()  # untimed warm pass
            n, ms, fl = instrumented_gemm_pass(pipe, one_step)
            ach = fl / (ms * 1e-3) / 1e12
            result["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": ach / MFMA_PEAK_TFLOPS, "traffic": None,
                                  "kernel": "igemm_bf16_kernel (Linear + Conv2d implicit GEMM)",
                                  "launches_per_denoise_step": n, "avg_launch_us": 1000.0 * ms / max(n, 1),
                                  "algorithmic_tflop_per_denoise_step": fl / 1e12,
                                  "end_to_end_frac": (world * args.steps / elapsed) * TFLOP_PER_IMAGE / world / MFMA_PEAK_TFLOPS}
        if world == 1 and not args.no_cpu_baseline:
            full = dict(UD)
            full.update(ucfg)
            log("cpu_baseline leg")
            import signal

            def _alarm(signum, frame):
                raise TimeoutError("cpu_baseline exceeded its wall-clock bound")
            signal.signal(signal.SIGALRM, _alarm)
            signal.alarm(150)
            try:
                result["cpu_baseline"] = cpu_baseline(unet_sd, full)
            except TimeoutError as e:
                result["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": None, "kind": "port",
                                          "sample": f"not measured: {e}"}
            finally:
                signal.alarm(0)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
