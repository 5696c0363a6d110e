#!/usr/bin/env python
"""Headline benchmark: images/sec @ SDXL-base 1024x1024, 50-step EulerDiscrete, CFG 5.0, bf16 (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...

A "step" is one complete image per GPU: 50 x (scale_model_input + CFG-batched U-Net + fused CFG/Euler step) + the VAE
decode, on synthetic prompt embeddings / latents that are resident in HBM before the timed region (weights are seeded
random, there are no checkpoints offline).  With N > 1 every rank owns one prompt (weak scaling, no collective in the
data path; rank 0 broadcasts the text embeddings once before the timed region).  ``python bench.py --gpus N`` without a
launcher re-executes itself under ``torch.distributed.run`` with N ranks (the reference's own recipe spawns its ranks the
same way, docs/source/en/training/distributed_inference.md:62-108).  Rank 0 prints ONE JSON line.

Besides the contract keys the line carries, all measured OUTSIDE the timed region on rank 0 at N = 1:
  roofline            dominant kernel family (igemm_bf16_kernel + igemm2_bf16_kernel + gemm3_bf16_kernel) over the launches of one eager denoising step,
                      timed with HIP events ATTACHED to every launch (da_set_launch_events: the dispatch's own begin / end, the
                      duration rocprofv3 reports; `timing` says which method ran, `host_event_pairs` carries the host-recorded
                      pairs of the same launches -- the method of rounds 1-3 -- as a cross-check); `kernels`: one entry per family
  parity              PSNR of the engine's 50-step image against the REFERENCE PACKAGE itself (StableDiffusionXLPipeline over
                      the reference's own classes, from the oracle/_ref archive) run by PyTorch-ROCm in fp32 on this GPU on
                      the same weights / latents / embeddings, with the bf16 run of the same pipeline as noise floor
  torch_rocm_baseline the same reference pipeline, eager bf16 on PyTorch-ROCm (hipBLASLt / MIOpen / SDPA): the denominator
                      of BASELINE.json's ">= 1.5x stock diffusers on PyTorch-ROCm"   ("kind": "reference"; "port" = the
                      oracle restatement, used only when the archive did not ship)
  cpu_baseline        the reference classes on the host cores: one full-size step + one decode (image = 50 x step + decode)
  dropin              the reference StableDiffusionXLPipeline.__call__ itself over ENGINE unet / vae / scheduler components, full
                      size: images/s, ratio to the graphed engine pipeline, PSNR vs the all-reference fp32 run
  other_configs       BASELINE configs 2, 4, 1, 5 (sd15, flux, ddpm, wan): three timed units each under the same timing rule
                      (Wan: 3 sampler steps scaled to 50, flagged), each with per-family roofline.kernels and a cpu_baseline
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

TFLOP_PER_IMAGE = 50 * 13.5225 + 10.4704   # 686.6 TFLOP per SDXL 1024^2 50-step image
UNET_TFLOP = 13.5225                        # one CFG-batched (B=2) U-Net forward at 128x128 latents
MFMA_PEAK_TFLOPS = 2500.0                   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
GUIDANCE = 5.0
TRAFFIC_FILE = ROOT / "profiles" / "sdxl_traffic.json"   # written by tools/pmc_traffic.py from rocprofv3 --pmc passes


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed images per GPU")
    ap.add_argument("--warmup", type=int, default=1, help="untimed images per GPU")
    ap.add_argument("--config", default="sdxl", choices=["sdxl", "sd15", "flux", "wan", "ddpm"],
                    help="BASELINE.json config: sdxl (3, the headline metric, default), sd15 (2), flux (4), wan (5), ddpm (1)")
    ap.add_argument("--denoise-steps", type=int, default=None, help="sampler steps per unit (default: the config's own: 50, flux 4)")
    ap.add_argument("--tiny", action="store_true", help="tiny config (plumbing check only; not a valid bench number)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic: do not run the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) in subprocesses; replay "
                         "profiles/sdxl_traffic.json instead (what a multi-rank run and a box without rocprofv3 do anyway)")
    ap.add_argument("--no-reference", action="store_true", help="skip the parity / torch_rocm_baseline / dropin legs")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other_configs leg (sd15, flux, ddpm, wan) of the default run")
    ap.add_argument("--save-tuning", default=None, help="write the GEMM variant table measured during warm-up here")
    return ap.parse_args(argv)


# ---- device plumbing (one place, so the world-size-2 CPU test can run main() on the kernel stand-ins) ----------------
def _device(local: int) -> torch.device:
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    torch.cuda.set_device(local)
    return torch.device("cuda", local)


def _sync() -> None:
    torch.cuda.synchronize()


def _tuned_live() -> int:
    """GEMM / conv shapes this rank had to tune live during warm-up (0 when the shipped table covers the workload)."""
    from diffusers_amd import tuning
    return int(tuning.LIVE_COUNT)


def box_mfma_probe(target_ms: float = 50.0):
    """`config.box_mfma_tflops`: the register-only MFMA loop of the library (da_mfma_probe, csrc/version.hip) for ~50 ms BEFORE the
    timed region -- matrix-pipe issue rate x the clock this box sustains under matrix load.  A normaliser for comparing lines
    measured on different boxes of the pool (+- 8 % on one build), never a peak: fractions stay against 2.5 PFLOP/s."""
    import ctypes
    from diffusers_amd import _lib as L
    lib = L.load()
    sink = torch.zeros(4, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    flop = ctypes.c_double(0.0)
    blocks, iters = 2048, 2000

    def run(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.da_mfma_probe(blocks, n, sink.data_ptr(), ctypes.byref(flop), st), "da_mfma_probe")
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1), flop.value
    run(200)                                     # clocks up, code object loaded
    ms, fl = run(iters)
    iters = max(200, int(iters * target_ms / max(ms, 1e-3)))
    ms, fl = run(iters)
    return {"tflops": fl / (ms * 1e-3) / 1e12, "ms": ms, "what": f"da_mfma_probe: {blocks} workgroups x 4 waves x {iters} x 16 "
            "v_mfma_f32_32x32x16_bf16, registers only"}


def box_memory_probe():
    """`config.box_hbm_copy_gbps` / `config.box_cold_gemm_us` (round 5): the pool's boxes come in two kinds with the SAME register-only
    MFMA rate (2450-2466 TFLOP/s) whose SDXL lines differ by 15 % (profiles/r05j_*, r05m_*: every weight-streaming GEMM 12-28 % slower,
    flash attention unchanged) -- the difference is on the memory side, which the MFMA probe cannot see.  Two more normalisers, never
    peaks: (1) a 1 GiB device copy (bytes read + written per second); (2) SDXL's 2048 x 1280 x 1280 projection with its weight rotated
    over 128 different tensors (420 MB: every launch streams its weight from HBM, as inside a denoising step), mean us per launch."""
    from diffusers_amd import ops
    bf16 = torch.bfloat16
    out = {}
    src = torch.empty(1 << 29, dtype=bf16, device="cuda")
    src.fill_(1.0)
    dst = torch.empty_like(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst.copy_(src)
    e0.record()
    for _ in range(4):
        dst.copy_(src)
    e1.record()
    e1.synchronize()
    out["hbm_copy_gbps"] = 4 * 2 * src.numel() * 2 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    x = torch.randn(2048, 1280, device="cuda").to(bf16)
    ws = (torch.randn(128, 1280, 1280, device="cuda") * 1280 ** -0.5).to(bf16)
    for i in range(8):
        ops.linear(x, ws[i])
    e0.record()
    for i in range(128):
        ops.linear(x, ws[i])
    e1.record()
    e1.synchronize()
    out["cold_gemm_us"] = e0.elapsed_time(e1) * 1e3 / 128
    out["what"] = ("1 GiB torch copy (read + written bytes / s); da_gemm_bf16 2048 x 1280 x 1280 with 128 rotating weights (420 MB), "
                   "mean per launch, back to back")
    return out


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` (N > 1) without a launcher: one process per GPU under torch.distributed.run, rendezvous
    on 127.0.0.1 (the container hostname may not resolve).  The children print the JSON line; this process only waits."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    log(f"--gpus {args.gpus} without a launcher: re-executing as {' '.join(cmd[1:8])} ...")
    return subprocess.call(cmd, env=env)


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


HBM_PEAK_GBS = 8000.0                       # MI355X HBM3E peak (MI355X_MICROARCH.md; ~6300 GB/s achievable)


def instrumented_pass(run, attach=True):
    """Roofline legs: run() eagerly with a HIP-event pair for every launch of the kernel families that carry the path, on
    the stream the kernels are launched on.  ``attach`` (default): the pair is ATTACHED to the launch (da_set_launch_events ->
    hipExtLaunchKernelGGL stamps the events with the dispatch's own begin / end: the kernel's duration as rocprofv3 reports it);
    ``attach=False``: the pair is recorded from the host either side of the call, which also counts the marker packets' own
    processing (+ 1-3 us per launch: it weighs most on the 10 us launches).  Returns {family: [launches, ms, algorithmic flop,
    algorithmic bytes]}:
      igemm      every Linear / Conv2d launch INCLUDING the paired Q|K + V^T launches (ops.linear_pair)
      attention  flash attention forward (4 B H Sq Skv D flop)
      groupnorm  GroupNorm (+ SiLU): read for the statistics, read + write for the apply pass = 6 B per element
      layernorm  LayerNorm: read + write = 4 B per element"""
    from diffusers_amd import ops
    records = []
    names = ("linear", "linear_pair", "conv2d_nhwc", "attention", "group_norm_nhwc", "layer_norm")
    orig = {n: getattr(ops, n) for n in names}

    from diffusers_amd import _lib as L
    lib = L.load()

    def timed(fn, work_of, family):
        def wrapper(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if attach:
                e1.record()                          # (both handles exist now; the launch below records them again)
                L.check(lib.da_set_launch_events(e0.cuda_event, e1.cuda_event), "da_set_launch_events")
                try:
                    out = fn(*a, **k)
                finally:
                    lib.da_set_launch_events(None, None)
            else:
                out = fn(*a, **k)
                e1.record()
            records.append((family, e0, e1, work_of(a, k, out)))
            return out
        return wrapper

    def lin_work(a, k, out):
        x = a[0] if a else k["x"]
        w = a[1] if len(a) > 1 else k["w"]
        if x.shape[0] <= 8:
            return None  # skinny path: not the igemm kernel
        # (flops, algorithmic bytes = each operand and the output once)
        return 2.0 * x.shape[0] * w.shape[0] * x.shape[1], 2.0 * (x.numel() + w.numel() + out.numel())

    def pair_work(a, k, out):
        if any(prob["x"].shape[0] <= 8 for prob in a):
            return None  # linear_pair falls back to two skinny linear() calls: neither the igemm kernel nor one launch
        tot_f = tot_b = 0.0
        for prob, o in zip(a, out):
            x, w = prob["x"], prob["w"]
            tot_f += 2.0 * x.shape[0] * w.shape[0] * x.shape[1]
            tot_b += 2.0 * (x.numel() + w.numel() + o.numel())
        return tot_f, tot_b

    def conv_work(a, k, out):
        x, w = a[0], a[1]
        x2 = k.get("x2")
        return (2.0 * out.shape[0] * out.shape[1] * out.shape[2] * w.shape[0] * w.shape[1],
                2.0 * (x.numel() + (x2.numel() if x2 is not None else 0) + w.numel() + out.numel()))

    def attn_work(a, k, out):
        q, kk, vt = a[0], a[1], a[2]
        B, H, D, Sq, Skv = k["B"], k["H"], k["D"], k["Sq"], k["Skv"]
        return 4.0 * B * H * Sq * Skv * D, 2.0 * (2 * B * H * Sq * D + 2 * B * H * Skv * D)

    def gn_work(a, k, out):
        x2 = k.get("x2")
        n = out.numel()
        return 0.0, 6.0 * n

    def ln_work(a, k, out):
        return 0.0, 4.0 * out.numel()

    ops.linear = timed(orig["linear"], lin_work, "igemm")
    ops.linear_pair = timed(orig["linear_pair"], pair_work, "igemm")
    ops.conv2d_nhwc = timed(orig["conv2d_nhwc"], conv_work, "igemm")
    ops.attention = timed(orig["attention"], attn_work, "attention")
    ops.group_norm_nhwc = timed(orig["group_norm_nhwc"], gn_work, "groupnorm")
    ops.layer_norm = timed(orig["layer_norm"], ln_work, "layernorm")
    try:
        run()
        torch.cuda.synchronize()
    finally:
        for n in names:
            setattr(ops, n, orig[n])
    fam = {}
    for family, e0, e1, work in records:
        if work is None:
            continue
        f = fam.setdefault(family, [0, 0.0, 0.0, 0.0])
        f[0] += 1
        f[1] += e0.elapsed_time(e1)
        f[2] += work[0]
        f[3] += work[1]
    return fam


TIMING_ATTACHED = ("HIP events attached to the launches (da_set_launch_events -> hipExtLaunchKernelGGL: begin / end of the dispatch, "
                   "the duration rocprofv3 reports)")
TIMING_HOST = "HIP event pair recorded from the host either side of every launch (includes the marker packets' processing)"


def measured_families(run):
    """(families, timing note, families by the host-recorded pairs or None): the attached-event pass, checked -- every interval
    positive and the total not above the host-recorded total, which brackets the same kernels from further out -- else the host pass
    alone (a runtime without hipExtLaunchKernelGGL support must not cost the line)."""
    host = instrumented_pass(run, attach=False)
    try:
        att = instrumented_pass(run, attach=True)
        ok = all(k in att and att[k][0] == host[k][0] and 0.0 < att[k][1] <= host[k][1] * 1.05 for k in host)
    except Exception as e:
        log(f"attached timing events unavailable ({type(e).__name__}: {e}); host-recorded pairs")
        ok = False
    if ok:
        return att, TIMING_ATTACHED, host
    return host, TIMING_HOST, None


def _kernel_entry(name, kernel, bound, rec, extra=None):
    n, ms, fl, nb = rec
    e = {"name": name, "kernel": kernel, "bound": bound, "launches": n, "ms": ms, "avg_launch_us": 1000.0 * ms / max(n, 1)}
    if bound == "mfma":
        e.update({"achieved": fl / (ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s"})
    else:
        e.update({"achieved": nb / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"})
    e["frac"] = e["achieved"] / e["peak"]
    e["algorithmic_gflop"], e["algorithmic_mbytes"] = fl / 1e9, nb / 1e6
    if extra:
        e.update(extra)
    return e


def build_fingerprint() -> str:
    """Identity of the kernels a measurement was taken on (the stamp diffusers_amd/build.py writes next to the library)."""
    stamp = ROOT / "diffusers_amd" / "_C" / "build.stamp"
    return stamp.read_text().strip()[:16] if stamp.exists() else "unknown"


def live_traffic(algo_bytes_per_launch: float, timeout_s: int = 240):
    """roofline.traffic MEASURED IN THIS RUN: two rocprofv3 passes (FETCH_SIZE, then WRITE_SIZE -- separate --pmc passes with
    --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes) over tools/pmc_one_step.py in SUBPROCESSES (one eager SDXL
    denoising step at the bench configuration: same library, same tuned table, same launches as the timed graph; this process cannot
    be profiled from inside), reduced by tools/pmc_traffic.py (gfx950's 2 x FETCH_SIZE correction, the 140 hoisted K / V^T launches
    set aside).  Returns (record, note) or (None, why not); never raises -- the caller then replays the committed measurement."""
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if Path("/opt/rocm/bin/rocprofv3").exists() else None)
    if rp is None:
        return None, "rocprofv3 not found"
    tmp = Path(tempfile.mkdtemp(prefix="da_pmc_", dir="/tmp"))
    env = dict(os.environ, DIFFUSERS_AMD_TUNE="0", TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        for name, counter, extra in (("fetch", "FETCH_SIZE", [str(tmp / "launch_log.json")]), ("write", "WRITE_SIZE", [])):
            cmd = [rp, "--kernel-trace", "--pmc", counter, "-f", "csv", "-d", str(tmp / name), "-o", "sdxl", "--",
                   sys.executable, str(ROOT / "tools" / "pmc_one_step.py"), "1", *extra]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0 or "pmc_one_step: done" not in r.stdout + r.stderr:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]!r}"
        r = subprocess.run([sys.executable, str(ROOT / "tools" / "pmc_traffic.py"), str(tmp / "fetch"), str(tmp / "write"),
                            str(tmp / "traffic.md"), str(tmp / "traffic.json"), repr(float(algo_bytes_per_launch)), "140",
                            str(tmp / "launch_log.json"), "1"], capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            return None, f"tools/pmc_traffic.py failed: {r.stderr[-200:]!r}"
        tr = json.loads((tmp / "traffic.json").read_text())
        if "igemm_bytes_per_launch" not in tr:
            return None, "no implicit-GEMM rows in the counter files"
        return tr, ("MEASURED IN THIS RUN: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two separate passes, 2x FETCH_SIZE "
                    "correction for gfx950's wide streams, WRITE_SIZE as reported) over one eager SDXL denoising step of this library in a "
                    "subprocess (tools/pmc_one_step.py; same tuned table and launches as the timed graph), bytes / implicit-GEMM launches "
                    "of the step (tools/pmc_traffic.py; the 140 hoisted K / V^T projections set aside)")
    except Exception as e:  # a diagnostic leg never costs the line
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def roofline_leg(pipe, mine, world, images_per_s, live=True):
    """Dominant kernel = the implicit-GEMM family (igemm_bf16_kernel + igemm2_bf16_kernel + gemm3_bf16_kernel: all Linear + Conv2d 3x3 / 1x1, paired
    launches included): MFMA-bound.  `kernels` carries the other families of the denoising step and the VAE decode.  The
    conditioning is built here (not taken from the graph's static inputs), so the leg also works after --no-graph."""
    sch = pipe.scheduler
    dev = mine["latents"].device
    pe = torch.cat([mine["negative_prompt_embeds"], mine["prompt_embeds"]], dim=0).contiguous()
    te = torch.cat([mine["negative_pooled"], mine["pooled"]], dim=0)
    ids = torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]], device=dev).repeat(pe.shape[0], 1)
    cond = pipe.unet.precompute_conditioning(pe, {"text_embeds": te, "time_ids": ids})
    sch.set_timesteps(50, device=dev)
    lat = mine["latents"].clone()

    from diffusers_amd import ops as _ops
    pf = getattr(pipe, "_weight_prefetch", None)      # the trace the timed runs recorded: the leg measures the same launches

    def one_step():
        sch.reset(0)
        with _ops.weight_prefetch(pf, "apply"):
            pipe._step(lat, cond, GUIDANCE, True)
    one_step()  # untimed warm pass
    fam, timing, fam_host = measured_families(one_step)
    n, ms, fl, nbytes = fam["igemm"]
    ach = fl / (ms * 1e-3) / 1e12
    fp = build_fingerprint()
    traffic, note, family_ratios = None, "no committed PMC measurement (profiles/sdxl_traffic.json)", None
    if live and world == 1:
        log("roofline leg: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over one eager step (subprocesses)")
        t0 = time.perf_counter()
        tr, why = live_traffic(nbytes / max(n, 1))
        if tr is not None:
            traffic, note = tr["igemm_bytes_per_launch"], why
            family_ratios = {k: round(v["ratio"], 3) for k, v in tr.get("igemm_family_ratios", {}).items()}
            log(f"roofline leg: traffic {traffic / 1e6:.1f} MB per implicit-GEMM launch, measured in {time.perf_counter() - t0:.0f} s")
        else:
            log(f"roofline leg: live traffic measurement unavailable ({why}); replaying {TRAFFIC_FILE.name}")
            note = f"live measurement unavailable ({why}); "
    if traffic is None and TRAFFIC_FILE.exists():
        live_note, note = (note if note.startswith("live measurement") else ""), ""
        try:
            tr = json.loads(TRAFFIC_FILE.read_text())
            if tr.get("build_fingerprint") == fp:
                traffic = tr["igemm_bytes_per_launch"]
                note = (f"NOT measured in this run: replayed from {TRAFFIC_FILE.name}, measured on build {fp} (the library this run "
                        f"loaded) on {tr.get('measured_on', 'an earlier date (unstamped)')} by " + tr.get("source", "rocprofv3 --pmc"))
            else:
                note = (f"stale: {TRAFFIC_FILE.name} was measured on build {tr.get('build_fingerprint', 'unstamped (round 2)')}, "
                        f"this library is {fp}; re-run tools/gpu_r4.sh traffic")
        except (ValueError, KeyError) as e:
            note = f"unreadable {TRAFFIC_FILE.name}: {e}"
        note = live_note + note
    kernels = [_kernel_entry("igemm", "igemm_bf16_kernel + igemm2_bf16_kernel + gemm3_bf16_kernel (Linear, Conv2d, paired Q|K + V^T)", "mfma", fam["igemm"])]
    if "attention" in fam:
        kernels.append(_kernel_entry("attention", "attn_fwd_kernel (flash attention forward)", "mfma", fam["attention"]))
    if "groupnorm" in fam:
        kernels.append(_kernel_entry("groupnorm", "gn_stats_kernel + gn_apply_kernel (GroupNorm + SiLU)", "hbm", fam["groupnorm"]))
    if "layernorm" in fam:
        kernels.append(_kernel_entry("layernorm", "layernorm_kernel", "hbm", fam["layernorm"]))
    # the VAE decode (once per image): wall time with events, its own implicit-GEMM share, algorithmic traffic of its norms
    try:
        z = (mine["latents"] / 0.13025).contiguous()
        pipe.vae.decode(z, return_dict=False)          # warm
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pipe.vae.decode(z, return_dict=False)
        e1.record()
        e1.synchronize()
        wall_ms = e0.elapsed_time(e1)
        vf = instrumented_pass(lambda: pipe.vae.decode(z, return_dict=False))
        vg = vf.get("igemm", [0, 0.0, 0.0, 0.0])
        vn = vf.get("groupnorm", [0, 0.0, 0.0, 0.0])
        kernels.append({"name": "vae_decode", "kernel": "AutoencoderKL.decode 128x128 -> 1024x1024 (whole call)", "bound": "mfma",
                        "ms": wall_ms, "achieved": 10.4704 / (wall_ms * 1e-3), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": 10.4704 / (wall_ms * 1e-3) / MFMA_PEAK_TFLOPS, "algorithmic_gflop": 10470.4,
                        "igemm_ms": vg[1], "igemm_tflops": vg[2] / max(vg[1], 1e-9) / 1e9,
                        "groupnorm_ms": vn[1], "groupnorm_gbs": vn[3] / max(vn[1], 1e-9) / 1e6,
                        "groupnorm_frac_of_hbm_peak": vn[3] / max(vn[1], 1e-9) / 1e6 / HBM_PEAK_GBS})
    except Exception as e:  # a diagnostic must not cost the line
        kernels.append({"name": "vae_decode", "error": f"{type(e).__name__}: {e}"})
    host_pair = None
    if fam_host is not None:       # the same launches by the host-recorded pairs (what rounds 1-3 reported)
        host_pair = {k: {"avg_launch_us": 1000.0 * v[1] / max(v[0], 1),
                         **({"achieved": v[2] / (v[1] * 1e-3) / 1e12, "frac": v[2] / (v[1] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS}
                            if k in ("igemm", "attention") else {})} for k, v in fam_host.items()}
    return {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / MFMA_PEAK_TFLOPS, "timing": timing, "host_event_pairs": host_pair,
            "traffic": traffic, "traffic_source": note, "traffic_ratio_by_kernel_family": family_ratios, "build_fingerprint": fp,
            "kernel": "igemm_bf16_kernel + igemm2_bf16_kernel + gemm3_bf16_kernel (Linear + Conv2d implicit GEMM, paired launches included)",
            "launches_per_denoise_step": n, "avg_launch_us": 1000.0 * ms / max(n, 1),
            "algorithmic_tflop_per_denoise_step": fl / 1e12,
            "algorithmic_bytes_per_launch": nbytes / max(n, 1),
            "end_to_end_frac": images_per_s * TFLOP_PER_IMAGE / world / MFMA_PEAK_TFLOPS,
            "kernels": kernels}


# CFG-batched (B=2) SDXL U-Net forward at 32x32 latents, from the 128x128 op census of SURVEY.md 8a: conv 3.246/16,
# linear (8.709 - 0.105)/16 + 0.105 (the cross-attention K/V projections see 154 text rows at any resolution),
# attention 1.503/256 (self, quadratic in tokens) + 0.0646/16 (cross)
CPU_SAMPLE_TFLOP = 3.246 / 16 + (8.709 - 0.105) / 16 + 0.105 + 1.503 / 256 + 0.0646 / 16
VAE_TFLOP = 10.4704


def _host_threads():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, 64))


def cpu_baseline_reference(ref, unet_sd, vae_sd, ucfg, vcfg, budget_s=200.0):
    """CPU leg, kind "reference": the REAL reference classes (oracle/_ref archive) in fp32 on the host cores, on the bounded
    sample SURVEY.md 8d plans: ONE full-size denoising step (CFG-batched `UNet2DConditionModel.forward` at 128x128 latents,
    13.52 TFLOP) and ONE `AutoencoderKL.decode` to 1024x1024 (10.47 TFLOP).  The 50 steps of an image are cost-identical, so
    seconds per image = 50 x step + decode (the only extrapolation).  The decode is skipped (and scaled by FLOPs from the step)
    if the step alone used most of the budget."""
    from oracle import ref_runtime as RR
    threads = _host_threads()
    torch.set_num_threads(threads)
    t_start = time.perf_counter()
    unet = RR.build_unet(ref, ucfg, unet_sd, "cpu", torch.float32)
    g = torch.Generator("cpu").manual_seed(7)
    sample = torch.randn((2, 4, 128, 128), generator=g)
    ehs = torch.randn((2, 77, 2048), generator=g)
    added = {"text_embeds": torch.randn((2, 1280), generator=g),
             "time_ids": torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]]).repeat(2, 1)}
    with torch.no_grad():
        t0 = time.perf_counter()
        unet(sample, torch.tensor(961.0), encoder_hidden_states=ehs, added_cond_kwargs=added, return_dict=False)
        t_step = time.perf_counter() - t0
    log(f"cpu_baseline: reference UNet2DConditionModel.forward (B=2, 128x128 latents, fp32) took {t_step:.1f} s on {threads} threads")
    del unet
    t_dec, dec_note = None, ""
    if time.perf_counter() - t_start + t_step * VAE_TFLOP / UNET_TFLOP < budget_s:
        vae = RR.build_vae(ref, vcfg, vae_sd, "cpu", torch.float32)
        with torch.no_grad():
            t0 = time.perf_counter()
            vae.decode(torch.randn((1, 4, 128, 128), generator=g), return_dict=False)
            t_dec = time.perf_counter() - t0
        log(f"cpu_baseline: reference AutoencoderKL.decode (1024x1024, fp32) took {t_dec:.1f} s")
    else:
        t_dec = t_step * VAE_TFLOP / UNET_TFLOP
        dec_note = " (decode not run: scaled by FLOPs from the step)"
    secs = 50 * t_step + t_dec
    return {"value": 1.0 / secs, "unit": "images/s", "cores": threads, "kind": "reference", "extrapolated": True,
            "seconds_per_image": secs,
            "sample": f"1 full-size CFG-batched reference U-Net step ({t_step:.1f} s, {UNET_TFLOP} TFLOP) + 1 reference VAE decode "
                      f"({t_dec:.1f} s{dec_note}), fp32, huggingface/diffusers 0.40.0.dev0 on torch CPU; image = 50 x step + decode",
            "cpu_tflops": (UNET_TFLOP + (VAE_TFLOP if not dec_note else 0)) / (t_step + (t_dec if not dec_note else 0))}


def cpu_baseline(unet_sd, cfg_full, budget_s=25.0):
    """CPU leg fallback, kind "port" (no reference archive on this machine): the oracle restatement of the U-Net forward
    (oracle/reference_math.py) on the host cores, on a bounded sample: CFG-batched forward of the FULL SDXL architecture at
    32x32 latents, fp32.  images/s is EXTRAPOLATED by algorithmic FLOPs (CPU_SAMPLE_TFLOP per sample, 686.6 TFLOP per image)."""
    from oracle import reference_math as R
    threads = _host_threads()
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
            "extrapolated": True,
            "sample": f"{reps}x oracle fp32 CFG-batched SDXL U-Net forward at 32x32 latents ({CPU_SAMPLE_TFLOP:.3f} "
                      f"TFLOP each, {tot:.1f} s); EXTRAPOLATED by FLOPs to 686.6 TFLOP/image",
            "cpu_tflops": tflops}


def _psnr01(a, b):
    """PSNR on [0, 1] images (output_type="pt" convention, 40 dB = MSE 1e-4)."""
    a = (a.float() * 0.5 + 0.5).clamp(0, 1)
    b = (b.float() * 0.5 + 0.5).clamp(0, 1)
    mse = float((a - b).pow(2).mean())
    return 10.0 * torch.log10(torch.tensor(1.0 / max(mse, 1e-12))).item()


def reference_pipeline_on_device(unet_sd, vae_sd, ucfg, vcfg, inp, steps, dtype, timed=False, budget_s=240.0):
    """CHECKER / BASELINE, never the product: StableDiffusionXLPipeline's denoising loop
    (pipeline_stable_diffusion_xl.py:1193-1257: cat([latents] * 2), scale_model_input, U-Net, uncond + g (text - uncond),
    scheduler.step) and vae.decode(latents / scaling_factor) (:1262-1290), written over the oracle restatement of the
    reference modules (plain torch ops = what the reference executes), on THIS GPU through PyTorch-ROCm in ``dtype``.
    Returns (final latents, image in [-1, 1], seconds for the timed loop + decode or None)."""
    from diffusers_amd import factory
    from oracle import reference_math as R
    from oracle import samplers as OS
    dev = inp["latents"].device
    usd = {k: v.to(dev, dtype) for k, v in unet_sd.items()}
    vsd = {k: v.to(dev, dtype) for k, v in vae_sd.items()}
    ehs = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], dim=0).to(dtype)
    added = {"text_embeds": torch.cat([inp["negative_pooled"], inp["pooled"]], dim=0).to(dtype),
             "time_ids": torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]], device=dev).repeat(2, 1)}
    sch = OS.EulerOracle(**factory.SDXL_SCHEDULER)

    def loop(n_run):
        sch.set_timesteps(steps)
        x = (inp["latents"].to(dtype) * sch.init_noise_sigma).to(dtype)
        for t_ in sch.timesteps[:n_run]:
            xin = sch.scale_model_input(torch.cat([x] * 2))
            eps = R.unet_forward(usd, ucfg, xin, float(t_), ehs, added)
            u, c = eps.chunk(2)
            x = sch.step(u + GUIDANCE * (c - u), x)
        return x

    with torch.no_grad():
        # warm-up: library handles, MIOpen / hipBLASLt solution search -- a ONE-TIME cost that has been seen to take from 2 s
        # to 100 s on otherwise identical boxes, so it is not what the budget is judged on ...
        x = loop(2)
        R.vae_decode(vsd, vcfg, x / vcfg["scaling_factor"])
        torch.cuda.synchronize()
        # ... the steady state is: two more steps, timed
        t1 = time.perf_counter()
        loop(2)
        torch.cuda.synchronize()
        per_step = (time.perf_counter() - t1) / 2
        if per_step * steps > budget_s:
            raise TimeoutError(f"{per_step:.2f} s per warm denoising step: the {steps}-step run would exceed {budget_s:.0f} s")
        t0 = time.perf_counter()
        x = loop(steps)
        img = R.vae_decode(vsd, vcfg, x / vcfg["scaling_factor"])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return x, img, (dt if timed else None)


def reference_package_on_device(ref, unet_sd, vae_sd, ucfg, vcfg, inp, steps, dtype, timed=False, hw=1024):
    """CHECKER / BASELINE, never the product: the REAL reference -- `StableDiffusionXLPipeline.__call__`
    (pipeline_stable_diffusion_xl.py:823-1308) over the reference's own `UNet2DConditionModel`, `AutoencoderKL` and
    `EulerDiscreteScheduler`, imported from the oracle/_ref archive -- on THIS GPU through PyTorch-ROCm in ``dtype``, on the
    weights / embeddings / latents the engine ran on.  "Stock diffusers on PyTorch-ROCm": the harness pattern of
    benchmarks/benchmarking_utils.py:22-68 (warm-up calls, then timed calls bracketed by synchronize; the best of two is
    reported).  Returns (final latents, image in [-1, 1], seconds per image or None)."""
    from diffusers_amd import factory
    from oracle import ref_runtime as RR
    dev = inp["latents"].device
    pipe = RR.build_sdxl_pipeline(ref, ucfg, vcfg, unet_sd, vae_sd, factory.SDXL_SCHEDULER, dev, dtype)
    RR.run_sdxl(pipe, inp, 2, GUIDANCE, hw, dtype)            # warm-up: library handles, MIOpen / hipBLASLt solution search
    torch.cuda.synchronize()
    secs = None
    if timed:
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            RR.run_sdxl(pipe, inp, steps, GUIDANCE, hw, dtype)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        secs = best
    img01, lat = RR.run_sdxl(pipe, inp, steps, GUIDANCE, hw, dtype, want_latents=True)
    torch.cuda.synchronize()
    return lat, img01.float() * 2.0 - 1.0, secs


def reference_legs(engine_img, engine_lat, unet_sd, vae_sd, ucfg, vcfg, ucfg_min, vcfg_min, inp, steps, engine_images_per_s):
    """parity + torch_rocm_baseline objects (rank 0, N = 1, outside the timed region).  With the reference archive present
    (oracle/_ref/diffusers_ref.zip, built by oracle/build_ref.py) both legs run the REAL reference package (`kind`:
    "reference"); without it they fall back to the oracle restatement of the module graph (`kind`: "port")."""
    from oracle import ref_runtime as RR
    ref = None
    try:
        ref = RR.load_reference()
    except Exception as e:  # a broken archive must not cost the measurement
        log(f"reference archive unusable ({type(e).__name__}: {e}); falling back to the oracle port")
    kind = "reference" if ref is not None else "port"
    what = ("stock diffusers 0.40.0.dev0 StableDiffusionXLPipeline.__call__ over reference UNet2DConditionModel / AutoencoderKL / "
            "EulerDiscreteScheduler (oracle/_ref archive)" if ref is not None else
            "port of the reference module graph (oracle restatement; F.conv2d / F.linear / F.scaled_dot_product_attention / F.group_norm)")
    parity = {"psnr_db": None, "ref": f"{what}, PyTorch-ROCm fp32, this GPU", "kind": kind, "steps": steps, "noise_floor_db": None}
    base = {"images_per_s": None, "kind": kind, "what": f"{what}, PyTorch-ROCm eager bf16", "unit": "images/s"}

    def run(dtype, timed):
        if ref is not None:
            return reference_package_on_device(ref, unet_sd, vae_sd, ucfg_min, vcfg_min, inp, steps, dtype, timed=timed)
        return reference_pipeline_on_device(unet_sd, vae_sd, ucfg, vcfg, inp, steps, dtype, timed=timed)
    try:
        log(f"torch_rocm_baseline leg ({kind}): eager bf16, {steps} steps + decode")
        lat_b, img_b, secs = run(torch.bfloat16, True)
        base.update({"images_per_s": 1.0 / secs, "seconds_per_image": secs})
        log(f"torch_rocm_baseline: {secs:.2f} s / image")
    except Exception as e:  # a failing baseline must not cost the measurement
        base["error"] = f"{type(e).__name__}: {e}"
        lat_b = img_b = None
        log(f"torch_rocm_baseline failed: {base['error']}")
    try:
        log(f"parity leg ({kind}): fp32, {steps} steps + decode")
        img_f = None
        lat_f, img_f, _ = run(torch.float32, False)
        parity["psnr_db"] = _psnr01(engine_img, img_f)
        d = engine_lat.float() - lat_f.float()
        parity["latents_rel_rms"] = float(d.pow(2).mean().sqrt() / lat_f.float().pow(2).mean().sqrt())
        if img_b is not None:
            parity["noise_floor_db"] = _psnr01(img_b, img_f)            # reference bf16 vs reference fp32
            parity["vs_torch_bf16_db"] = _psnr01(engine_img, img_b)
            db = lat_b.float() - lat_f.float()
            parity["noise_floor_latents_rel_rms"] = float(db.pow(2).mean().sqrt() / lat_f.float().pow(2).mean().sqrt())
        parity["finite"] = bool(torch.isfinite(img_f).all())
        log(f"parity: engine vs fp32 {parity['psnr_db']:.1f} dB, torch-bf16 vs fp32 {parity['noise_floor_db']} dB")
    except Exception as e:
        parity["error"] = f"{type(e).__name__}: {e}"
        log(f"parity leg failed: {parity['error']}")
    vs = engine_images_per_s / base["images_per_s"] if base.get("images_per_s") else None
    return parity, base, vs, (img_f if parity.get("psnr_db") is not None else None)


# ---- the other BASELINE configs under the same driver contract ------------------------------------------
WAN_VAE_DECODE_TFLOP = 275.0   # algorithmic (unpadded channels) flop of AutoencoderKLWan.decode at 21 x 60 x 104 latents
OTHER_CONFIGS = {
    # name: (metric, unit, workload description, default sampler steps, algorithmic TFLOP per unit at that many steps)
    "sd15": ("images/sec @ stable-diffusion-v1-5 512x512 50-step DDIM CFG 7.5 bf16", "images/s",
             "stable-diffusion-v1-5 U-Net (860 M params) x {n} DDIM steps, CFG 7.5 (batch 2), 64x64 latents + AutoencoderKL decode "
             "to 512x512; 1 prompt per GPU", 50, lambda n: n * 1.6065 + 2.5145),
    "flux": ("images/sec @ FLUX.1-schnell 1024x1024 4-step FlowMatchEuler bf16", "images/s",
             "FLUX.1-schnell transformer (11.9 B params) x {n} FlowMatchEuler steps, no CFG, 4096 image + 512 text tokens + 16-channel "
             "AutoencoderKL decode to 1024x1024; 1 prompt per GPU", 4, lambda n: n * 74.3846 + 10.5),
    "wan": ("videos/sec @ Wan2.1-T2V-1.3B 832x480x81 50-step CFG 5.0 bf16", "videos/s",
            "Wan2.1-T2V-1.3B as shipped (pipeline_wan.py:52-59, :560-661): transformer x {n} UniPC steps (flow_prediction, order 2, fp32 "
            "latents), CFG 5.0 (cond + uncond as one batch-2 call), 32 760 tokens, + AutoencoderKLWan.decode of the 21 x 60 x 104 latents "
            "to 81 x 480 x 832; 1 prompt per GPU", 50, lambda n: n * 2 * 283.0018 + WAN_VAE_DECODE_TFLOP),
    "ddpm": ("images/sec @ google/ddpm-cat-256 50-step DDPM", "images/s",
             "UNet2DModel (114 M params) x {n} ancestral DDPM steps at 256x256, batch 1 per GPU", 50, lambda n: n * 0.4970),
}


def _build_other(config, dev, rank, tiny, no_graph):
    """(pipeline, unit(n) -> callable running one unit with n sampler steps, eager(n) -> the same without HIP-graph replay)."""
    from diffusers_amd import factory
    bf = torch.bfloat16
    g = torch.Generator("cpu").manual_seed(1234 + rank)
    if config == "sd15":
        pipe = factory.build_sd15_pipeline(device=dev, tiny=tiny, seed=0)
        cd, lat = (64, 16) if tiny else (768, 64)
        pe, ne = (torch.randn((1, 77, cd), generator=g).to(bf).to(dev) for _ in range(2))
        x = torch.randn((1, 4, lat, lat), generator=g).to(bf).to(dev)

        def unit(n, graph=not no_graph):
            return lambda: pipe(prompt_embeds=pe, negative_prompt_embeds=ne, latents=x.clone(), num_inference_steps=n,
                                guidance_scale=7.5, output_type="raw", use_graph=graph).images
    elif config == "flux":
        if tiny:
            raise SystemExit("--config flux has no --tiny form here (tests/ cover the tiny pipeline)")
        pipe = factory.build_flux_pipeline(device=dev, tiny=False, seed=5)
        pe = torch.randn((1, 512, 4096), generator=g).to(bf).to(dev)
        pooled = torch.randn((1, 768), generator=g).to(bf).to(dev)
        x = torch.randn((1, 4096, 64), generator=g).to(bf).to(dev)

        def unit(n, graph=not no_graph):
            return lambda: pipe(prompt_embeds=pe, pooled_prompt_embeds=pooled, latents=x, num_inference_steps=n, guidance_scale=0.0,
                                height=1024, width=1024, output_type="raw", use_graph=graph).images
    elif config == "wan":
        if tiny:
            raise SystemExit("--config wan has no --tiny form here (tests/ cover the tiny pipeline)")
        pipe = factory.build_wan_pipeline(device=dev, tiny=False, seed=9, scheduler="unipc", with_vae=True)
        pe, ne = (torch.randn((1, 512, 4096), generator=g).to(bf).to(dev) for _ in range(2))
        x = torch.randn((1, 16, 21, 60, 104), generator=g).to(dev)                      # fp32 latents (pipeline_wan.py:559-572)

        def unit(n, graph=not no_graph, output_type="raw"):
            return lambda: pipe(prompt_embeds=pe, negative_prompt_embeds=ne, latents=x, num_inference_steps=n, guidance_scale=5.0,
                                height=480, width=832, num_frames=81, use_graph=graph, output_type=output_type).images
    else:
        pipe = factory.build_ddpm_pipeline(device=dev, tiny=tiny, seed=0)

        def unit(n, graph=not no_graph):
            return lambda: pipe(batch_size=1, generator=torch.Generator().manual_seed(rank), num_inference_steps=n,
                                output_type="pt", use_graph=graph).images
    return pipe, unit


def _other_kernels(unit):
    """`roofline.kernels` of another config: HIP events around every launch of the kernel families in ONE eager sampler step
    (+ the decode where the unit has one): the algorithmic work is summed from the launches themselves."""
    run = unit(1, False)
    run()                                   # untimed warm pass (variant lookup)
    fam, timing, _ = measured_families(run)
    kernels = []
    for name, kern, bound in (("igemm", "igemm_bf16_kernel + igemm2_bf16_kernel + gemm3_bf16_kernel (Linear, Conv2d, paired launches)", "mfma"),
                              ("attention", "attn2_fwd_kernel / attn_fwd_kernel (flash attention forward)", "mfma"),
                              ("groupnorm", "gn_stats_kernel + gn_apply_kernel", "hbm"), ("layernorm", "layernorm_kernel", "hbm")):
        if name in fam:
            kernels.append(_kernel_entry(name, kern, bound, fam[name], {"scope": "one sampler step + decode, eager", "timing": timing}))
    return kernels


def _other_cpu_baseline(config, unit_tflop, n_steps):
    """`cpu_baseline` of another config, kind "reference": the reference classes (oracle/_ref archive, random weights) on the host
    cores in fp32 on a BOUNDED sample of the same workload (oracle.ref_runtime.cpu_step_seconds says which)."""
    from diffusers_amd import init as dinit
    from oracle import ref_runtime as RR
    unit_name = "videos/s" if config == "wan" else "images/s"
    ref = RR.load_reference() if RR.available() else None
    if ref is None:
        return {"value": None, "unit": unit_name, "cores": None, "kind": "reference", "sample": "not measured: the reference archive did not ship"}
    cfgs = {"sd15": {"unet": dinit.SD15_UNET, "vae": dinit.SD_VAE}, "ddpm": {"unet": dinit.DDPM_CAT},
            "flux": {"transformer": dinit.FLUX_SCHNELL}, "wan": {"transformer": dinit.WAN_1_3B}}[config]
    threads = _host_threads()
    ts, tfl_s, td, tfl_d, what = RR.cpu_step_seconds(ref, config, cfgs, threads)
    cpu_tflops = (tfl_s + tfl_d) / (ts + td)
    if config in ("sd15", "ddpm"):
        secs = n_steps * ts + td                       # the steps of a unit are cost-identical: the only extrapolation
    else:
        secs = unit_tflop / cpu_tflops                 # depth- / length-reduced sample: scaled by algorithmic FLOPs
    return {"value": 1.0 / secs, "unit": unit_name, "cores": threads, "kind": "reference", "extrapolated": True,
            "seconds_per_unit": secs, "cpu_tflops": cpu_tflops,
            "sample": what + ", fp32, huggingface/diffusers 0.40.0.dev0 on torch CPU"}


def measure_other_config(config, dev, rank, world, args, units, warmup, n_steps=None, extrapolate_from=None, legs=True):
    """One BASELINE config other than the headline under the driver contract's timing rule: `warmup` untimed units, `units` timed
    units between synchronisations (+ barriers when world > 1), max over ranks.  ``extrapolate_from`` = k: a unit is timed with k
    sampler steps and scaled to the config's step count (flagged) -- the 50-step Wan video inside the default run.
    ``legs``: per-family `roofline.kernels` and the `cpu_baseline` (rank 0, world == 1)."""
    from diffusers_amd import distributed as D
    metric, unit_name, what, default_steps, tflop_of = OTHER_CONFIGS[config]
    n = n_steps or default_steps
    run_steps = extrapolate_from or n
    pipe, unit = _build_other(config, dev, rank, args.tiny, args.no_graph)
    fn = unit(run_steps)
    out = None
    for _ in range(max(warmup, 1)):          # at least one: variant lookup, graph capture
        out = fn()
    _sync()
    if _pg():
        torch.distributed.barrier()
    _sync()
    t0 = time.perf_counter()
    for _ in range(units):
        out = fn()
    _sync()
    mine_s = time.perf_counter() - t0
    if _pg():
        torch.distributed.barrier()
    _sync()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    scale = n / run_steps if extrapolate_from else 1.0
    side = {}
    if config == "wan":
        # the unit holds a decode that does not scale with the step count: time the latents-only unit next to it (outside the
        # contract's timed region) -- decode = full - latents-only; an extrapolated unit scales only the sampler part
        lat_fn = unit(run_steps, output_type="latent")
        lat_fn()
        _sync()
        t1 = time.perf_counter()
        for _ in range(units):
            lat_fn()
        _sync()
        lat_s = (time.perf_counter() - t1) / units
        full_s = elapsed / units
        dec_s = max(full_s - lat_s, 0.0)
        side = {"seconds_per_unit_latents_only": lat_s * scale, "seconds_decode": dec_s,
                "videos_per_s_without_decode": 1.0 / (lat_s * scale), "seconds_per_sampler_step": lat_s / run_steps}
        if extrapolate_from:
            elapsed_n = units * (lat_s * scale + dec_s)
            scale = elapsed_n / elapsed
    value = world * units / (elapsed * scale)
    tfl = tflop_of(n)
    ach = value / world * tfl
    res = {"metric": metric, "value": value, "unit": unit_name, "n_gpus": world, "steps": units, "warmup": max(warmup, 1),
           "ms_per_step": 1000.0 * elapsed * scale / units, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16", "data": "synthetic (seeded random weights, embeddings, latents)",
           "config": {"workload": what.format(n=n), "global_batch": world, "parallelism": f"dp{world} (independent prompts, replicas)",
                      "denoise_steps": n, "output_finite": bool(torch.isfinite(out.float()).all()),
                      "rank_seconds_per_unit": mine_s * scale / units, "tuned_live": _tuned_live(),
                      "algorithmic_tflop_per_unit": tfl, **side},
           "roofline": {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                        "traffic": None, "kernel": "whole unit (end to end): algorithmic TFLOP of SURVEY.md 8d / wall time"}}
    if extrapolate_from:
        res["config"]["extrapolated"] = (f"timed with {run_steps} sampler steps per unit and scaled to {n} (the steps are cost-identical; "
                                         f"`python bench.py --config {config}` runs the full unit)")
    if legs and rank == 0 and world == 1 and not args.tiny:
        try:
            res["roofline"]["kernels"] = _other_kernels(unit)
        except Exception as e:  # a diagnostic must not cost the line
            res["roofline"]["kernels_error"] = f"{type(e).__name__}: {e}"
        if not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = _other_cpu_baseline(config, tfl, n)
            except Exception as e:
                res["cpu_baseline"] = {"value": None, "unit": unit_name, "cores": None, "kind": "reference",
                                       "sample": f"not measured: {type(e).__name__}: {e}"}
    if "cpu_baseline" not in res:
        res["cpu_baseline"] = {"value": None, "unit": unit_name, "cores": None, "kind": "reference",
                               "sample": "not measured in this run (rank 0 at N = 1 without --no-cpu-baseline measures it)"}
    del pipe
    torch.cuda.empty_cache()
    return res


def run_other_config(args, argv):
    """`--config sd15 | flux | wan | ddpm`: the same contract (W untimed units, K timed units between barriers, max over ranks, ONE
    JSON line from rank 0) on the other BASELINE configs; one prompt per rank, no collective in the data path.  `roofline` is the
    whole unit against the MFMA peak (algorithmic TFLOP of SURVEY.md 8d / wall time) with per-family `kernels` from one eager
    step; `cpu_baseline` is the reference classes on a bounded sample of the same workload."""
    from diffusers_amd import distributed as D
    rank, world, local = D.init_from_env()
    dev = _device(local)
    log(f"rank {rank}/{world}: {args.config} on {dev}")
    result = measure_other_config(args.config, dev, rank, world, args, args.steps, args.warmup, n_steps=args.denoise_steps)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if _pg():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return result


def other_configs_leg(dev, args):
    """The other four BASELINE configs inside the default run (rank 0, N = 1, outside the timed region): three timed units each
    (Wan: two units of 3 sampler steps, scaled to 50 and flagged), each with its per-family kernels and its CPU baseline."""
    out = []
    for cfg, units, extra in (("sd15", 3, {}), ("flux", 3, {}), ("ddpm", 3, {}), ("wan", 2, {"extrapolate_from": 3})):
        log(f"other_configs leg: {cfg}")
        try:
            r = measure_other_config(cfg, dev, 0, 1, args, units, 1, **extra)
            out.append({k: r[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline",
                                          "cpu_baseline")})
            log(f"other_configs: {cfg} {r['value']:.4g} {r['unit']}")
        except Exception as e:  # never lose the headline line to a side leg
            out.append({"metric": OTHER_CONFIGS[cfg][0], "value": None, "error": f"{type(e).__name__}: {e}"})
            log(f"other_configs: {cfg} failed: {type(e).__name__}: {e}")
    return out


def dropin_leg(unet, vae, inp, steps, hw, ref_img_fp32, engine_images_per_s):
    """The drop-in promise, timed at full size (rank 0, N = 1, outside the timed region): the REAL reference
    `StableDiffusionXLPipeline.__call__` (pipeline_stable_diffusion_xl.py:823-1308, from the oracle/_ref archive) over ENGINE `unet` /
    `vae` / `scheduler` objects in its component slots (pipeline_utils.py:224-252): every step is the reference's own Python --
    cat([latents] * 2), scale_model_input, unet(...), torch CFG combine, scheduler.step -- launching the HIP kernels eagerly.
    images/s, the ratio to the engine's graphed pipeline, PSNR against the all-reference fp32 run of the parity leg."""
    import diffusers_amd as da
    from diffusers_amd import factory
    from oracle import ref_runtime as RR
    ref = RR.load_reference()
    dev = inp["latents"].device
    bf = torch.bfloat16
    sch = da.EulerDiscreteScheduler(**factory.SDXL_SCHEDULER)
    pipe = RR.engine_under_reference_sdxl(ref, unet, vae, sch, dev, bf, int(inp["pooled"].shape[-1]))
    RR.run_sdxl(pipe, inp, 2, GUIDANCE, hw, bf)                 # warm-up
    torch.cuda.synchronize()
    best = None
    img = None
    for _ in range(2):
        t0 = time.perf_counter()
        img, _ = RR.run_sdxl(pipe, inp, steps, GUIDANCE, hw, bf)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out = {"images_per_s": 1.0 / best, "seconds_per_image": best, "unit": "images/s", "steps": steps,
           "vs_engine_pipeline": (1.0 / best) / engine_images_per_s,
           "what": "stock diffusers 0.40.0.dev0 StableDiffusionXLPipeline.__call__ (oracle/_ref archive) over engine UNet2DConditionModel / "
                   "AutoencoderKL / EulerDiscreteScheduler registered as its components; eager launches, best of two"}
    if ref_img_fp32 is not None:
        out["psnr_db_vs_reference_fp32"] = _psnr01(img.float() * 2.0 - 1.0, ref_img_fp32)
    return out


def two_in_flight_leg(unet, vae, dev, denoise_steps, hw, one_at_a_time_images_per_s, images_per_pipeline=2):
    """INFORMATIONAL -- not `value`, not BASELINE.json's protocol (one prompt per GPU, one image at a time).  What a serving host that
    keeps TWO independent batch-1 requests in flight per GPU gets out of the same kernels: two pipelines over the SAME U-Net / VAE
    weights, each with its own scheduler, step graph, stream and -- through pipelines.STREAM_DOMAIN -- its own capture stream and
    workspaces, driven from two host threads.  A batch-1 launch leaves launch-latency and tail bubbles (one tile per CU, 5-10 us of
    fixed cost per launch); the second request's launches fill them.  The step graphs reproduce their one-at-a-time latents bit for bit;
    the eager VAE decode is serialised across the two requests and still, now and then, does not reproduce its bits when it overlaps
    the other request's step replays (pipelines._exclusive_decode: cause not isolated) -- the leg reports whether THIS run's images were
    bit-identical and the largest difference.  Measured: profiles/r06_two_in_flight.json (+ 11 %)."""
    import threading
    from diffusers_amd import factory, pipelines as P
    from diffusers_amd.schedulers import EulerDiscreteScheduler
    pipes = [P.StableDiffusionXLPipeline(vae=vae, unet=unet, scheduler=EulerDiscreteScheduler(**factory.SDXL_SCHEDULER)) for _ in range(2)]
    inp = synth_inputs(2, False, dev)
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs, errs = [None, None], []

    def work(i, count):
        try:
            P.STREAM_DOMAIN.tag = 101 + i
            with torch.cuda.stream(streams[i]):
                for _ in range(count):
                    outs[i] = pipes[i](prompt_embeds=inp["prompt_embeds"][i:i + 1], negative_prompt_embeds=inp["negative_prompt_embeds"][i:i + 1],
                                       pooled_prompt_embeds=inp["pooled"][i:i + 1], negative_pooled_prompt_embeds=inp["negative_pooled"][i:i + 1],
                                       latents=inp["latents"][i:i + 1].clone(), num_inference_steps=denoise_steps, guidance_scale=GUIDANCE,
                                       height=hw, width=hw, output_type="pt").images
                streams[i].synchronize()
        except Exception as e:       # surfaced by the caller
            errs.append(e)

    def run(concurrent, count):
        th = [threading.Thread(target=work, args=(i, count)) for i in range(2)]
        if concurrent:
            for t in th:
                t.start()
            for t in th:
                t.join()
        else:
            for t in th:
                t.start()
                t.join()
        if errs:
            raise errs[0]
    run(False, 1)                       # warm-up / capture, one pipeline at a time
    _sync()
    ref = [o.clone() for o in outs]
    t0 = time.perf_counter()
    run(False, images_per_pipeline)
    _sync()
    seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    run(True, images_per_pipeline)
    _sync()
    con = time.perf_counter() - t0
    same = all(torch.equal(o, r) for o, r in zip(outs, ref))
    worst = max(float((o.float() - r.float()).abs().max()) for o, r in zip(outs, ref))
    n = 2 * images_per_pipeline
    return {"images_per_s": n / con, "one_at_a_time_images_per_s_same_leg": n / seq, "vs_one_at_a_time": seq / con,
            "headline_images_per_s": one_at_a_time_images_per_s, "bit_identical_to_one_at_a_time": same,
            "max_abs_diff_vs_one_at_a_time": worst, "images": n,
            "what": "two independent batch-1 SDXL requests in flight on one GPU (two host threads, two streams, own step graphs and "
                    "workspaces, shared weights; decodes serialised); INFORMATIONAL, a measurement and not a supported mode: the step "
                    "graphs reproduce their one-at-a-time latents, a VAE decode that overlaps the other request's launches "
                    "occasionally does not (pipelines._exclusive_decode; [0, 1] image units); the headline `value` is one request at a time"}


def _pg() -> bool:
    """A default process group exists: N > 1 ranks, or ONE rank started by a launcher (distributed.init_from_env)."""
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, argv))
    if args.config != "sdxl":
        return run_other_config(args, argv)
    args.denoise_steps = args.denoise_steps or 50
    from diffusers_amd import distributed as D
    rank, world, local = D.init_from_env()
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dev = _device(local)

    from diffusers_amd import factory, init as dinit
    from diffusers_amd.autoencoder_kl import _DEFAULTS as VD
    from diffusers_amd.unet_2d_condition import _DEFAULTS as UD
    ucfg = dinit.TINY_SDXL_UNET if args.tiny else dinit.SDXL_UNET
    vcfg = dinit.TINY_VAE if args.tiny else dinit.SDXL_VAE
    log(f"rank {rank}/{world}: building models on {dev}")
    unet, unet_sd = factory.build_unet(ucfg, seed=0, device=dev, init_device=str(dev))
    vae, vae_sd = factory.build_vae(vcfg, seed=1, device=dev, init_device=str(dev))
    from diffusers_amd.pipelines import StableDiffusionXLPipeline
    from diffusers_amd.schedulers import EulerDiscreteScheduler
    pipe = StableDiffusionXLPipeline(vae=vae, unet=unet, scheduler=EulerDiscreteScheduler(**factory.SDXL_SCHEDULER))

    # rank 0 owns the "prompts"; one broadcast of the embeddings over RCCL, then every rank keeps its shard
    inputs = synth_inputs(world, args.tiny, dev)
    if _pg():          # (also with ONE launched rank: the RCCL path of the 8-GPU job runs on a single-GPU box)
        if rank != 0:
            inputs = {k: torch.empty_like(v) for k, v in inputs.items()}
        D.broadcast_tensors(inputs, src=0)
    mine = D.select_shard(inputs, D.shard_indices(world, rank, world))
    hw = 128 if args.tiny else 1024

    state = {"graph": not args.no_graph}

    def one_image(output_type="pt"):
        # the timed call returns [0, 1] images, SURVEY.md 8d's protocol (output_type="pt"); the postprocess is fused into the
        # decoder's last pass.  The parity legs below ask for "raw" ([-1, 1]) / "latent" explicitly, outside the timed region.
        return pipe(prompt_embeds=mine["prompt_embeds"], negative_prompt_embeds=mine["negative_prompt_embeds"],
                    pooled_prompt_embeds=mine["pooled"], negative_pooled_prompt_embeds=mine["negative_pooled"],
                    latents=mine["latents"].clone(), num_inference_steps=args.denoise_steps, guidance_scale=GUIDANCE,
                    height=hw, width=hw, output_type=output_type, use_graph=state["graph"]).images

    box = None
    if not args.tiny and dev.type == "cuda":
        try:
            box = box_mfma_probe()
            log(f"box normaliser: {box['tflops']:.0f} TFLOP/s on the register-only MFMA loop ({box['ms']:.1f} ms)")
        except Exception as e:  # a diagnostic must not cost the line
            box = {"tflops": None, "error": f"{type(e).__name__}: {e}"}
        try:
            box["memory"] = box_memory_probe()
            log(f"box normaliser: {box['memory']['hbm_copy_gbps']:.0f} GB/s device copy, "
                f"{box['memory']['cold_gemm_us']:.1f} us per cold-weight 2048 x 1280 x 1280 projection")
        except Exception as e:
            box["memory"] = {"hbm_copy_gbps": None, "cold_gemm_us": None, "error": f"{type(e).__name__}: {e}"}
    log("warm-up (tunes GEMM variants for unseen shapes, captures the denoising-step HIP graph)")
    img = None
    for _ in range(args.warmup):
        try:
            img = one_image()
        except RuntimeError as e:
            # a capture that another component of the process invalidated must not cost the measurement: the eager
            # launch path is the same kernels in the same order, and the JSON line says which one ran
            if not state["graph"]:
                raise
            log(f"HIP-graph capture failed ({e}); continuing with eager launches")
            _sync()
            state["graph"] = False
            pipe._graph = None
            img = one_image()
    _sync()
    if _pg() and torch.distributed.get_world_size() != int(os.environ.get("WORLD_SIZE", world)):
        # the line's `n_gpus` and `value` assume every launched rank is inside the group that brackets the timed region
        raise SystemExit(f"process group has {torch.distributed.get_world_size()} ranks, the launcher started "
                         f"{os.environ.get('WORLD_SIZE')}: refusing to time a partial job")
    log("timed region")
    if _pg():
        torch.distributed.barrier()
    _sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img = one_image()
    _sync()
    mine_s = time.perf_counter() - t0
    if _pg():
        torch.distributed.barrier()
    _sync()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    per_rank = [args.steps / mine_s]
    if _pg():
        tt = torch.tensor([args.steps / mine_s], dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(tt) for _ in range(world)]
        torch.distributed.all_gather(gathered, tt)
        per_rank = [float(g.item()) for g in gathered]
    # per-rank box normalisers on the line: a replica on a slow-kind box (or starved of host cores) is visible next to its rate
    cold = (box or {}).get("memory", {}).get("cold_gemm_us")
    per_rank_cold = [cold]
    per_rank_cpus = [len(os.sched_getaffinity(0))]
    if _pg():
        tt = torch.tensor([cold if cold is not None else -1.0, float(per_rank_cpus[0])], dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(tt) for _ in range(world)]
        torch.distributed.all_gather(gathered, tt)
        per_rank_cold = [(float(g[0]) if float(g[0]) >= 0 else None) for g in gathered]
        per_rank_cpus = [int(g[1]) for g in gathered]
    finite = bool(torch.isfinite(img.float()).all())
    log(f"timed region done: {elapsed:.3f} s for {args.steps} image(s) per GPU")
    if args.save_tuning and rank == 0:
        from diffusers_amd import tuning
        tuning.save(args.save_tuning)

    value = world * args.steps / elapsed
    result = {
        "metric": "images/sec @ SDXL-base 1024x1024 50-step EulerDiscrete CFG bf16",
        "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic (seeded random weights, embeddings, latents)",
        "config": {"workload": "SDXL-base-1.0 U-Net (2567 M params) x 50 EulerDiscrete steps, CFG 5.0 (batch 2), "
                               "128x128 latents + AutoencoderKL decode to 1024x1024; 1 prompt per GPU"
                               if not args.tiny else "TINY plumbing config (not a benchmark)",
                   "global_batch": world, "parallelism": f"dp{world} (independent prompts, replicas)",
                   "denoise_steps": args.denoise_steps, "output_type": "pt", "hip_graph": state["graph"], "output_finite": finite,
                   "rccl_ranks": torch.distributed.get_world_size() if _pg() else 1,
                   "process_group": D.backend_name(),     # "nccl" = RCCL; None: a lone process without a launcher (no collectives)
                   "tuned_live": _tuned_live(),
                   "box_mfma_tflops": box["tflops"] if box else None, "box_mfma_probe": box,
                   "box_hbm_copy_gbps": (box or {}).get("memory", {}).get("hbm_copy_gbps"),
                   "box_cold_gemm_us": (box or {}).get("memory", {}).get("cold_gemm_us"),
                   "images_per_s_per_rank": per_rank, "box_cold_gemm_us_per_rank": per_rank_cold,
                   "host_cores_per_rank": per_rank_cpus,
                   "cpu_binding": ("per-rank NUMA-local core share (distributed.bind_rank_to_cpus)" if world > 1 and
                                   os.environ.get("DIFFUSERS_AMD_BIND", "1") != "0" else "none")},
    }

    if rank == 0 and not args.tiny:
        if not args.no_roofline:
            log("roofline leg: one eager denoising step with HIP events around every igemm launch")
            try:
                result["roofline"] = roofline_leg(pipe, mine, world, value, live=not args.no_live_traffic)
            except Exception as e:  # never lose the measured line to a diagnostic leg
                result["roofline"] = {"bound": "mfma", "achieved": None, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": None, "traffic": None, "error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_reference:
            full_u, full_v = dict(UD), dict(VD)
            full_u.update(ucfg)
            full_v.update(vcfg)
            # the engine's own image / latents for the parity check: the same call as the timed one
            eng_lat = one_image("latent").clone()
            eng_img = one_image("raw")
            parity, base, vs, img_f = reference_legs(eng_img, eng_lat, unet_sd, vae_sd, full_u, full_v, ucfg, vcfg, mine,
                                                     args.denoise_steps, value)
            result["parity"], result["torch_rocm_baseline"], result["vs_torch_rocm"] = parity, base, vs
            if parity.get("kind") == "reference":
                log("dropin leg: the reference pipeline's own __call__ over engine components, full size")
                try:
                    result["dropin"] = dropin_leg(unet, vae, mine, args.denoise_steps, hw, img_f, value)
                    log(f"dropin: {result['dropin']['seconds_per_image']:.2f} s / image = {result['dropin']['vs_engine_pipeline']:.3f} x the "
                        f"graphed engine pipeline, {result['dropin'].get('psnr_db_vs_reference_fp32')} dB vs the reference in fp32")
                except Exception as e:  # never lose the measured line to a side leg
                    result["dropin"] = {"images_per_s": None, "error": f"{type(e).__name__}: {e}"}
                    log(f"dropin leg failed: {result['dropin']['error']}")
            del img_f
        if world == 1 and not args.no_cpu_baseline:
            full = dict(UD)
            full.update(ucfg)
            log("cpu_baseline leg")
            import signal

            def _alarm(signum, frame):
                raise TimeoutError("cpu_baseline exceeded its wall-clock bound")
            signal.signal(signal.SIGALRM, _alarm)
            signal.alarm(420)
            try:
                from oracle import ref_runtime as RR
                ref = RR.load_reference() if RR.available() else None
                if ref is not None:
                    result["cpu_baseline"] = cpu_baseline_reference(ref, unet_sd, vae_sd, ucfg, vcfg)
                else:
                    result["cpu_baseline"] = cpu_baseline(unet_sd, full)
            except TimeoutError as e:
                result["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": None, "kind": "port",
                                          "sample": f"not measured: {e}"}
            finally:
                signal.alarm(0)
        if world == 1 and not args.no_other_configs:
            log("serving leg: two batch-1 pipelines in flight on this GPU (informational; not the headline's protocol)")
            try:
                result["serving_two_in_flight"] = two_in_flight_leg(unet, vae, dev, args.denoise_steps, hw, value)
                log(f"serving: {result['serving_two_in_flight']['images_per_s']:.3f} images/s with two requests in flight "
                    f"({result['serving_two_in_flight']['vs_one_at_a_time']:.3f} x one at a time)")
            except Exception as e:  # a side leg must not cost the line
                result["serving_two_in_flight"] = {"images_per_s": None, "error": f"{type(e).__name__}: {e}"}
                log(f"serving leg failed: {result['serving_two_in_flight']['error']}")
            del pipe, unet, vae
            torch.cuda.empty_cache()
            result["other_configs"] = other_configs_leg(dev, args)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if _pg():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
