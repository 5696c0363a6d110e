"""Launch plans on hardware (include/diffusers_amd.h "launch plans", csrc/plan.hip, diffusers_amd/plan.py): one denoising step
of an SDXL-shaped U-Net (pipeline_stable_diffusion_xl.py:1186-1250) + one VAE decode (:1283-1299), recorded once and replayed by
da_plan_launch with no Python between the launches -- bit-identical to the Python-driven step; and the same plan written to a
file and run by the C++ program examples/abi_demo.cpp in a process with no Python and no torch."""
import shutil
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _recorded():
    from diffusers_amd import plan as P
    from tools.make_plan_demo import build
    st = build()
    st["reset"]()
    st["run"]()                                      # warm-up: live tuning / lazy caches happen here, not in the recording
    st["reset"]()
    pl, image = P.record(st["run"], keep=st["keep"])  # eager, Python-driven: the reference result of step 1
    torch.cuda.synchronize()
    return P, st, pl, image


def test_plan_replays_a_unet_step_and_a_decode_bit_identically():
    P, st, pl, image = _recorded()
    lat = st["latents"]
    assert pl.foreign_ops == [] and len(pl) == len(pl.names) > 50
    kinds = set(pl.names)
    assert {"da_gemm_bf16", "da_attention_bf16", "da_groupnorm_nhwc_bf16", "da_euler_step", "da_advance_step"} <= kinds
    want1 = (lat.clone(), image.clone())
    image2 = st["run"]()                             # Python-driven step 2 (the step counter advanced on the device)
    torch.cuda.synchronize()
    want2 = (lat.clone(), image2.clone())
    assert not torch.equal(want1[0], want2[0])
    # replay from the same start state; the outputs are poisoned first so that a launch that did not run shows
    st["reset"]()
    image.fill_(float("nan"))
    pl.launch()
    torch.cuda.synchronize()
    assert torch.equal(lat, want1[0]) and torch.equal(image, want1[1])
    pl.launch()                                      # and the second step: the plan reads the device-side step counter
    torch.cuda.synchronize()
    assert torch.equal(lat, want2[0]) and torch.equal(image, want2[1])
    print(f"[plan] {len(pl)} launches replayed by da_plan_launch: latents and image bit-identical to the Python-driven steps")


def test_plan_file_runs_in_a_process_without_python(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("no hipcc on this box")
    P, st, pl, image = _recorded()
    st["reset"]()
    path = tmp_path / "step.daplan"
    info = pl.save(path, outputs=[st["latents"], image])
    # save() leaves the device state as it found it
    assert torch.equal(st["latents"], st["latents0"])
    exe = tmp_path / "abi_demo"
    libdir = ROOT / "diffusers_amd" / "_C"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", f"-I{ROOT / 'include'}", str(ROOT / "examples" / "abi_demo.cpp"),
                        f"-L{libdir}", "-ldiffusers_amd", f"-Wl,-rpath,{libdir}", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), str(path)], capture_output=True, text=True, timeout=300)
    print(f"[plan] {info}\n{r.stdout}")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("0 differ") == 2
    # the same file through the Python-side loader: fresh buffers holding the regions' contents, da_plan_relocate onto them, launch
    import ctypes as C
    from diffusers_amd import _lib as L
    doc = P.read_file(path)
    bufs = [torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda() for _, _, data in doc["regions"]]
    h, _ = P.create_from_file(path, new_bases=[b.data_ptr() for b in bufs])
    failed = C.c_int(-1)
    L.check(L.load().da_plan_launch(h, torch.cuda.current_stream().cuda_stream, C.byref(failed)), "da_plan_launch")
    torch.cuda.synchronize()
    L.load().da_plan_destroy(h)
    for ptr, want in doc["outputs"]:
        r = next(i for i, (b, n, _) in enumerate(doc["regions"]) if b <= ptr < b + n)
        off = ptr - doc["regions"][r][0]
        assert bytes(bufs[r][off:off + len(want)].cpu().numpy().tobytes()) == want
    # and the built-in checks of the program (direct calls + a plan built in C++)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "identical bytes" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_plan_launch_reports_the_failing_op():
    from diffusers_amd import _lib as L, plan as P
    lib = L.load()
    x = torch.zeros(64, 64, dtype=torch.bfloat16, device="cuda")
    rec = P.Recorder()
    good = L.GemmParams()
    good.A, good.W, good.C = x.data_ptr(), x.data_ptr(), torch.empty_like(x).data_ptr()
    good.M = good.N = good.K = good.lda = good.ldw = good.ldc = 64
    good.alpha = good.out_scale = 1.0
    bad = L.GemmParams.from_buffer_copy(good)
    bad.lda = 3                                       # not a multiple of 8: DA_ERR_INVALID
    import ctypes as C
    for p in (good, bad, good):
        rec.note("da_gemm_bf16", L.FN_IDS["da_gemm_bf16"], (C.byref(p), 0))
    pl = P.Plan(rec)
    with pytest.raises(RuntimeError, match=r"op 1: da_gemm_bf16"):
        pl.launch()
    assert lib.da_plan_op_count(pl._h) == 3


def _rand(shape, seed):
    return torch.randn(shape, generator=torch.Generator("cpu").manual_seed(seed)).to(torch.bfloat16).to("cuda")


def _pipelines():
    """(name, pipeline factory, call kwargs) of the five BASELINE families at their small test sizes."""
    from diffusers_amd import factory
    yield ("sdxl", lambda: factory.build_sdxl_pipeline(device="cuda", tiny=True, seed=0),
           dict(prompt_embeds=_rand((1, 77, 64), 1), negative_prompt_embeds=_rand((1, 77, 64), 2), pooled_prompt_embeds=_rand((1, 64), 3),
                negative_pooled_prompt_embeds=_rand((1, 64), 4), latents=_rand((1, 4, 16, 16), 5), num_inference_steps=4,
                guidance_scale=5.0, height=128, width=128, output_type="pt"))
    yield ("sd15", lambda: factory.build_sd15_pipeline(device="cuda", tiny=True, seed=0),
           dict(prompt_embeds=_rand((1, 7, 64), 1), negative_prompt_embeds=_rand((1, 7, 64), 2), latents=_rand((1, 4, 16, 16), 5),
                num_inference_steps=4, guidance_scale=7.5, height=32, width=32, output_type="raw"))
    yield ("flux", lambda: factory.build_flux_pipeline(device="cuda", tiny=True, seed=5),
           dict(prompt_embeds=_rand((1, 16, 64), 1), pooled_prompt_embeds=_rand((1, 64), 2), num_inference_steps=4, guidance_scale=0.0,
                height=64, width=64, max_sequence_length=16, output_type="raw", generator=lambda: torch.Generator("cpu").manual_seed(3)))
    yield ("wan", lambda: factory.build_wan_pipeline(device="cuda", tiny=True, seed=9),
           dict(prompt_embeds=_rand((1, 16, 64), 1), negative_prompt_embeds=_rand((1, 16, 64), 2), num_inference_steps=3,
                guidance_scale=5.0, height=64, width=64, num_frames=9, generator=lambda: torch.Generator("cpu").manual_seed(3)))
    yield ("ddpm", lambda: factory.build_ddpm_pipeline(device="cuda", tiny=True, seed=11),
           dict(batch_size=1, num_inference_steps=5, output_type="np", generator=lambda: torch.Generator("cpu").manual_seed(0)))


@pytest.mark.parametrize("name", ["sdxl", "sd15", "flux", "wan", "ddpm"])
def test_every_pipeline_replays_its_step_as_a_plan(name):
    """`use_graph="plan"`: the denoising step of each model family recorded once and replayed by da_plan_launch (Flux's
    da_rmsnorm_rope_bf16 brings the host-array arguments, Wan the fp32 / CFG sampler kernels, DDPM the noise table) -- the same
    image, bit for bit, as the captured HIP graph and as the eager loop."""
    import numpy as np
    from diffusers_amd import plan as P
    make, kw = next((m, k) for n, m, k in _pipelines() if n == name)
    pipe = make()

    def run(mode):
        k = {a: (b() if callable(b) else b.clone() if torch.is_tensor(b) else b) for a, b in kw.items()}
        out = pipe(use_graph=mode, **k).images
        return torch.from_numpy(out) if isinstance(out, np.ndarray) else out.clone()
    eager = run(False)
    graph = run(True)
    planned = run("plan")
    assert isinstance(pipe._graph, P.Plan) and pipe._graph.foreign_ops == [] and len(pipe._graph) > 20
    again = run("plan")                                       # second call: the recorded plan is reused on refreshed static inputs
    assert torch.equal(graph, eager) and torch.equal(planned, graph) and torch.equal(again, graph)
    print(f"[plan] {name}: {len(pipe._graph)} launches per step by da_plan_launch, image bit-identical to the HIP graph and to the eager loop")


@pytest.mark.parametrize("mode", [True, "plan"])
def test_step_callbacks_between_graph_and_plan_replays(mode):
    """callback_on_step_end (pipeline_stable_diffusion_xl.py:1239-1247) between the replays of the captured step: the latents a
    callback sees after each step, a replacement it hands back and an interrupt behave as in the eager loop, bit for bit."""
    make, kw = next((m, k) for n, m, k in _pipelines() if n == "sdxl")
    pipe = make()
    kw = dict(kw, output_type="latent")

    def run(use_graph, cb):
        k = {a: (b.clone() if torch.is_tensor(b) else b) for a, b in kw.items()}
        return pipe(use_graph=use_graph, callback_on_step_end=cb, **k).images.clone()

    def script(log):
        def cb(p, i, t, kwargs):
            log.append((i, float(t), kwargs["latents"].clone()))
            if i == 1:
                return {"latents": kwargs["latents"] * 0.5}
            if i == 2:
                p._interrupt = True
            return kwargs
        return cb
    want_log, got_log = [], []
    want = run(False, script(want_log))
    got = run(mode, script(got_log))
    assert len(want_log) == len(got_log) == 3 and pipe.scheduler.step_index == 3
    for a, b in zip(want_log, got_log):
        assert a[:2] == b[:2] and torch.equal(a[2], b[2])
    assert torch.equal(want, got)
    assert torch.equal(run(mode, None), run(False, None))        # and an un-hooked call afterwards runs all four steps again


def test_swapping_the_model_recaptures_the_step():
    """A captured step (graph or plan) points into one model's packed weights: after `pipe.unet = other` the next call must capture
    again, not replay the old model (the capture key carries the model's identity)."""
    from diffusers_amd import factory
    make, kw = next((m, k) for n, m, k in _pipelines() if n == "sdxl")
    pipe = make()
    other = factory.build_sdxl_pipeline(device="cuda", tiny=True, seed=7).unet

    def run(mode):
        k = {a: (b.clone() if torch.is_tensor(b) else b) for a, b in kw.items()}
        return pipe(use_graph=mode, **k).images.clone()
    for mode in (True, "plan"):
        first = run(mode)
        keep, pipe.unet = pipe.unet, other
        swapped = run(mode)
        assert not torch.equal(swapped, first) and torch.equal(swapped, run(False))
        pipe.unet = keep
        assert torch.equal(run(mode), first)
