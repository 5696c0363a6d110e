"""CPU / gloo, world_size 2: the batch-parallel sharding of diffusers_amd.distributed (the N > 1 path of bench.py).

The data path has no collective: prompts are dealt round-robin, rank 0 broadcasts the text embeddings once, every rank
works on its shard, images are gathered back in prompt order.  Mirrors the reference's recipe
(docs/source/en/training/distributed_inference.md:29-108), which has no library code or tests of its own."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_prompts, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from diffusers_amd import distributed as D
    try:
        before = sorted(os.sched_getaffinity(0))
        r, w, _ = D.init_from_env(backend="gloo")
        assert (r, w) == (rank, world)
        # every rank pinned itself to ITS share of the host's cores (no GPU here: the equal contiguous split)
        after = sorted(os.sched_getaffinity(0))
        assert after == D.cpu_share(rank, world, before, [-1] * world, {}), (before, after)
        g = torch.Generator().manual_seed(1234)
        full = {"prompt_embeds": torch.randn((n_prompts, 7, 16), generator=g),
                "pooled": torch.randn((n_prompts, 8), generator=g)}
        # only rank 0 owns the prompts; the others receive them through ONE broadcast per tensor
        inputs = {k: (v.clone() if rank == 0 else torch.zeros_like(v)) for k, v in full.items()}
        D.broadcast_tensors(inputs, src=0)
        for k in full:
            assert torch.equal(inputs[k], full[k]), f"rank {rank}: broadcast of {k} differs"
        idx = D.shard_indices(n_prompts, rank, world)
        mine = D.select_shard(inputs, idx)
        assert mine["prompt_embeds"].shape[0] == len(idx)
        # stand-in for the per-prompt pipeline: something that depends on the prompt only
        local_img = mine["prompt_embeds"].sum(dim=(1, 2)).view(-1, 1, 1, 1) + torch.zeros((len(idx), 3, 4, 4))
        out = D.gather_images(local_img, n_prompts, dst=0)
        t = D.max_over_ranks(float(rank + 1), torch.device("cpu"))
        assert t == float(world)
        if rank == 0:
            want = full["prompt_embeds"].sum(dim=(1, 2)).view(-1, 1, 1, 1) + torch.zeros((n_prompts, 3, 4, 4))
            assert torch.equal(out, want), "gathered images are not in prompt order"
        else:
            assert out is None
        dist.barrier()
        q.put((rank, "ok", idx))
    except Exception as e:  # surfaced in the parent
        q.put((rank, f"fail: {e!r}", None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("n_prompts", [2, 5, 8])
def test_two_rank_sharding_broadcast_gather(n_prompts):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_prompts, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, idx in res:
        assert status == "ok", f"rank {rank}: {status}"
    shards = {rank: idx for rank, status, idx in res}
    assert sorted(shards[0] + shards[1]) == list(range(n_prompts))  # every prompt exactly once
    assert abs(len(shards[0]) - len(shards[1])) <= 1                # balanced


def test_shard_indices_properties():
    from diffusers_amd import distributed as D
    for n in (0, 1, 7, 8, 64):
        for world in (1, 2, 4, 8):
            parts = [D.shard_indices(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        D.shard_indices(4, 2, 2)


def test_single_process_paths_are_identity():
    from diffusers_amd import distributed as D
    t = {"a": torch.arange(6.0).view(3, 2)}
    assert D.broadcast_tensors(t) is t
    assert D.gather_images(t["a"], 3) is t["a"]
    assert D.max_over_ranks(3.5, torch.device("cpu")) == 3.5


def test_bench_main_world2_gloo(tmp_path):
    """The N > 1 path of bench.py END TO END as the driver launches it (python -m torch.distributed.run --nproc-per-node 2
    bench.py --gpus 2 ...), on CPU: gloo, kernel stand-ins (tests/bench_cpu_worker.py).  Rank 0 prints exactly one JSON line
    with n_gpus = 2, the whole-job rate over both ranks, weak scaling and one rate per rank."""
    import json
    import subprocess
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "bench_cpu_worker.py"), "--gpus", "2", "--steps", "2", "--warmup",
           "1", "--tiny", "--no-graph", "--denoise-steps", "2"]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}:\n{r.stdout[-2000:]}"
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["config"]["rccl_ranks"] == 2 and len(rec["config"]["images_per_s_per_rank"]) == 2
    assert rec["config"]["global_batch"] == 2 and rec["config"]["hip_graph"] is False
    # whole-job rate = 2 ranks x 2 images / max-over-ranks time; it cannot exceed the sum of the per-rank rates
    assert 0 < rec["value"] <= sum(rec["config"]["images_per_s_per_rank"]) * 1.001
    assert abs(rec["ms_per_step"] * rec["value"] - 2 * 1000.0) < 1e-6 * 2000.0
    assert rec["config"]["tuned_live"] == 0      # reported so that a tuning pass on 8 ranks at once cannot hide in a scaling run


def test_bench_main_world8_gloo(tmp_path):
    """The launch the driver's 8-GPU scaling run will make, on CPU stand-ins (no 8-GPU node was available to any
    round, so the first real run must not fail on plumbing): eight ranks under torch.distributed.run, one prompt per rank (seeds
    1234 ... 1241 through the ONE broadcast from rank 0), rccl_ranks == 8, eight per-rank rates, no rank tuning GEMM variants live."""
    import json
    import subprocess
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "bench_cpu_worker.py"), "--gpus", "8", "--steps", "1", "--warmup",
           "1", "--tiny", "--no-graph", "--denoise-steps", "2"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}:\n{r.stdout[-2000:]}"
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["steps"] == 1 and rec["scaling"] == "weak"
    assert rec["config"]["rccl_ranks"] == 8 and len(rec["config"]["images_per_s_per_rank"]) == 8
    assert rec["config"]["global_batch"] == 8 and rec["config"]["tuned_live"] == 0
    assert 0 < rec["value"] <= sum(rec["config"]["images_per_s_per_rank"]) * 1.001
    assert "other_configs" not in rec and "dropin" not in rec          # the side legs belong to rank 0 at N = 1 only
    # every rank built its own prompt from the broadcast: the synthetic inputs of prompt i are seeded 1234 + i
    import bench
    inp = bench.synth_inputs(8, True, "cpu")
    assert inp["latents"].shape[0] == 8 and not torch.equal(inp["latents"][0], inp["latents"][7])


@pytest.mark.parametrize("config,unit", [("sd15", "images/s"), ("ddpm", "images/s")])
def test_bench_other_configs_world2_gloo(tmp_path, config, unit):
    """`bench.py --config sd15 | ddpm` (the other BASELINE configs under the driver contract), world size 2
    on CPU stand-ins: one JSON line with the contract keys, the config's own metric and n_gpus = 2."""
    import json
    import subprocess
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "bench_cpu_worker.py"), "--gpus", "2", "--steps", "1", "--warmup",
           "1", "--tiny", "--no-graph", "--denoise-steps", "2", "--config", config]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["n_gpus"] == 2 and rec["unit"] == unit and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["denoise_steps"] == 2 and rec["config"]["output_finite"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rec["roofline"])


def test_shipped_tuning_table_covers_every_sdxl_shape():
    """at N = 8 all ranks warm up at once; if the shipped table missed a shape every rank would time ~80
    kernel variants for it simultaneously -- the likeliest way to miss 0.9x linear scaling.  The keys below were recorded by a
    full-size bench run that started from an EMPTY table (tools/gpu_r3.sh retune): every one must be in the shipped table, so a
    bench warm-up never enters da_gemm_tune (bench.py reports `config.tuned_live`, 0 then)."""
    import json
    keys = json.loads((ROOT / "tests" / "golden" / "sdxl_gemm_shape_keys.json").read_text())["keys"]
    table = json.loads((ROOT / "diffusers_amd" / "tuned" / "gfx950.json").read_text())
    missing = [k for k in keys if k not in table["entries"]]
    assert len(keys) >= 60 and not missing, f"shapes the shipped table does not hold: {missing[:5]}"
    from diffusers_amd import _lib as L
    assert table["tiles"] == list(L.TILE_NAMES), "table written for another tile enumeration"
    for k, v in table["entries"].items():
        assert 1 <= v[0] < len(L.TILE_NAMES) and 0 <= v[1] <= L.STAGE_PINGPONG3, (k, v)


def test_bench_self_launch_builds_the_torchrun_command(monkeypatch):
    """`python bench.py --gpus 8` without a launcher re-executes itself under torch.distributed.run with 8 ranks on
    127.0.0.1 (VERDICT r1 missing #4); with WORLD_SIZE already set (the driver's launch) it must NOT spawn again."""
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ei:
        bench.main(["--gpus", "8", "--steps", "2", "--warmup", "1"])
    assert ei.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "8", "--steps", "2", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert Path(cmd[cmd.index("--master-port") + 2]).name == "bench.py"


def test_bench_main_one_launched_rank_runs_the_collectives(tmp_path):
    """ONE rank under torch.distributed.run (what `tests/test_distributed_gpu.py` does on the single-GPU box with RCCL): the process
    group IS initialised and the broadcast / barrier / max-over-ranks / all_gather run through it instead of taking the
    lone-process early-outs -- `config.process_group` names the backend (gloo here, nccl on the GPU)."""
    import json
    import subprocess
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "bench_cpu_worker.py"), "--gpus", "1", "--steps", "1", "--warmup",
           "1", "--tiny", "--no-graph", "--denoise-steps", "2"]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["config"]["rccl_ranks"] == 1 and rec["config"]["process_group"] == "gloo"
    assert len(rec["config"]["images_per_s_per_rank"]) == 1 and rec["config"]["tuned_live"] == 0


def test_cpu_share_policy():
    """distributed.cpu_share: ranks on the cores of their GPU's NUMA node, peers of a node split it in rank order, every rank
    computes the same disjoint partition on its own; unknown topology -> equal contiguous runs of the allowed cores."""
    from diffusers_amd import distributed as D
    allowed = list(range(128))
    node_cpus = {0: list(range(0, 64)), 1: list(range(64, 128))}
    gpu_nodes = [0, 0, 0, 0, 1, 1, 1, 1]                                   # the usual 8-GPU node: four GPUs per socket
    shares = [D.cpu_share(r, 8, allowed, gpu_nodes, node_cpus) for r in range(8)]
    assert shares[0] == list(range(0, 16)) and shares[3] == list(range(48, 64)) and shares[4] == list(range(64, 80))
    assert sorted(c for sh in shares for c in sh) == allowed              # disjoint, nothing left over
    # a cpuset that hides half of socket 1 (container limits): its ranks split what is left
    lim = [c for c in allowed if c < 96]
    sh = [D.cpu_share(r, 8, lim, gpu_nodes, node_cpus) for r in range(8)]
    assert sh[4] == list(range(64, 72)) and sh[7] == list(range(88, 96)) and all(set(x) <= set(lim) for x in sh)
    # unknown topology
    flat = [D.cpu_share(r, 8, allowed, [-1] * 8, {}) for r in range(8)]
    assert flat[2] == list(range(32, 48)) and sorted(c for x in flat for c in x) == allowed
    # one GPU's node unknown: that rank takes its flat run, the others their node's
    mixed = D.cpu_share(1, 8, allowed, [0, -1, 0, 0, 1, 1, 1, 1], node_cpus)
    assert mixed == list(range(16, 32))
    # fewer cores than ranks: ranks share, never an empty set
    assert all(len(D.cpu_share(r, 8, [0, 1, 2], [-1] * 8, {})) == 1 for r in range(8))
    with pytest.raises(ValueError):
        D.cpu_share(8, 8, allowed, gpu_nodes, node_cpus)
    assert D._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def _solo(port_env, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.pop("MASTER_PORT", None)
    if port_env:
        os.environ["MASTER_PORT"] = str(port_env)
    from diffusers_amd import distributed as D
    try:
        D.init_from_env(backend="gloo")
        q.put(("ok", int(os.environ["MASTER_PORT"]), D.max_over_ranks(3.0, torch.device("cpu"))))
        import time
        time.sleep(2.0)          # hold the store while the sibling job starts
    except Exception as e:
        q.put((f"fail: {e!r}", None, None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_two_independent_single_rank_jobs_do_not_collide():
    """SLURM / k8s export RANK=0 WORLD_SIZE=1 without a MASTER_PORT; two such jobs on one host must both come up (each
    one-rank group rendezvouses on a free port of its own, not on the shared default 29500); a launcher-provided port is kept."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_solo, args=(None, q)) for _ in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[0] == "ok" and r[2] == 3.0 for r in res), res
    assert res[0][1] != res[1][1] and 29500 not in (res[0][1], res[1][1])
    port = _free_port()
    p = ctx.Process(target=_solo, args=(port, q))
    p.start()
    got = q.get(timeout=120)
    p.join(timeout=60)
    assert got[0] == "ok" and got[1] == port
