"""The RCCL path on the hardware that is available (ONE MI355X): the `nccl` process group initialises, and every collective the
8-GPU job issues (one broadcast per input tensor, barrier, MAX all-reduce, all_gather of the per-rank rates, the image gather) runs
through the communicator with one rank.  Then bench.py itself exactly as the driver launches it for N > 1
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`)
with N = 1.  A scaling curve needs 8 GPUs and is NOT measured here; these tests make the first 8-GPU run boring."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(script_and_args, timeout=900):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_and_args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_rccl_collectives_with_one_rank():
    rec = _launch([str(ROOT / "tests" / "dist_gpu_worker.py")])
    print(f"[rccl] {rec}")
    assert rec["backend"] == "nccl" and rec["world"] == 1
    assert rec["broadcast_ok"] and rec["gather_ok"] and rec["max"] == 1.25


def test_bench_under_the_drivers_launcher_on_the_gpu():
    rec = _launch([str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--tiny", "--denoise-steps", "4"])
    cfg = rec["config"]
    print(f"[rccl] bench --tiny under torch.distributed.run: {rec['value']:.2f} images/s, process_group {cfg['process_group']}, "
          f"hip_graph {cfg['hip_graph']}, tuned_live {cfg['tuned_live']}")
    assert rec["n_gpus"] == 1 and cfg["rccl_ranks"] == 1 and cfg["process_group"] == "nccl"
    assert cfg["hip_graph"] is True and cfg["output_finite"] and len(cfg["images_per_s_per_rank"]) == 1
    assert rec["scaling"] == "weak" and rec["value"] > 0
