"""Worker of tests/test_distributed_cpu.py::test_bench_main_world2_gloo: runs bench.main() -- the driver's N > 1 entry,
launched by torch.distributed.run exactly as the driver launches it -- on CPU: the kernels are replaced by the torch
stand-ins of tests/ops_emulation.py and the device plumbing of bench.py is pointed at the CPU, so what is exercised is the
rank / world handling, the one broadcast, the barrier + max-over-ranks timing and the single JSON line of rank 0.
TEST INFRASTRUCTURE: nothing here is reachable from bench.py itself."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import bench  # noqa: E402
import ops_emulation  # noqa: E402
from diffusers_amd import ops  # noqa: E402


class _Patch:
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


ops_emulation.install(_Patch(), ops)
ops.TUNING = False
bench._device = lambda local: torch.device("cpu")
bench._sync = lambda: None
bench.main(sys.argv[1:])
