import os
import sys
from pathlib import Path

# The CPU suite runs tiny models through the kernel stand-ins (torch ops on bf16 tensors of a few KB): with torch's default of one
# intra-op thread per core every such op pays a fork / join across 8 threads and the suite takes 15 minutes instead of 2.  Two threads,
# set before torch is imported so that the ranks the distributed tests spawn inherit it.  A GPU box keeps its defaults.
if not os.path.exists("/dev/kfd"):
    os.environ.setdefault("OMP_NUM_THREADS", "2")

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(GOLDEN / f"{name}.npz"))
    return load


def rel_rms(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12))


def max_abs(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.float().cpu() - b.float().cpu()).abs().max())


def assert_close_bf16(a: torch.Tensor, ref: torch.Tensor, what: str, rtol=1.6e-2, atol_rms=1.6e-2, rel_rms_max=None):
    """Reference-style bf16 tolerance (tests/models/testing_utils/attention.py:352-390 uses atol=rtol=1e-2 for bf16
    kernels): |a-ref| <= atol + rtol*|ref| with atol scaled to the tensor's rms, plus a bound on the relative rms."""
    a, ref = a.float().cpu(), ref.float().cpu()
    rms = float(ref.pow(2).mean().sqrt())
    err = (a - ref).abs()
    bound = atol_rms * rms + rtol * ref.abs()
    bad = int((err > bound).sum())
    rr = rel_rms(a, ref)
    print(f"[parity] {what}: rel_rms={rr:.3e} max_abs={float(err.max()):.3e} ref_rms={rms:.3e} violations={bad}/{err.numel()}")
    assert torch.isfinite(a).all(), f"{what}: non-finite output"
    assert bad == 0, f"{what}: {bad} elements outside bf16 tolerance (max err {float(err.max()):.4e}, rel_rms {rr:.3e})"
    if rel_rms_max is not None:
        assert rr <= rel_rms_max, f"{what}: rel_rms {rr:.3e} > {rel_rms_max}"
