"""Guard against a defect class found in round 6 by reading the ISA: a load written as ``cond ? *p : 0`` (or inside ``if (cond)``) gets
its own exec-masked branch and its own ``s_waitcnt vmcnt(0)``, so loads the source declares "in flight" are in fact serialised round
trips (``gn_apply_kernel``'s fold of 16 partials, the NCH row chunks of ``layernorm_kernel`` / ``rmsnorm_rows_kernel`` /
``rmsnorm_rope_kernel``, the 72 input loads of ``conv_thin_in4_kernel``: each was most of its kernel's time).  The fix is an
unconditional load from a clamped address plus a select.  This test compiles ``csrc/norm.hip`` for gfx950 to assembly (hipcc
cross-compiles without a GPU) and fails if any kernel shows a run of three or more vector-memory loads that are each followed at once
by ``s_waitcnt vmcnt(0)``."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _serialised_runs(asm: str):
    best, kern, run, prev_load = {}, None, 0, -10
    for i, line in enumerate(asm.split("\n")):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern, run = m.group(1), 0
        if kern is None:
            continue
        if "global_load" in line or ("buffer_load" in line and " lds" not in line):
            prev_load = i
        if "s_waitcnt vmcnt(0)" in line:
            if i - prev_load <= 2:
                run += 1
                best[kern] = max(best.get(kern, 0), run)
            else:
                run = 0
        if "s_endpgm" in line:
            kern = None
    return best


def test_scanner_sees_the_pattern():
    asm = "_Zfoo:\n" + "global_load_dwordx2 v[0:1], v[2:3], off\ns_waitcnt vmcnt(0)\nv_add_f32 v0, v0, v1\n" * 4 + "s_endpgm\n" \
          "_Zbar:\n" + "global_load_dwordx2 v[0:1], v[2:3], off\n" * 4 + "s_waitcnt vmcnt(0)\ns_endpgm\n"
    runs = _serialised_runs(asm)
    assert runs.get("_Zfoo") == 4 and runs.get("_Zbar", 0) <= 1


# kernels whose remaining runs are cold or once-per-launch paths: the per-output-channel bias scalars of the thin-output conv (3 .. 16 per
# LAUNCH), the generic thin-input kernel's runtime tap loops (1 x 1 / Cin > 4 cases), the unaligned-bias fallback of the four-pixel
# conv_in, the postprocess kernels' four channel planes
ALLOW = {"misc": ("conv_thin_out_kernel", "conv_thin_in_kernel", "conv_thin_in4_kernel", "image_postprocess_kernel")}


@pytest.mark.parametrize("unit", ["norm", "misc"])
def test_small_kernels_keep_their_loads_in_flight(unit, tmp_path):
    hipcc = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if Path("/opt/rocm/bin/hipcc").exists() else None)
    if hipcc is None:
        pytest.skip("hipcc not available")
    out = tmp_path / f"{unit}.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", f"-I{ROOT / 'include'}",
                        f"-I{ROOT / 'diffusers_amd' / 'csrc'}", "-S", "--cuda-device-only", str(ROOT / "diffusers_amd" / "csrc" / f"{unit}.hip"),
                        "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-400:]
    runs = _serialised_runs(out.read_text())
    bad = {k: v for k, v in runs.items() if v >= 3 and not any(a in k for a in ALLOW.get(unit, ()))}
    assert not bad, f"serialised loads (load immediately followed by s_waitcnt vmcnt(0), three or more in a row): {bad}"
    if unit == "misc":      # the four-pixel conv_in: only its unaligned-bias fallback (eight scalars + the staging loop's one) may remain
        assert max(v for k, v in runs.items() if "conv_thin_in4_kernel" in k) <= 9
