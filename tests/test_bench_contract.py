"""Static guard on the driver contract of bench.py / __graft_entry__.py (they only run on an MI355X): the JSON line's keys,
the roofline / cpu_baseline objects, the CLI flags and the two entry points are checked from the source AST, and the last
bench line recorded under profiles/ is checked for the same schema."""
import ast
import json
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LINE_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config"}
ROOFLINE_KEYS = {"bound", "achieved", "peak", "unit", "frac", "traffic"}
CPU_KEYS = {"value", "unit", "cores", "kind", "sample"}


def _dict_literals(tree):
    for node in ast.walk(tree):
        if isinstance(node, ast.Dict):
            keys = {k.value for k in node.keys if isinstance(k, ast.Constant) and isinstance(k.value, str)}
            yield keys


def test_bench_source_emits_the_contract_keys():
    src = (ROOT / "bench.py").read_text()
    tree = ast.parse(src)
    dicts = list(_dict_literals(tree))
    assert any(LINE_KEYS <= d for d in dicts), "bench.py: the result line lost a contract key"
    assert any(ROOFLINE_KEYS <= d for d in dicts), "bench.py: roofline object incomplete"
    assert any(CPU_KEYS <= d for d in dicts), "bench.py: cpu_baseline object incomplete"
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in src
    assert "json.dumps(result)" in src and "max_over_ranks" in src and "barrier()" in src
    # the oracle (restatement AND the reference archive, oracle/ref_runtime.py) is test / baseline infrastructure: bench.py may
    # import it only inside the legs that run it as the thing compared against, never at module level or in the timed path
    allowed = {"cpu_baseline", "cpu_baseline_reference", "reference_pipeline_on_device", "reference_package_on_device",
               "reference_legs", "main", "_other_cpu_baseline", "dropin_leg"}     # baseline / checker legs, all outside the timed region
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        for node in ast.walk(fn):
            mod = node.module if isinstance(node, ast.ImportFrom) else None
            names = [a.name for a in node.names] if isinstance(node, ast.Import) else []
            if (mod and mod.split(".")[0] == "oracle") or any(n.split(".")[0] == "oracle" for n in names):
                assert fn.name in allowed, f"bench.py: {fn.name}() imports the oracle"
    for node in tree.body:
        assert not (isinstance(node, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(node)), "module-level oracle import"
    # in main() the only oracle use is the cpu_baseline dispatch (after the timed region)
    main_src = src.split("def main(")[1]
    assert main_src.index("from oracle") > main_src.index("timed region done")


def test_graft_entry_has_build_and_smoke():
    tree = ast.parse((ROOT / "__graft_entry__.py").read_text())
    names = {n.name for n in tree.body if isinstance(n, ast.FunctionDef)}
    assert {"build", "smoke"} <= names


def test_product_package_never_imports_the_oracle_or_test_stand_ins():
    for f in (ROOT / "diffusers_amd").glob("*.py"):
        src = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+(oracle|ops_emulation|tests)\b", src, re.M), f"{f.name} imports test infrastructure"


def test_recorded_bench_line_has_the_schema():
    txt = (ROOT / "profiles" / "README.md").read_text()
    assert "0.659 images/s" in txt
    line = ROOT / "gpurun_out" / "bench.json"
    if not line.exists():          # scratch directory: present in the build container only
        return
    rec = json.loads(line.read_text().strip().splitlines()[-1])
    assert LINE_KEYS <= set(rec) and ROOFLINE_KEYS <= set(rec["roofline"])
    if "cpu_baseline" in rec:       # absent when the scratch line came from a --no-cpu-baseline experiment run
        assert CPU_KEYS <= set(rec["cpu_baseline"])
    assert rec["higher_is_better"] is True and rec["scaling"] == "weak" and rec["n_gpus"] == 1
    assert 0 < rec["roofline"]["frac"] < 1 and rec["roofline"]["bound"] in ("mfma", "hbm")


def test_reference_archive_recipe(tmp_path):
    """oracle/build_ref.py packs the reference package (where /root/reference exists) into ONE importable archive under the
    git-ignored oracle/_ref/; oracle/ref_runtime.py imports exactly that version from it."""
    import subprocess
    import sys
    gi = (ROOT / ".gitignore").read_text()
    assert "oracle/_ref/" in gi
    gri = ROOT / ".gpurunignore"
    assert not gri.exists() or "oracle/_ref" not in gri.read_text(), "the archive must ship to the GPU box"
    sys.path.insert(0, str(ROOT))
    from oracle import build_ref
    out = build_ref.build()
    if out is None:                      # neither the reference tree nor a shipped archive: nothing to check here
        return
    assert out.exists() and out.suffix == ".zip"
    code = ("import sys; sys.path.insert(0, %r); from oracle import ref_runtime as RR; m = RR.load_reference(); "
            "print(m.__version__, m.__file__)" % str(ROOT))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    ver, where = r.stdout.strip().splitlines()[-1].split()
    assert ver == build_ref.EXPECT_VERSION and "diffusers_ref.zip" in where


def test_live_traffic_leg_never_costs_the_line(monkeypatch):
    """bench.live_traffic (roofline.traffic measured inside the run: two rocprofv3 --pmc subprocess passes) returns (None, reason)
    instead of raising when the profiler is missing or a pass fails -- the caller then replays profiles/sdxl_traffic.json."""
    import shutil
    import subprocess
    import sys
    sys.path.insert(0, str(ROOT))
    import bench
    monkeypatch.setattr(shutil, "which", lambda name: None)
    real_exists = Path.exists
    monkeypatch.setattr(Path, "exists", lambda self: False if str(self).endswith("rocprofv3") else real_exists(self))
    tr, why = bench.live_traffic(1.0)
    assert tr is None and "rocprofv3 not found" in why
    monkeypatch.undo()
    # a pass that fails (here: the stand-in profiler exits non-zero) is reported, not raised
    monkeypatch.setattr(shutil, "which", lambda name: "/bin/false")
    calls = []
    real_run = subprocess.run

    def fake_run(cmd, **kw):
        calls.append(cmd)
        return real_run(["/bin/false"], capture_output=True, text=True)
    monkeypatch.setattr(subprocess, "run", fake_run)
    tr, why = bench.live_traffic(1.0)
    assert tr is None and "FETCH_SIZE pass failed" in why and len(calls) == 1
    assert "--pmc" in calls[0] and "--kernel-trace" in calls[0] and not any(f in calls[0] for f in ("--sys-trace", "-s", "--hip-trace"))
