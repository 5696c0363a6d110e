#!/usr/bin/env python
"""TEST / BASELINE INFRASTRUCTURE ONLY -- packs the reference package itself so that it can travel to the GPU box.

The reference (huggingface/diffusers, /root/reference, v0.40.0.dev0) is pure Python: there is nothing to compile, but
`/root/reference` does not exist where `bench.py` and the `-m gpu` tests run.  This recipe zips `src/diffusers/**/*.py` where
it lies into ONE importable archive, `oracle/_ref/diffusers_ref.zip` (zipimport; `oracle/_ref/` is git-ignored, so no
reference source ever enters the history, and it is not gpurun-ignored, so the archive ships with the built `.so`).

Who may use it (checked by tests/test_bench_contract.py): `bench.py`'s `torch_rocm_baseline` / `parity` / `cpu_baseline` legs and
`tests/` -- as the thing compared AGAINST (stock diffusers on PyTorch-ROCm: the north star's denominator), never by
`diffusers_amd/`.  `oracle.ref_runtime.load_reference()` returns the imported package or None when the archive is absent.

    python oracle/build_ref.py            # (re)build if /root/reference is present; no-op otherwise
"""
from __future__ import annotations

import sys
import zipfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
SRC = Path("/root/reference/src/diffusers")
OUT = HERE / "_ref" / "diffusers_ref.zip"
EXPECT_VERSION = "0.40.0.dev0"


def reference_version(src: Path = SRC) -> str:
    for line in (src / "__init__.py").read_text().splitlines():
        if line.startswith("__version__"):
            return line.split("=", 1)[1].strip().strip("\"'")
    raise RuntimeError(f"no __version__ in {src / '__init__.py'}")


def build(force: bool = False) -> Path | None:
    """Returns the archive path, or None when the reference tree is not on this machine (the GPU box: use what shipped)."""
    if not SRC.exists():
        return OUT if OUT.exists() else None
    ver = reference_version()
    if ver != EXPECT_VERSION:
        raise RuntimeError(f"reference at {SRC} is version {ver}, this repository was built against {EXPECT_VERSION}")
    files = sorted(p for p in SRC.rglob("*.py") if "__pycache__" not in p.parts)
    stamp = f"{ver}:{len(files)}:{max(int(p.stat().st_mtime) for p in files)}"
    tag = OUT.with_suffix(".stamp")
    if not force and OUT.exists() and tag.exists() and tag.read_text() == stamp:
        return OUT
    OUT.parent.mkdir(parents=True, exist_ok=True)
    with zipfile.ZipFile(OUT, "w", compression=zipfile.ZIP_DEFLATED, compresslevel=6) as z:
        for p in files:
            z.write(p, arcname=str(Path("diffusers") / p.relative_to(SRC)))
    tag.write_text(stamp)
    return OUT


if __name__ == "__main__":
    out = build(force="--force" in sys.argv)
    print(out if out else "reference tree not present and no archive shipped")
