"""Build the gfx950 HIP kernels into the in-tree C-ABI shared library ``diffusers_amd/_C/libdiffusers_amd.so``.

hipcc cross-compiles for gfx950 without a GPU, so this runs in CI containers as well as on MI355X boxes.  The
library has no Python / torch dependency: it is a plain ``extern "C"`` surface declared in ``include/diffusers_amd.h``.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
ROOT = PKG_DIR.parent
CSRC = PKG_DIR / "csrc"
OUT_DIR = PKG_DIR / "_C"
LIB_PATH = OUT_DIR / "libdiffusers_amd.so"
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the diffusers_amd kernels can only be built with the ROCm toolchain")


def _sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _fingerprint() -> str:
    h = hashlib.sha256()
    for p in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cuh")) + [ROOT / "include" / "diffusers_amd.h"]):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def _unit_fingerprint(src: Path, flags) -> str:
    import re
    seen, todo = {}, [src]
    while todo:
        f = todo.pop()
        if f in seen or not f.exists():
            continue
        text = f.read_bytes()
        seen[f] = text
        for name in re.findall(rb'#include\s+"([^"]+)"', text):
            for base in (f.parent, CSRC, ROOT / "include"):
                cand = base / name.decode()
                if cand.exists():
                    todo.append(cand)
                    break
    h = hashlib.sha256(" ".join(flags).encode())
    for f in sorted(seen):
        h.update(f.name.encode())
        h.update(seen[f])
    return h.hexdigest()


def build_extension(force: bool = False, verbose: bool = False) -> Path:
    """Compile every ``csrc/*.hip`` for gfx950 and link ``libdiffusers_amd.so``; returns the library path."""
    OUT_DIR.mkdir(exist_ok=True)
    stamp = OUT_DIR / "build.stamp"
    fp = _fingerprint()
    if not force and LIB_PATH.exists() and stamp.exists() and stamp.read_text().strip() == fp:
        return LIB_PATH
    hipcc = _hipcc()
    flags = [
        f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
        f"-I{ROOT / 'include'}", f"-I{CSRC}",
    ]
    objs = []

    def compile_one(src: Path) -> Path:
        # per-object stamp: the source, the headers it includes (transitively) and the flags -- an unchanged translation unit is
        # not recompiled (the two GEMM families take minutes)
        obj = OUT_DIR / (src.stem + ".o")
        ostamp = OUT_DIR / (src.stem + ".o.stamp")
        ofp = _unit_fingerprint(src, flags)
        if not force and obj.exists() and ostamp.exists() and ostamp.read_text().strip() == ofp:
            return obj
        cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stdout}\n{r.stderr}")
        ostamp.write_text(ofp)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, _sources()))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB_PATH), *map(str, objs)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(fp)
    return LIB_PATH


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose=True))
