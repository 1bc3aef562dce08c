"""Builds fitclip_amd/csrc/libfitclip_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m fitclip_amd.build [--force] [--save-temps]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
REPO = CSRC.parent.parent
LIB = CSRC / "libfitclip_hip.so"
SOURCES = ["api.hip", "gemm.hip", "attention.hip", "rowops.hip", "score.hip"]
HEADERS = [CSRC / "common.h", CSRC / "gemm_kernel.h", REPO / "include" / "fitclip_hip.h"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, save_temps: bool = False, verbose: bool = True) -> Path:
    hipcc = _hipcc()
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    flags = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
             "-fvisibility=hidden", "-DFITCLIP_BUILD"]

    def compile_one(src: str) -> Path:
        obj = objdir / (src.replace(".hip", ".o"))
        if force or _stale(obj, [CSRC / src, *HEADERS]):
            cmd = [hipcc, *flags, "-c", str(CSRC / src), "-o", str(obj)]
            if save_temps:
                cmd += ["-save-temps=obj"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=str(objdir))
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
    print(LIB)
