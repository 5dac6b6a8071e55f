"""Compile the gfx950 C-ABI library in-tree (hipcc cross-compiles without a GPU).

    python medtok_amd/csrc/build.py [--force]

Output: medtok_amd/csrc/libmedtok_vq.so (git-ignored; travels to the GPU box
with the gpurun snapshot).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
SRC = HERE / "medtok_vq.hip"
OUT = HERE / "libmedtok_vq.so"
HEADER = HERE.parents[1] / "include" / "medtok_vq.h"

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # every fused multiply-add in the kernels is written as fmaf(); nothing else may contract
    "-ffp-contract=off",
    "-Wall", "-Wno-unused-function",
]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


def build(force: bool = False, verbose: bool = False) -> Path:
    # every source the translation unit includes (medtok_vq.hip pulls in the *.h next to it)
    deps = [SRC, HEADER, Path(__file__), *HERE.glob("*.h")]
    newest = max(p.stat().st_mtime for p in deps)
    if not force and OUT.exists() and OUT.stat().st_mtime >= newest:
        return OUT
    # (compiled next to the target and renamed into place: a reader -- a running process, a snapshot of the tree -- never sees a
    # half-written library)
    tmp = OUT.with_name(OUT.name + f".tmp{os.getpid()}")
    cmd = [hipcc(), *FLAGS, str(SRC), "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, OUT)
    finally:
        if tmp.exists():
            tmp.unlink()
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
