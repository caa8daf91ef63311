"""Build libmstg_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

The shared object lands next to this file so that it travels with the repository snapshot to the GPU box and
shows up as a loaded in-tree library.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "..", "include"))
LIB = os.path.join(HERE, "libmstg_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("MSTG_HIPCC_FLAGS", "").split()  # e.g. -DMSTG_STAMPS for tools/diag_stamps.py
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", *extra, "-I", INCLUDE, "-I", CSRC,
           *sources(), "-o", LIB]
    if verbose:
        print("[mstg_hip.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
