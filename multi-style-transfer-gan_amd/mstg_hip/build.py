"""Build libmstg_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

The shared object lands next to this file so that it travels with the repository snapshot to the GPU box and
shows up as a loaded in-tree library.  hipcc cross-compiles without a GPU.  Every ``csrc/*.hip`` is compiled to its own
object (in parallel, only when it or a header changed) and the objects are linked into the library.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "..", "include"))
LIB = os.path.join(HERE, "libmstg_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))


def _obj(src: str) -> str:
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(p) > t for p in deps)


# Per-source compiler switches.  attention_reg.hip: the C = 64 backward runs one wave per SIMD with the whole 512-register file;
# left to its heuristic the compiler then puts EVERY MFMA accumulator into AGPRs (an accvgpr copy on each side of all VALU work on
# a chain's result) and spills ~90 VGPRs to scratch; forced to the VGPR form it uses the AGPRs as spill space and needs no scratch.
# The other kernels of the file stay below 256 registers and compile to the same code either way.
PER_SOURCE_FLAGS = {"attention_reg.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _flags() -> str:
    return " ".join(os.environ.get("MSTG_HIPCC_FLAGS", "").split())


def needs_build() -> bool:
    flags_file = os.path.join(OBJ, "flags.txt")
    if os.path.exists(flags_file) and open(flags_file).read() != _flags():
        return True  # built with other MSTG_HIPCC_FLAGS
    return _stale(LIB, sources() + headers())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("MSTG_HIPCC_FLAGS", "").split()  # e.g. -DMSTG_STAMPS for tools/diag_stamps.py
    os.makedirs(OBJ, exist_ok=True)
    flags_file = os.path.join(OBJ, "flags.txt")
    flags = " ".join(extra)
    if not os.path.exists(flags_file) or open(flags_file).read() != flags:
        force = True
        open(flags_file, "w").write(flags)
    base = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", *extra, "-I", INCLUDE, "-I", CSRC]
    todo = [s for s in sources() if force or _stale(_obj(s), [s] + headers())]

    def compile_one(src):
        cmd = base + PER_SOURCE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", _obj(src)]
        if verbose:
            print("[mstg_hip.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as ex:
        list(ex.map(compile_one, todo))
    for stale in set(glob.glob(os.path.join(OBJ, "*.o"))) - {_obj(s) for s in sources()}:
        os.remove(stale)
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *[_obj(s) for s in sources()], "-o", LIB]
    if verbose:
        print("[mstg_hip.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
