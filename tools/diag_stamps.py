"""DEBUG: phase stamps of igemm_light workgroups for one conv case (GPU box); needs a library built with
MSTG_HIPCC_FLAGS=-DMSTG_STAMPS python multi-style-transfer-gan_amd/mstg_hip/build.py --force.  usage: diag_stamps.py d4|d1|1x1"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-style-transfer-gan_amd")]
import torch
from mstg_hip import ops, _lib
import importlib.util
lib = _lib.load()
N, H, W, Cin, Cout, k, s_, p, d = 32, 256, 256, 16, 4, 3, 1, 4, 4
if len(sys.argv) > 1 and sys.argv[1] == "d1": p, d = 1, 1
if len(sys.argv) > 1 and sys.argv[1] == "1x1": Cout, k, p, d = 16, 1, 0, 1
if len(sys.argv) > 1 and sys.argv[1] == "k7": Cout, k, p, d = 3, 7, 3, 1
dev = "cuda:0"
Ho, Wo = ops.conv_out_hw(H, W, k, s_, p, d, 0)
x = torch.randn((N, H, W, Cin), device=dev); w = torch.randn((Cout, Cin, k, k), device=dev) * 0.05; b = torch.randn(Cout, device=dev)
y = torch.empty((N, Ho, Wo, Cout), device=dev)
desc = ops.make_desc(N, H, W, Cin, Ho, Wo, Cout, k, s_, p, d, 0, 0, 0)
for _ in range(3): ops.conv_fwd_raw(desc, x, w, b, y)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
lib._handle if False else None
cdll = ctypes.CDLL(_lib.LIB_PATH) if hasattr(_lib, "LIB_PATH") else None
fn = (cdll or lib).mstg_debug_stamps
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
import statistics
rows = [[buf[i * 8 + j] for j in range(7)] for i in range(64)]
rows = [r for r in rows if r[0] and r[6] > r[0]]
names = ["setup->patch staged", "->barrier1", "->filter staged", "->barrier2", "->mfma done", "->epilogue done"]
if os.environ.get("MSTG_STREAM", "1") != "0": names = ["issue next loads", "mfma", "epilogue", "barrier A", "wait loads + LDS write", "barrier B"]
print(f"{len(rows)} sampled workgroups; s_memtime ticks = shader cycles")
for j in range(6):
    ds = [r[j + 1] - r[j] for r in rows]
    print(f"  {names[j]:24s} median {statistics.median(ds):8.0f}  min {min(ds):8.0f}  max {max(ds):8.0f}")
tot = [r[6] - r[0] for r in rows]
print(f"  total                    median {statistics.median(tot):8.0f}")
