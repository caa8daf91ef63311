"""DEBUG: phase stamps of the fused attention backward (library built with MSTG_HIPCC_FLAGS=-DMSTG_STAMPS). usage: diag_stamps_attn.py [C]"""
import os, sys, ctypes, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-style-transfer-gan_amd")]
import torch
from mstg_hip import ops, _lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N, H, W = (32, 256, 256) if C == 16 else (32, 128, 128)
dev = "cuda:0"
x = torch.randn((N, H, W, C), device=dev, requires_grad=True)
wq = (torch.randn((3 * C, C, 1, 1), device=dev) * 0.2).requires_grad_(True); bq = torch.zeros(3 * C, device=dev, requires_grad=True)
wp = (torch.randn((C, C, 1, 1), device=dev) * 0.2).requires_grad_(True); bp = torch.zeros(C, device=dev, requires_grad=True)
for _ in range(2):
    y = ops.LocalAttentionFusedFn.apply(x, wq, bq, wp, bp)
    y.sum().backward()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
fn = ctypes.CDLL(_lib.LIB_PATH)\
    .mstg_debug_stamps_attn
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
rows = [[buf[i * 8 + j] for j in range(7)] for i in range(64)]
rows = [r for r in rows if r[0] and r[6] > r[0]]
names = ["load dy + forward recompute", "proj backward", "softmax/core bwd (dS, dV)", "dq^, dk^ + normalise bwd", "qkv conv bwd + weight grads", "-"]
print(f"C={C}: {len(rows)} sampled waves; shader cycles per window")
for j in range(5):
    ds = [r[j + 1] - r[j] for r in rows]
    print(f"  {names[j]:32s} median {statistics.median(ds):8.0f}  min {min(ds):8.0f}  max {max(ds):8.0f}")
print(f"  total {statistics.median([r[5] - r[0] for r in rows]):8.0f}")

fn2 = ctypes.CDLL(_lib.LIB_PATH).mstg_debug_stamps_attn_fwd
fn2.restype = ctypes.c_int; fn2.argtypes = [ctypes.c_void_p]
assert fn2(buf) == 0
rows = [[buf[i * 8 + j] for j in range(6)] for i in range(64)]
rows = [r for r in rows if r[0] and r[5] > r[0]]
names = ["x window -> LDS + sync", "qkv 1x1 conv (+bias, store)", "normalise q, k", "S = q^T k, softmax, store P", "O = P V, store"]
print(f"forward tiles (last window of each sampled wave, inside the backward kernel):")
for j in range(5):
    ds = [r[j + 1] - r[j] for r in rows]
    print(f"  {names[j]:32s} median {statistics.median(ds):8.0f}  min {min(ds):8.0f}  max {max(ds):8.0f}")
