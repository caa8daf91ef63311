"""DEBUG: phase stamps of tap-split weight-gradient workgroups (third tile of each sampled workgroup; GPU box); needs a library
built with MSTG_HIPCC_FLAGS=-DMSTG_STAMPS python multi-style-transfer-gan_amd/mstg_hip/build.py --force.
usage: diag_stamps_wgrad.py [conv|convT]"""
import os, sys, ctypes, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-style-transfer-gan_amd")]
import torch
from mstg_hip import ops, _lib
lib = _lib.load()
T = len(sys.argv) > 1 and sys.argv[1] == "convT"
N, H, W, Cin, Cout, k, s_, p = (64, 128, 128, 32, 16, 4, 2, 1) if T else (64, 256, 256, 16, 32, 4, 2, 1)
dev = "cuda:0"
conv = (torch.nn.ConvTranspose2d if T else torch.nn.Conv2d)(Cin, Cout, k, s_, p)
Ho, Wo = (2 * H, 2 * W) if T else (H // 2, W // 2)
x = torch.randn((N, H, W, Cin), device=dev)
dy = torch.randn((N, Ho, Wo, Cout), device=dev)
desc = ops.make_desc(N, H, W, Cin, Ho, Wo, Cout, k, s_, p, 1, int(T), 0, 0)
dw = torch.empty_like(conv.weight, device=dev)
db = None if T else torch.empty(Cout, device=dev)
print("kernel:", ops.conv_kernel_name(desc, 2) if hasattr(ops, "conv_kernel_name") else "?")
for _ in range(3):
    ops.conv_wgrad_raw(desc, x, dy, dw, db)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); ops.conv_wgrad_raw(desc, x, dy, dw, db); t1.record(); torch.cuda.synchronize()
print(f"call {t0.elapsed_time(t1) * 1e3:.0f} us")
buf = (ctypes.c_ulonglong * 1024)()
fn = lib.mstg_debug_stamps_wgrad
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
names = ["wait barrier 1 (others still computing)", "stage: loads + LDS writes", "wait barrier 2", "bias column sums", "MFMA loop"]
for wv, label in ((0, "thread 0 (wave 0)"), (1, "thread 192 (wave 3)")):
    rows = [[buf[(wv * 64 + i) * 8 + j] for j in range(6)] for i in range(64)]
    rows = [r for r in rows if r[0] and r[5] > r[0]]
    print(f"{label}: {len(rows)} sampled workgroups; s_memtime ticks = shader cycles")
    for j in range(5):
        ds = [r[j + 1] - r[j] for r in rows]
        print(f"  {names[j]:42s} median {statistics.median(ds):7.0f}  min {min(ds):7.0f}  max {max(ds):7.0f}")
    print(f"  {'tile total':42s} median {statistics.median([r[5] - r[0] for r in rows]):7.0f}")
