"""Micro-benchmark of single convolution passes through the C ABI (GPU box).  Usage:
   python tools/bench_conv.py [filter-substring] [fwd|dgrad|wgrad]
Prints per case: ms, TFLOP/s (algorithmic), GB/s (algorithmic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-style-transfer-gan_amd")]
import torch
from mstg_hip import ops

# name, N,H,W,Cin,Cout,k,s,p,d,transposed,x_nchw,y_nchw
CASES = [
    ("stem7 3->16", 32, 256, 256, 3, 16, 7, 1, 3, 1, 0, 1, 0),
    ("head7 16->3", 32, 256, 256, 16, 3, 7, 1, 3, 1, 0, 0, 1),
    ("k4s2 16->32 @256", 32, 256, 256, 16, 32, 4, 2, 1, 1, 0, 0, 0),
    ("k4s2 32->64 @128", 32, 128, 128, 32, 64, 4, 2, 1, 1, 0, 0, 0),
    ("convT 64->32 @64", 32, 64, 64, 64, 32, 4, 2, 1, 1, 1, 0, 0),
    ("convT 32->16 @128", 32, 128, 128, 32, 16, 4, 2, 1, 1, 1, 0, 0),
    ("1x1 32->96 @128", 32, 128, 128, 32, 96, 1, 1, 0, 1, 0, 0, 0),
    ("1x1 16->16 @256", 32, 256, 256, 16, 16, 1, 1, 0, 1, 0, 0, 0),
    ("1x1 64->64 @64", 32, 64, 64, 64, 64, 1, 1, 0, 1, 0, 0, 0),
    ("k3d1 16->4 @256", 32, 256, 256, 16, 4, 3, 1, 1, 1, 0, 0, 0),
    ("k3d4 16->4 @256", 32, 256, 256, 16, 4, 3, 1, 4, 4, 0, 0, 0),
    ("k3d2 32->8 @128", 32, 128, 128, 32, 8, 3, 1, 2, 2, 0, 0, 0),
    ("k3d1 64->16 @64", 32, 64, 64, 64, 16, 3, 1, 1, 1, 0, 0, 0),
    ("D k4s2 3->16", 32, 256, 256, 3, 16, 4, 2, 1, 1, 0, 1, 0),
    ("D k4s2 64->128 @32", 32, 32, 32, 64, 128, 4, 2, 1, 1, 0, 0, 0),
    ("D k3 128->128 @16", 32, 16, 16, 128, 128, 3, 1, 1, 1, 0, 0, 0),
    ("D k4s1 128->1 @16", 32, 16, 16, 128, 1, 4, 1, 1, 1, 0, 0, 0),
]

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

filt = sys.argv[1] if len(sys.argv) > 1 else ""
only = sys.argv[2] if len(sys.argv) > 2 else ""   # fwd | dgrad | wgrad
dev = "cuda:0"
for name, N, H, W, Cin, Cout, k, s_, p, d, tr, xn, yn in CASES:
    if filt and filt not in name: continue
    Ho, Wo = ops.conv_out_hw(H, W, k, s_, p, d, tr)
    x = torch.randn((N, Cin, H, W) if xn else (N, H, W, Cin), device=dev)
    w = torch.randn((Cin, Cout, k, k) if tr else (Cout, Cin, k, k), device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    y = torch.empty((N, Cout, Ho, Wo) if yn else (N, Ho, Wo, Cout), device=dev)
    dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(b)
    desc = ops.make_desc(N, H, W, Cin, Ho, Wo, Cout, k, s_, p, d, tr, xn, yn)
    fl, by = ops._conv_cost(desc)
    for tag, fn in (("fwd", lambda: ops.conv_fwd_raw(desc, x, w, b, y)), ("dgrad", lambda: ops.conv_dgrad_raw(desc, dy, w, dx)),
                    ("wgrad", lambda: ops.conv_wgrad_raw(desc, x, dy, dw, None if (tr or Cout <= 4) else db))):
        if only and only != tag: continue
        ms = timeit(fn)
        print(f"{name:22s} {tag:6s} {ms:8.3f} ms  {fl / ms / 1e9:7.2f} TFLOP/s  {by / ms / 1e6:8.1f} GB/s", flush=True)
