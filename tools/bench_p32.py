"""GPU box: the persistent 4x4 stride-2 kernel (conv_p32.hip) against igemm_light on the layers of the headline step.
usage: python tools/bench_p32.py   (prints ms per launch for MSTG_P32=0 / default / forced variants)"""
import os
import sys
import torch
sys.path.insert(0, "multi-style-transfer-gan_amd")
from mstg_hip import ops

dev = "cuda:0"
# (tag, N, H, W, Cin, Cout, transposed)
LAYERS = [("down1 16->32", 64, 256, 256, 16, 32, 0), ("down2 32->64", 64, 128, 128, 32, 64, 0), ("up1 T 64->32", 64, 64, 64, 64, 32, 1),
          ("up2 T 32->16", 64, 128, 128, 32, 16, 1), ("D 16->32", 32, 128, 128, 16, 32, 0), ("D 32->64", 32, 64, 64, 32, 64, 0)]


def time_it(fn, reps=10):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


def run(tag_env):
    for k in ("MSTG_P32", "MSTG_P32_TH", "MSTG_P32_WLDS", "MSTG_P32_DBG"):
        os.environ.pop(k, None)
    for kv in tag_env.split():
        k, v = kv.split("=")
        os.environ[k] = v
    ops.refresh_env()
    out = []
    for tag, N, H, W, Ci, Co, tr in LAYERS:
        Ho, Wo = (2 * H, 2 * W) if tr else (H // 2, W // 2)
        x = torch.randn(N, H, W, Ci, device=dev)
        w = torch.randn((Ci, Co, 4, 4) if tr else (Co, Ci, 4, 4), device=dev) * 0.05
        b = torch.randn(Co, device=dev)
        y = torch.empty(N, Ho, Wo, Co, device=dev)
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        d = ops.make_desc(N, H, W, Ci, Ho, Wo, Co, 4, 2, 1, 1, transposed=tr)
        tf = time_it(lambda: ops.conv_fwd_raw(d, x, w, b, y))
        tb = time_it(lambda: ops.conv_dgrad_raw(d, dy, w, dx))
        dw, db = torch.empty_like(w), torch.empty_like(b)
        tw = time_it(lambda: ops.conv_wgrad_raw(d, x, dy, dw, None if tr else db))
        fl = 2.0 * N * (Ho * Wo if not tr else H * W * 4) * Co * Ci * (16 if not tr else 4)
        names = " / ".join(ops._kernel_name(d, k).replace("conv_p32_kernel", "p32").replace("igemm_light_kernel", "light") for k in (0, 1))
        out.append(f"{tag:14s} {names:46s} fwd {tf:.3f} ms ({fl / tf / 1e9:5.1f} TF)  dgrad {tb:.3f} ms ({fl / tb / 1e9:5.1f} TF)  wgrad {tw:.3f} ms ({fl / tw / 1e9:5.1f} TF)")
    for tag, N, H, Ci in (("1x1 16->16", 64, 256, 16), ("1x1 32->32", 64, 128, 32), ("1x1 64->64", 64, 64, 64)):
        x = torch.randn(N, H, H, Ci, device=dev)
        w = torch.randn(Ci, Ci, 1, 1, device=dev) * 0.1
        b = torch.randn(Ci, device=dev)
        y = torch.empty_like(x)
        d = ops.make_desc(N, H, H, Ci, H, H, Ci, 1, 1, 0, 1)
        tf = time_it(lambda: ops.conv_fwd_raw(d, x, w, b, y))
        tb = time_it(lambda: ops.conv_dgrad_raw(d, y, w, x))
        dw, db = torch.empty_like(w), torch.empty_like(b)
        xx, yy = torch.randn_like(x), torch.randn_like(x)
        tw = time_it(lambda: ops.conv_wgrad_raw(d, xx, yy, dw, db))
        gb = 2 * x.numel() * 4 / 1e9
        out.append(f"{tag:14s} fwd {tf:.3f} ms ({gb / tf * 1e3:5.0f} GB/s)  dgrad {tb:.3f} ms ({gb / tb * 1e3:5.0f} GB/s)  wgrad {tw:.3f} ms ({gb / tw * 1e3:5.0f} GB/s)")
    for tag, N, H, Ci, Co, xn, yn in (("stem 3->16 k7", 64, 256, 3, 16, 1, 0), ("head 16->3 k7", 64, 256, 16, 3, 0, 1)):
        x = torch.randn((N, Ci, H, H) if xn else (N, H, H, Ci), device=dev)
        w = torch.randn(Co, Ci, 7, 7, device=dev) * 0.05
        b = torch.randn(Co, device=dev)
        y = torch.empty((N, Co, H, H) if yn else (N, H, H, Co), device=dev)
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        d = ops.make_desc(N, H, H, Ci, H, H, Co, 7, 1, 3, 1, x_nchw=xn, y_nchw=yn)
        tf = time_it(lambda: ops.conv_fwd_raw(d, x, w, b, y))
        tb = time_it(lambda: ops.conv_dgrad_raw(d, dy, w, dx))
        dw, db = torch.empty_like(w), torch.empty_like(b)
        tw = time_it(lambda: ops.conv_wgrad_raw(d, x, dy, dw, db))
        fl = 2.0 * N * H * H * Co * Ci * 49
        out.append(f"{tag:14s} fwd {tf:.3f} ms ({fl / tf / 1e9:5.1f} TF)  dgrad {tb:.3f} ms ({fl / tb / 1e9:5.1f} TF)  wgrad {tw:.3f} ms ({fl / tw / 1e9:5.1f} TF)")
    print(f"--- {tag_env or 'default'}")
    print("\n".join(out), flush=True)


for env in sys.argv[1:] or ["MSTG_P32=0", ""]:
    run(env)
