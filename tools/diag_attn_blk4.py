"""Diagnostic (GPU box): four-wave attention core against the one-wave row-blocked kernel on the same inputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-style-transfer-gan_amd")]
import torch
from mstg_hip import ops

def run(qkv, do, blk4):
    os.environ["MSTG_ATTN_BLK4"] = "1" if blk4 else "0"
    q = qkv.clone().requires_grad_(True)
    o = ops.WindowAttnCoreFn.apply(q)
    (g,) = torch.autograd.grad((o * do).sum(), [q])
    return o.detach(), g.detach()

torch.manual_seed(0)
for (N, H, W, C) in [(1, 16, 16, 128), (1, 8, 8, 256), (2, 8, 8, 128), (1, 4, 4, 256)]:
    for scale, offs in [(2.0, 0.0), (0.3, 1.0), (8.0, 0.0), (1.0, 3.0)]:
        qkv = (torch.randn(N, H, W, 3 * C) * scale + offs * torch.randn(1, 1, 1, 3 * C)).cuda()
        do = torch.randn(N, H, W, C).cuda()
        o1, g1 = run(qkv, do, False)
        o4, g4 = run(qkv, do, True)
        rel = lambda a, b: float((a - b).norm() / b.norm())
        line = f"N{N} {H}x{W} C{C} scale {scale} offs {offs}: o {rel(o4, o1):.2e}  dq {rel(g4[..., :C], g1[..., :C]):.2e}  dk {rel(g4[..., C:2*C], g1[..., C:2*C]):.2e}  dv {rel(g4[..., 2*C:], g1[..., 2*C:]):.2e}"
        d = (g4 - g1).abs()
        bad = (d > 1e-3 * g1.abs().max()).nonzero()
        line += f"  bad {bad.shape[0]}"
        if bad.shape[0]:
            line += f" first {bad[:4].tolist()} chan-part {sorted(set((int(b[3]) // C) for b in bad))} px {sorted(set((int(b[1]) % 4) * 4 + int(b[2]) % 4 for b in bad))[:16]}"
        print(line)
