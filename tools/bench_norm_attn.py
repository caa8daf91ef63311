"""GPU box: fused norm+attention against the chain, per direction (HIP events).  usage: python tools/bench_norm_attn.py [C N H]"""
import sys
import torch
sys.path.insert(0, "multi-style-transfer-gan_amd")
from mstg_hip import ops

C, N, H = (int(v) for v in (sys.argv[1:4] + ["16", "64", "256"][len(sys.argv) - 1:]))
dev = "cuda:0"
x = torch.randn(N, H, H, C, device=dev)
p = [torch.randn(3 * C, C, 1, 1, device=dev) * 0.2, torch.randn(3 * C, device=dev) * 0.1, torch.randn(C, C, 1, 1, device=dev) * 0.2,
     torch.randn(C, device=dev) * 0.1]
dy = torch.randn_like(x)


def run(fused, reps=5):
    t = [v.clone().requires_grad_(True) for v in [x] + p]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for r in range(reps + 2):
        for v in t:
            v.grad = None
        ev[0].record()
        y = (ops.NormLocalAttentionFn.apply(*t) if fused else ops.LocalAttentionFusedFn.apply(ops.instnorm_act(t[0], ops.ACT_RELU), *t[1:]))
        ev[1].record()
        y.backward(dy)
        ev[2].record()
        torch.cuda.synchronize()
        if r >= 2:
            tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    return tf / reps, tb / reps


for fused in (False, True, False, True):
    f, b = run(fused)
    print(f"C{C} N{N} {H}x{H} fused={int(fused)}: fwd {f:.3f} ms  bwd {b:.3f} ms", flush=True)
