"""Diagnostic (GPU box): where the four-wave attention core changes the class-default generator's input gradient --
forward (activation perturbation) or backward.  MSTG_ATTN_BLK4 is read per call, so it can differ between the two passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-style-transfer-gan_amd")]
import torch
from oracle import restatement as R
import enhanced_generator as eg

def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

C, H, W, seed = 64, 32, 32, 31
sd = R.make_state_dict(R.generator_spec(C), seed)
x = R.make_input((1, 3, H, W), seed + 1)
s2 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
xi = x.double().requires_grad_(True)
y = R.generator_forward(s2, xi)
(ref,) = torch.autograd.grad(y.abs().mean(), [xi])
m = eg.EnhancedGenerator(C, 0); m.load_state_dict(sd); m.cuda()
for fwd4, bwd4 in ((0, 0), (1, 0), (0, 1), (1, 1)):
    os.environ["MSTG_ATTN_BLK4"] = str(fwd4)
    xg = x.cuda().requires_grad_(True)
    yg = m(xg)
    os.environ["MSTG_ATTN_BLK4"] = str(bwd4)
    (g,) = torch.autograd.grad(yg.abs().mean(), [xg])
    print(f"fwd blk4={fwd4} bwd blk4={bwd4}: out vs f64 {rel(yg, y):.2e}  dx vs f64 {rel(g, ref):.2e}")

# ---- which activations differ between the two forwards? ----
acts = {}
def hook(name, store):
    def f(mod, inp, out):
        if torch.is_tensor(out):
            store[name] = out.detach().clone()
    return f
runs = []
for fwd4 in (0, 1):
    os.environ["MSTG_ATTN_BLK4"] = str(fwd4)
    store = {}
    hs = [mod.register_forward_hook(hook(n, store)) for n, mod in m.named_modules() if n]
    with torch.no_grad():
        m(x.cuda())
    for h in hs:
        h.remove()
    runs.append(store)
for n in runs[0]:
    a, b = runs[0][n], runs[1][n]
    d = rel(b, a)
    flips = int(((a > 0) != (b > 0)).sum())
    zeros = int((a == 0).sum())
    if d > 0 or flips:
        print(f"{n:28s} shape {tuple(a.shape)} rel diff {d:.2e} sign flips {flips} exact zeros {zeros}/{a.numel()} max|a| {float(a.abs().max()):.2e}")
