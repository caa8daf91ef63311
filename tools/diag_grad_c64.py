"""Diagnostic (GPU box): generator gradients at the class-default width -- HIP fp32 and CPU-oracle fp32, both against the
CPU oracle in fp64.  usage: diag_grad_c64.py [C] [H] [W] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-style-transfer-gan_amd")]
import torch
from oracle import restatement as R
import enhanced_generator as eg

def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
W = int(sys.argv[3]) if len(sys.argv) > 3 else 32
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 31
shape = (1, 3, H, W)
sd = R.make_state_dict(R.generator_spec(C), seed)
names = [k for k in sd if not k.startswith("style_encoder")]
x = R.make_input(shape, seed + 1)
res = {}
for tag, dt in (("cpu32", torch.float32), ("cpu64", torch.float64)):
    s2 = {k: v.to(dt).requires_grad_(True) for k, v in sd.items()}
    xi = x.to(dt).requires_grad_(True)
    y = R.generator_forward(s2, xi)
    res[tag] = (y, torch.autograd.grad(y.abs().mean(), [xi] + [s2[k] for k in names]))
m = eg.EnhancedGenerator(C, 0); m.load_state_dict(sd); m.cuda()
xi = x.cuda().requires_grad_(True)
y = m(xi)
params = dict(m.named_parameters())
res["hip"] = (y, torch.autograd.grad(y.abs().mean(), [xi] + [params[k] for k in names]))
print(f"== C={C} shape={shape}: out hip-vs-f64 {rel(res['hip'][0], res['cpu64'][0]):.2e}  cpu32-vs-f64 {rel(res['cpu32'][0], res['cpu64'][0]):.2e}")
for i, k in enumerate(["dx"] + names):
    a, b = rel(res["hip"][1][i], res["cpu64"][1][i]), rel(res["cpu32"][1][i], res["cpu64"][1][i])
    flag = "  <<<" if a > 5 * b + 1e-5 else ""
    print(f"   {k:32s} hip-vs-f64 {a:.2e}   cpu32-vs-f64 {b:.2e}   hip-vs-cpu32 {rel(res['hip'][1][i], res['cpu32'][1][i]):.2e}{flag}")
