"""Arbiter for tools/diag_style_step.py: the same generator-phase gradients from the CPU oracle in fp32 and in fp64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from oracle import restatement as R
def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
C, shape = 8, (2, 3, 32, 32)
out = {}
for dt in (torch.float32, torch.float64):
    sds = [R.make_state_dict(R.generator_spec(C), 91), R.make_state_dict(R.generator_spec(C), 92),
           R.make_state_dict(R.discriminator_spec(C), 93), R.make_state_dict(R.discriminator_spec(C), 94)]
    g_ab = {k: v.to(dt).requires_grad_(True) for k, v in sds[0].items()}
    g_ba = {k: v.to(dt).requires_grad_(True) for k, v in sds[1].items()}
    d_a, d_b = {k: v.to(dt) for k, v in sds[2].items()}, {k: v.to(dt) for k, v in sds[3].items()}
    a, b = R.make_input(shape, 98).to(dt), R.make_input(shape, 99).to(dt)
    fake_B, fake_A = R.generator_forward(g_ab, a), R.generator_forward(g_ba, b)
    for sd_, inp in ((d_a, a), (d_b, b), (d_a, fake_A.detach()), (d_b, fake_B.detach())):
        R.discriminator_forward(sd_, inp)
    idt = (R.l1(R.generator_forward(g_ba, a), a) + R.l1(R.generator_forward(g_ab, b), b)) * 2.0
    fa, _ = R.discriminator_forward(d_a, fake_A); fb, _ = R.discriminator_forward(d_b, fake_B)
    gl = R.mse(fa, 1.0) + R.mse(fb, 1.0)
    cyc = (R.l1(R.generator_forward(g_ba, fake_B), a) + R.l1(R.generator_forward(g_ab, fake_A), b)) * 10.0
    _, ras = R.discriminator_forward(d_a, a); _, fas = R.discriminator_forward(d_a, fake_A)
    _, rbs = R.discriminator_forward(d_b, b); _, fbs = R.discriminator_forward(d_b, fake_B)
    st = (R.l1(ras, fas) + R.l1(rbs, fbs)) * 0.5
    names = [k for k in g_ab if k.endswith("weight") and not k.startswith("style_encoder")]
    out[dt] = (names, torch.autograd.grad(gl + cyc + idt + st, [g_ab[k] for k in names] + [g_ba[k] for k in names]))
names = out[torch.float32][0]
errs = sorted([(rel(x, y), k) for x, y, k in zip(out[torch.float32][1], out[torch.float64][1], names + names)], reverse=True)
print("oracle fp32 vs fp64, worst tensors:", errs[:5])
