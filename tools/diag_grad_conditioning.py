"""Diagnostic (GPU box): is a gradient mismatch a kernel bug or fp32 conditioning?  Compares HIP fp32 and CPU-oracle
fp32 against the CPU oracle in fp64 for the end-to-end generator gradient case of tests/golden."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-style-transfer-gan_amd")]
import torch
from oracle import restatement as R
import enhanced_generator as eg

def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

for C, shape, seed in ((8, (2, 3, 32, 48), 31), (16, (1, 3, 64, 64), 32)):
    sd = R.make_state_dict(R.generator_spec(C), seed)
    names = [k for k in sd if not k.startswith("style_encoder")]
    x = R.make_input(shape, seed + 100)
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
    y64 = res["cpu64"][0]
    print(f"   sign flips of y vs f64: hip {int((torch.sign(res['hip'][0].cpu().double()) != torch.sign(y64)).sum())}  cpu32 {int((torch.sign(res['cpu32'][0].double()) != torch.sign(y64)).sum())}  of {y64.numel()}")
    for i, k in enumerate(["dx"] + names):
        if k.endswith("bias") and "qkv" not in k and "proj" not in k and not k.startswith("output"):
            continue
        a, b = rel(res["hip"][1][i], res["cpu64"][1][i]), rel(res["cpu32"][1][i], res["cpu64"][1][i])
        flag = "  <<<" if a > 5 * b + 1e-5 else ""
        print(f"   {k:32s} hip-vs-f64 {a:.2e}   cpu32-vs-f64 {b:.2e}{flag}")

# ---- where do the C=16 gradients first diverge?  ReLU masks of the down2 multi-scale block, HIP vs fp64 ------------
import torch.nn.functional as F
from mstg_hip import ops
C, shape, seed = 16, (1, 3, 64, 64), 32
sd = R.make_state_dict(R.generator_spec(C), seed)
x = R.make_input(shape, seed + 100)
s64 = {k: v.double() for k, v in sd.items()}
h = F.relu(R.instance_norm(F.conv2d(x.double(), s64["initial.0.weight"], s64["initial.0.bias"], padding=3)))
h = R._stage(h, s64, "down1", False)
h = F.conv2d(h, s64["down2.0.weight"], s64["down2.0.bias"], stride=2, padding=1)
h = F.relu(R.instance_norm(h))
h = R.local_attention(h, s64, "down2.3", 4)
pre64 = torch.cat([R.instance_norm(F.conv2d(h, s64["down2.4.branch1.0.weight"], s64["down2.4.branch1.0.bias"]))] +
                  [R.instance_norm(F.conv2d(h, s64[f"down2.4.branch{b}.0.weight"], s64[f"down2.4.branch{b}.0.bias"], padding=d, dilation=d))
                   for b, d in ((2, 1), (3, 2), (4, 4))], 1)
m = eg.EnhancedGenerator(C, 0); m.load_state_dict(sd); m.cuda()
with torch.no_grad():
    g = m.initial[0](x.cuda(), nhwc=True, x_nchw=True); g = ops.instnorm_act(g, 1)
    g = m.down1.forward_nhwc(g)
    g = m.down2[0](g, nhwc=True); g = ops.instnorm_act(g, 1); g = m.down2[3].forward_nhwc(g)
    msb = m.down2[4]
    wb = []
    for br in (msb.branch1, msb.branch2, msb.branch3, msb.branch4):
        wb += [br[0].weight, br[0].bias]
    cat, _ = ops.MSBranchesFn.apply(g, *wb)
    pre_hip = ops.instnorm_act(cat, 0).permute(0, 3, 1, 2).cpu().double()
flips = (pre_hip > 0) != (pre64 > 0)
print(f"== down2.4 normalised branch outputs: max|hip - f64| {float((pre_hip - pre64).abs().max()):.2e}; ReLU-mask flips: {int(flips.sum())} of {flips.numel()}")
for idx in flips.nonzero().tolist():
    n, c, yy, xx = idx
    print(f"   flip at channel {c} (branch {c // 16 + 1}) pixel ({yy},{xx}): hip {float(pre_hip[n, c, yy, xx]):+.3e}  f64 {float(pre64[n, c, yy, xx]):+.3e}")
