"""Generate tests/golden/*.npz from the REFERENCE's own modules -- run in the build container only.

TEST INFRASTRUCTURE.  Imports /root/reference by path (it never travels to the GPU box), feeds the
reference's ``EnhancedGenerator`` / ``EnhancedDiscriminator`` / ``LocalAttention`` /
``MultiScaleBlock`` / plain ``Generator`` / unmodified ``EnhancedCycleGAN.train_step`` with the
deterministic numpy-seeded weights and inputs of ``oracle.restatement`` and stores inputs' seeds and
the reference's outputs.  Also asserts that ``oracle.restatement`` reproduces every stored vector
(<= 1e-5 relative), which is what pins the oracle.

Two placeholders are registered before import:
  * ``structural_transformer`` -- the reference file is missing from the snapshot; the class is only
    constructed for num_transformer_blocks > 0, which is never used here;
  * ``torchvision.transforms`` -- absent from this image; only needed so ``pretrain.py`` imports.

Usage:  python oracle/make_golden.py            (writes tests/golden/)
"""
from __future__ import annotations

import os
import sys
import types
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restatement as R  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def import_reference():
    st = types.ModuleType("structural_transformer")

    class StructuralTransformerBlock(torch.nn.Module):  # never instantiated (blocks=0)
        def __init__(self, *a, **k):
            raise RuntimeError("structural_transformer.py is missing from the reference snapshot")

    st.StructuralTransformerBlock = StructuralTransformerBlock
    sys.modules["structural_transformer"] = st
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")
        tv.transforms = tvt
        sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    sys.path.insert(0, REF)
    import enhanced_generator as eg  # type: ignore
    import enhanced_train as et  # type: ignore
    import pretrain as pt  # type: ignore
    return eg, et, pt


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def check(name, mine, ref, tol=1e-5):
    e = rel(mine, ref)
    print(f"  restatement vs reference  {name:38s} rel-L2 {e:.2e}")
    assert e <= tol, (name, e)


def npy(t):
    return t.detach().cpu().numpy().copy()  # copy: buffers such as weight_u are updated in place later


def clone_sd(sd):
    return {k: v.clone() for k, v in sd.items()}


def grads_of(loss, sd, keys):
    g = torch.autograd.grad(loss, [sd[k] for k in keys], allow_unused=True)
    return {k: gi for k, gi in zip(keys, g)}


def gen_ops(eg):
    """Per-op fixtures from the reference's LocalAttention / MultiScaleBlock (+ grads)."""
    out = {}
    for tag, ch, shape, seed in (("a", 8, (2, 8, 8, 12), 11), ("b", 16, (1, 16, 4, 8), 12)):
        spec = [("qkv.weight", (3 * ch, ch, 1, 1)), ("qkv.bias", (3 * ch,)), ("proj.weight", (ch, ch, 1, 1)), ("proj.bias", (ch,))]
        sd = R.make_state_dict(spec, seed)
        x = R.make_input(shape, seed + 100).requires_grad_(True)
        m = eg.LocalAttention(ch, window_size=4)
        m.load_state_dict(sd)
        y = m(x)
        gy = R.make_input(tuple(y.shape), seed + 200)
        gx, *gp = torch.autograd.grad((y * gy).sum(), [x] + list(m.parameters()))
        names = [k for k, _ in m.named_parameters()]
        # restatement
        sd2 = {("p." + k): v.clone().requires_grad_(True) for k, v in sd.items()}
        x2 = x.detach().clone().requires_grad_(True)
        y2 = R.local_attention(x2, sd2, "p", 4)
        g2 = torch.autograd.grad((y2 * gy).sum(), [x2] + [sd2["p." + k] for k in names])
        check(f"local_attention[{tag}] y", y2, y)
        check(f"local_attention[{tag}] dx", g2[0], gx)
        for k, a, b in zip(names, g2[1:], gp):
            check(f"local_attention[{tag}] d{k}", a, b)
        out.update({f"attn_{tag}_ch": ch, f"attn_{tag}_shape": np.array(shape), f"attn_{tag}_seed": seed,
                    f"attn_{tag}_y": npy(y), f"attn_{tag}_dx": npy(gx)})
        out.update({f"attn_{tag}_d_{k}": npy(g) for k, g in zip(names, gp)})
    for tag, ch, shape, seed in (("a", 8, (2, 8, 12, 8), 21), ("b", 16, (1, 16, 16, 16), 22)):
        spec = [("branch1.0.weight", (ch // 4, ch, 1, 1)), ("branch1.0.bias", (ch // 4,))]
        for b in (2, 3, 4):
            spec += [(f"branch{b}.0.weight", (ch // 4, ch, 3, 3)), (f"branch{b}.0.bias", (ch // 4,))]
        spec += [("fusion.0.weight", (ch, ch, 1, 1)), ("fusion.0.bias", (ch,))]
        sd = R.make_state_dict(spec, seed)
        x = R.make_input(shape, seed + 100).requires_grad_(True)
        m = eg.MultiScaleBlock(ch)
        m.load_state_dict(sd)
        y = m(x)
        gy = R.make_input(tuple(y.shape), seed + 200)
        names = [k for k, _ in m.named_parameters()]
        gx, *gp = torch.autograd.grad((y * gy).sum(), [x] + list(m.parameters()))
        sd2 = {("p." + k): v.clone().requires_grad_(True) for k, v in sd.items()}
        x2 = x.detach().clone().requires_grad_(True)
        y2 = R.multi_scale_block(x2, sd2, "p")
        g2 = torch.autograd.grad((y2 * gy).sum(), [x2] + [sd2["p." + k] for k in names])
        check(f"multi_scale_block[{tag}] y", y2, y)
        check(f"multi_scale_block[{tag}] dx", g2[0], gx, 2e-5)
        for k, a, b in zip(names, g2[1:], gp):
            if not k.endswith("bias"):  # conv biases in front of IN have (numerically noisy) zero gradient
                check(f"multi_scale_block[{tag}] d{k}", a, b, 2e-5)
        out.update({f"msb_{tag}_ch": ch, f"msb_{tag}_shape": np.array(shape), f"msb_{tag}_seed": seed,
                    f"msb_{tag}_y": npy(y), f"msb_{tag}_dx": npy(gx)})
        out.update({f"msb_{tag}_d_{k}": npy(g) for k, g in zip(names, gp)})
    np.savez_compressed(os.path.join(GOLD, "ops.npz"), **out)


def gen_generator(eg):
    """End-to-end EnhancedGenerator(blocks=0): taps + all parameter grads of loss mean(|y|) + dx."""
    for tag, C, shape, seed in (("c8_32x48", 8, (2, 3, 32, 48), 31), ("c16_64x64", 16, (1, 3, 64, 64), 32)):
        sd = R.make_state_dict(R.generator_spec(C), seed)
        x = R.make_input(shape, seed + 100).requires_grad_(True)
        m = eg.EnhancedGenerator(channels=C, num_transformer_blocks=0)
        assert [k for k in m.state_dict()] == [k for k, _ in R.generator_spec(C)], "state_dict key order/spec mismatch"
        assert all(tuple(v.shape) == s for (k, s), v in zip(R.generator_spec(C), m.state_dict().values()))
        m.load_state_dict(sd)
        taps_ref = {}
        h = m.initial(x); taps_ref["initial"] = h
        h = m.down1(h); taps_ref["down1"] = h
        h = m.down2(h); taps_ref["down2"] = h
        h = m.up1(h); taps_ref["up1"] = h
        h = m.up2(h); taps_ref["up2"] = h
        pre = m.output[0](h); taps_ref["pre_tanh"] = pre
        y = m(x)
        assert torch.equal(torch.tanh(pre), y)
        loss = y.abs().mean()
        names = [k for k, p in m.named_parameters() if not k.startswith("style_encoder")]
        params = [p for k, p in m.named_parameters() if not k.startswith("style_encoder")]
        gx, *gp = torch.autograd.grad(loss, [x] + params)
        # checkpointed path is bit-identical
        m2 = eg.EnhancedGenerator(channels=C, num_transformer_blocks=0)
        m2.load_state_dict(sd)
        m2.gradient_checkpointing_enable()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y_ck = m2(x.detach().clone().requires_grad_(True))
        assert torch.equal(y_ck, y), "checkpointed forward differs"
        # restatement
        sd2 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        x2 = x.detach().clone().requires_grad_(True)
        taps = {}
        y2 = R.generator_forward(sd2, x2, taps)
        g2 = torch.autograd.grad(y2.abs().mean(), [x2] + [sd2[k] for k in names])
        for k in ("initial", "down1", "down2", "up1", "up2", "pre_tanh"):
            check(f"G[{tag}] tap {k}", taps[k], taps_ref[k])
        check(f"G[{tag}] out", y2, y)
        check(f"G[{tag}] dx", g2[0], gx, 5e-5)
        worst = max(rel(a, b) for k, a, b in zip(names, g2[1:], gp) if k.endswith("weight"))
        print(f"  restatement vs reference  G[{tag}] worst weight-grad rel-L2 {worst:.2e}")
        assert worst <= 1e-4
        out = {"C": C, "shape": np.array(shape), "seed": seed, "out": npy(y), "pre_tanh": npy(pre), "dx": npy(gx),
               "loss": float(loss)}
        for k in ("initial", "down1", "down2", "up1", "up2"):
            out["tap_" + k] = npy(taps_ref[k])
        out.update({"d_" + k: npy(g) for k, g in zip(names, gp)})
        np.savez_compressed(os.path.join(GOLD, f"generator_{tag}.npz"), **out)


def gen_discriminator(eg):
    C, shape, seed = 8, (2, 3, 64, 64), 41
    sd = R.make_state_dict(R.discriminator_spec(C), seed)
    m = eg.EnhancedDiscriminator(channels=C)
    ref_keys = list(m.state_dict().keys())
    assert sorted(ref_keys) == sorted(k for k, _ in R.discriminator_spec(C)), "D state_dict keys mismatch"
    m.load_state_dict(sd)
    m.train()
    x = R.make_input(shape, seed + 100).requires_grad_(True)
    out = {"C": C, "shape": np.array(shape), "seed": seed}
    sd2 = {k: v.clone() for k, v in sd.items()}
    for k in sd2:
        if not k.endswith(("_u", "_v")):
            sd2[k].requires_grad_(True)
    names = [k for k, _ in m.named_parameters()]
    for it in (1, 2):
        s, st = m(x)
        loss = ((s - 1.0) ** 2).mean() + st.abs().mean()
        gx, *gp = torch.autograd.grad(loss, [x] + list(m.parameters()))
        x2 = x.detach().clone().requires_grad_(True)
        s2, st2 = R.discriminator_forward(sd2, x2, train=True)
        l2 = ((s2 - 1.0) ** 2).mean() + st2.abs().mean()
        g2 = torch.autograd.grad(l2, [x2] + [sd2[k] for k in names])
        check(f"D train fwd#{it} score", s2, s)
        check(f"D train fwd#{it} struct", st2, st)
        check(f"D train fwd#{it} dx", g2[0], gx, 5e-5)
        for k, a, b in zip(names, g2[1:], gp):
            if k.endswith("weight_orig"):
                check(f"D train fwd#{it} d{k}", a, b, 1e-4)
        out.update({f"t{it}_score": npy(s), f"t{it}_struct": npy(st), f"t{it}_dx": npy(gx)})
        out.update({f"t{it}_d_{k}": npy(g) for k, g in zip(names, gp)})
        for k, v in m.state_dict().items():
            if k.endswith(("_u", "_v")):
                out[f"t{it}_{k}"] = npy(v)
                check(f"D train fwd#{it} {k}", sd2[k], v)
    m.eval()
    with torch.no_grad():
        s, st = m(x)
        s2, st2 = R.discriminator_forward(sd2, x.detach(), train=False)
    check("D eval score", s2, s)
    check("D eval struct", st2, st)
    out.update({"eval_score": npy(s), "eval_struct": npy(st)})
    # N=1: score is 0-dim (squeeze)
    with torch.no_grad():
        s1, _ = m(x[:1])
    assert s1.dim() == 0
    np.savez_compressed(os.path.join(GOLD, "discriminator_c8_64x64.npz"), **out)


def gen_plain_generator(pt):
    C, shape, seed = 8, (2, 3, 32, 32), 51
    sd = R.make_state_dict(R.plain_generator_spec(C), seed)
    m = pt.Generator(channels=C)
    assert list(m.state_dict().keys()) == [k for k, _ in sorted(R.plain_generator_spec(C), key=lambda kv: list(m.state_dict().keys()).index(kv[0]))]
    m.load_state_dict(sd)
    m.train()
    x = R.make_input(shape, seed + 100).requires_grad_(True)
    y = m(x)
    names = [k for k, _ in m.named_parameters()]
    gx, *gp = torch.autograd.grad(y.abs().mean(), [x] + list(m.parameters()))
    sd2 = {k: v.clone() for k, v in sd.items()}
    for k in names:
        sd2[k].requires_grad_(True)
    x2 = x.detach().clone().requires_grad_(True)
    y2 = R.plain_generator_forward(sd2, x2, train=True)
    g2 = torch.autograd.grad(y2.abs().mean(), [x2] + [sd2[k] for k in names])
    check("plain G train out", y2, y)
    check("plain G train dx", g2[0], gx, 5e-5)
    for k, a, b in zip(names, g2[1:], gp):
        if k.endswith("weight"):
            check(f"plain G d{k}", a, b, 1e-4)
    out = {"C": C, "shape": np.array(shape), "seed": seed, "train_out": npy(y), "train_dx": npy(gx)}
    out.update({"d_" + k: npy(g) for k, g in zip(names, gp)})
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            out["after_" + k] = npy(v)
            check(f"plain G {k}", sd2[k].float(), v.float())
    m.eval()
    with torch.no_grad():
        ye = m(x)
        ye2 = R.plain_generator_forward(sd2, x.detach(), train=False)
    check("plain G eval out", ye2, ye)
    out["eval_out"] = npy(ye)
    np.savez_compressed(os.path.join(GOLD, "plain_generator_c8_32x32.npz"), **out)


def gen_train_step(eg, et):
    """The reference's UNMODIFIED train_step on an instance assembled without its ctor (which hard-codes
    num_transformer_blocks=1, enhanced_train.py:18-19, and therefore needs the missing module)."""
    C, shape = 8, (2, 3, 64, 64)
    sds = [R.make_state_dict(R.generator_spec(C), 61), R.make_state_dict(R.generator_spec(C), 62),
           R.make_state_dict(R.discriminator_spec(C), 63), R.make_state_dict(R.discriminator_spec(C), 64)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        self = et.EnhancedCycleGAN.__new__(et.EnhancedCycleGAN)
        self.device = torch.device("cpu")
        self.G_AB = eg.EnhancedGenerator(channels=C, num_transformer_blocks=0)
        self.G_BA = eg.EnhancedGenerator(channels=C, num_transformer_blocks=0)
        self.D_A = eg.EnhancedDiscriminator(channels=C)
        self.D_B = eg.EnhancedDiscriminator(channels=C)
        for m, sd in zip((self.G_AB, self.G_BA, self.D_A, self.D_B), sds):
            m.load_state_dict(sd)
        self.G_AB.gradient_checkpointing_enable()
        self.G_BA.gradient_checkpointing_enable()
        import itertools
        self.g_optimizer = torch.optim.Adam(itertools.chain(self.G_AB.parameters(), self.G_BA.parameters()), lr=5e-5, betas=(0.5, 0.999))
        self.d_optimizer = torch.optim.Adam(itertools.chain(self.D_A.parameters(), self.D_B.parameters()), lr=2e-4, betas=(0.5, 0.999))
        self.scaler = torch.cuda.amp.GradScaler()
        self.criterion_gan = torch.nn.MSELoss()
        self.criterion_cycle = torch.nn.L1Loss()
        self.criterion_identity = torch.nn.L1Loss()
        self.criterion_structure = torch.nn.L1Loss()
        self.lambda_cycle, self.lambda_identity, self.lambda_structure = 10.0, 2.0, 0.5
        oracle = R.CycleGANOracle(*[clone_sd(sd) for sd in sds])
        out = {"C": C, "shape": np.array(shape), "seeds": np.array([61, 62, 63, 64])}
        keys = ("d_loss", "g_loss", "cycle_loss", "identity_loss", "structure_loss")
        for step in range(3):
            a = R.make_input(shape, 700 + 2 * step)
            b = R.make_input(shape, 701 + 2 * step)
            lr = self.train_step(a, b)
            lo = oracle.train_step(a, b)
            print(f"  step {step}: ref {[round(lr[k], 6) for k in keys]}")
            print(f"          mine {[round(lo[k], 6) for k in keys]}")
            for k in keys:
                # step 0 is a pure function of the inputs; later steps inherit the +-lr sign noise of
                # Adam on zero-gradient elements (see below), which moves losses by a few 1e-4 relative.
                tol = 2e-5 if step == 0 else 2e-3
                assert abs(lr[k] - lo[k]) <= tol * max(1.0, abs(lr[k])), (step, k, lr[k], lo[k])
            out[f"losses_{step}"] = np.array([lr[k] for k in keys], dtype=np.float64)
            for name, m, sd in (("G_AB", self.G_AB, oracle.G_AB), ("G_BA", self.G_BA, oracle.G_BA),
                                ("D_A", self.D_A, oracle.D_A), ("D_B", self.D_B, oracle.D_B)):
                st = m.state_dict()
                for k in st:
                    if k.startswith("style_encoder"):
                        assert torch.equal(st[k], sds[0 if name == "G_AB" else 1][k])  # never updated
                delta = torch.cat([(st[k] - s0[k]).flatten() for s0 in [sds[("G_AB", "G_BA", "D_A", "D_B").index(name)]] for k in st])
                dmine = torch.cat([(sd[k].detach() - s0[k]).flatten() for s0 in [sds[("G_AB", "G_BA", "D_A", "D_B").index(name)]] for k in st])
                # Adam's very first update is lr*g/(|g|+eps) ~= lr*sign(g): elements whose gradient is
                # rounding noise (|g| <~ 1e-6, e.g. every conv bias in front of an InstanceNorm) get a
                # +-lr step of arbitrary sign, in the reference as much as here.  So parameters are
                # compared element-wise: all within 2.1*lr*(steps), and all but a few % within 5 % of lr.
                lr_ = 5e-5 if name.startswith("G") else 2e-4
                diff = (dmine - delta).abs()
                frac = float((diff > 0.05 * lr_).float().mean())
                print(f"          param-delta {name}: |d| {float(delta.norm()):.4e}  max|mine-ref| {float(diff.max()):.2e}  "
                      f"frac(>5% lr) {frac:.4f}")
                assert float(diff.max()) <= 2.1 * lr_ * (step + 1), (name, float(diff.max()))
                if step == 0:  # later steps: the early-Adam m/sqrt(v) ratio amplifies the 1e-4 loss drift
                    assert frac <= 0.05, (name, frac)
                out[f"delta_norm_{step}_{name}"] = float(delta.norm())
                out[f"param_sum_{step}_{name}"] = float(sum(v.double().sum() for v in st.values()))
        # final parameters of G_AB head + D_A first conv: small, exact vectors to compare against
        out["final_G_AB_output.0.weight"] = npy(self.G_AB.state_dict()["output.0.weight"])
        out["final_D_A_main.0.weight_orig"] = npy(self.D_A.state_dict()["main.0.weight_orig"])
        out["final_D_A_main.0.weight_u"] = npy(self.D_A.state_dict()["main.0.weight_u"])
    np.savez_compressed(os.path.join(GOLD, "train_step_c8_64x64.npz"), **out)


def assemble_reference_cyclegan(eg, et, C, sds, dtype):
    """An ``EnhancedCycleGAN`` of the reference built without its ctor (which hard-codes num_transformer_blocks=1,
    enhanced_train.py:18-19, and therefore needs the missing module): same attributes, hyper-parameters of :36-57."""
    import itertools
    self = et.EnhancedCycleGAN.__new__(et.EnhancedCycleGAN)
    self.device = torch.device("cpu")
    self.G_AB = eg.EnhancedGenerator(channels=C, num_transformer_blocks=0)
    self.G_BA = eg.EnhancedGenerator(channels=C, num_transformer_blocks=0)
    self.D_A = eg.EnhancedDiscriminator(channels=C)
    self.D_B = eg.EnhancedDiscriminator(channels=C)
    for m, sd in zip((self.G_AB, self.G_BA, self.D_A, self.D_B), sds):
        m.load_state_dict(sd)
        m.to(dtype)
    self.G_AB.gradient_checkpointing_enable()
    self.G_BA.gradient_checkpointing_enable()
    self.g_optimizer = torch.optim.Adam(itertools.chain(self.G_AB.parameters(), self.G_BA.parameters()), lr=5e-5, betas=(0.5, 0.999))
    self.d_optimizer = torch.optim.Adam(itertools.chain(self.D_A.parameters(), self.D_B.parameters()), lr=2e-4, betas=(0.5, 0.999))
    self.scaler = torch.cuda.amp.GradScaler()
    self.criterion_gan = torch.nn.MSELoss()
    self.criterion_cycle = torch.nn.L1Loss()
    self.criterion_identity = torch.nn.L1Loss()
    self.criterion_structure = torch.nn.L1Loss()
    self.lambda_cycle, self.lambda_identity, self.lambda_structure = 10.0, 2.0, 0.5
    self.captured = {}

    def hook(name):
        def h(opt, args, kwargs):  # gradients as the optimizer sees them, right before its step
            self.captured[name] = [None if p.grad is None else p.grad.detach().clone() for g in opt.param_groups for p in g["params"]]
        return h

    self.d_optimizer.register_step_pre_hook(hook("d"))
    self.g_optimizer.register_step_pre_hook(hook("g"))
    return self


LOSS_KEYS = ("d_loss", "g_loss", "cycle_loss", "identity_loss", "structure_loss")


def _force_state(dst, src, dtype):
    """Teacher forcing: copy models (parameters + spectral-norm u/v buffers) and both optimizers' states from ``src``."""
    for name in ("G_AB", "G_BA", "D_A", "D_B"):
        getattr(dst, name).load_state_dict({k: v.to(dtype) for k, v in getattr(src, name).state_dict().items()})
    for name in ("g_optimizer", "d_optimizer"):
        sd = getattr(src, name).state_dict()
        st = {i: {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() and k != "step" else (v.clone() if torch.is_tensor(v) else v))
                  for k, v in s.items()} for i, s in sd["state"].items()}
        getattr(dst, name).load_state_dict({"state": st, "param_groups": sd["param_groups"]})


def _tensor_dists(g32, g64):
    out = []
    for a, b in zip(g32, g64):
        if b is None:
            out.append(-1.0)
        else:
            out.append(float((a.double() - b).norm() / b.norm().clamp_min(1e-300)))
    return out


def _agg_dist(g32, g64, live):
    num = sum(float((a.double() - b).pow(2).sum()) for a, b, lv in zip(g32, g64, live) if lv)
    den = sum(float(b.pow(2).sum()) for b, lv in zip(g64, live) if lv)
    return (num / max(den, 1e-300)) ** 0.5


def run_forced(eg, et, C, shape, seeds, in_seed, steps=3):
    """Reference train_step for ``steps`` steps in fp64 (free running) and, beside it, in fp32 TEACHER-FORCED: before each
    step the fp32 instance receives the fp64 instance's models and optimizer states, so each fp32 step is the same function
    as the fp64 step evaluated in single precision.  Returns per step: fp64 losses / gradients, the fp32 run's distances
    from them, and the state (spectral-norm vectors) the step started from."""
    sds = [R.make_state_dict(R.generator_spec(C), seeds[0]), R.make_state_dict(R.generator_spec(C), seeds[1]),
           R.make_state_dict(R.discriminator_spec(C), seeds[2]), R.make_state_dict(R.discriminator_spec(C), seeds[3])]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r64 = assemble_reference_cyclegan(eg, et, C, sds, torch.float64)
        r32 = assemble_reference_cyclegan(eg, et, C, sds, torch.float32)
        g_names = [k for m in (r64.G_AB, r64.G_BA) for k, _ in m.named_parameters()]
        d_names = [k for m in (r64.D_A, r64.D_B) for k, _ in m.named_parameters()]
        rec = []
        for step in range(steps):
            a = R.make_input(shape, in_seed + 2 * step)
            b = R.make_input(shape, in_seed + 1 + 2 * step)
            _force_state(r32, r64, torch.float32)
            uv = {f"{dn}.{k}": v.detach().clone() for dn, D in (("D_A", r64.D_A), ("D_B", r64.D_B)) for k, v in D.state_dict().items()
                  if k.endswith(("_u", "_v"))}
            l32 = r32.train_step(a, b)
            l64 = r64.train_step(a.double(), b.double())
            g64, d64, g32, d32 = r64.captured["g"], r64.captured["d"], r32.captured["g"], r32.captured["d"]
            live_g = [x is not None and not R_dead_bias(n) for n, x in zip(g_names, g64)]
            live_d = [x is not None and not R_dead_bias(n) for n, x in zip(d_names, d64)]
            rec.append({"l64": l64, "l32": l32, "g64": g64, "d64": d64, "uv": uv,
                        "g32_dist": _tensor_dists(g32, g64), "d32_dist": _tensor_dists(d32, d64),
                        "g32_agg": _agg_dist(g32, g64, live_g), "d32_agg": _agg_dist(d32, d64, live_d)})
    return sds, g_names, d_names, rec


def R_dead_bias(name: str) -> bool:
    """Bias of a convolution feeding an InstanceNorm: exactly-zero gradient in exact arithmetic (the norm removes the mean)."""
    if not name.endswith(".bias"):
        return False
    stem = name[:-5]
    return (stem == "initial.0" or stem.endswith((".branch1.0", ".branch2.0", ".branch3.0", ".branch4.0", ".fusion.0"))
            or stem in ("down1.0", "down2.0", "up1.0", "up2.0", "main.2", "main.5", "main.8", "structure_head.0"))


def gen_train_step_fp64(eg, et):
    """Multi-step pin of the train step in fp64.  The reference's own fp32 gradients sit 1e-2 (step 0) to
    several 1e-1 (steps 1-2) from its fp64 gradients on a free-running trajectory: Adam's first updates are lr*sign(g), so
    every element whose gradient sign is rounding noise moves 2*lr apart between two correct implementations, and two chained
    generators (50 un-affine InstanceNorm+ReLU) turn that into ReLU-mask flips.  No seed/size makes steps 1-2 well conditioned
    (free-running scan over 160 draws: best 8e-3 / 6e-2 at steps 1 / 2).  What IS well conditioned is each step as a function of its starting
    state.  So the fixture holds, for 3 steps of the reference's unmodified train_step run in fp64: the five losses, every
    gradient both optimizers see, the spectral-norm vectors each step starts from -- and, per step and tensor, how far the
    reference's own fp32 evaluation of THE SAME step (teacher-forced to the fp64 state) is from the fp64 result.  The GPU test
    teacher-forces the HIP build the same way (parameters and Adam moments re-derived in fp64 from the stored gradients) and
    holds it to  max(1e-3, 1.5 x the reference's own fp32 distance)  per step, per loss and per live gradient tensor."""
    C = 8
    # candidates: the two draws of an 80-seed scan at 32x32 (seeds 300..616) whose three forced steps are all under 1e-3, one
    # typical draw per size for the record (they print how ill-conditioned the usual case is)
    cands = [((2, 3, 32, 32), s0) for s0 in (388, 512, 144)] + [((2, 3, 48, 48), 244), ((2, 3, 64, 64), 132)]
    best = None
    for shape, s0 in cands:
        seeds, ins = (s0, s0 + 1, s0 + 2, s0 + 3), 1000 + s0
        sds, g_names, d_names, rec = run_forced(eg, et, C, shape, seeds, ins)
        score = max(max(r["g32_agg"], r["d32_agg"]) for r in rec)
        print(f"  fp64 pin candidate {shape} seeds {seeds}: reference-fp32 forced distance per step "
              f"{[(round(r['d32_agg'], 6), round(r['g32_agg'], 6)) for r in rec]}")
        if best is None or score < best[0]:
            best = (score, shape, seeds, ins, sds, g_names, d_names, rec)
    score, shape, seeds, ins, sds, g_names, d_names, rec = best
    print(f"  chosen: {shape} seeds {seeds} input seed {ins}: worst forced reference-fp32 distance {score:.2e}")
    out = {"C": C, "shape": np.array(shape), "seeds": np.array(seeds), "in_seed": ins, "steps": len(rec),
           "g_names": np.array(g_names), "d_names": np.array(d_names)}
    for k, r in enumerate(rec):
        out[f"losses64_{k}"] = np.array([r["l64"][key] for key in LOSS_KEYS], dtype=np.float64)
        out[f"losses32_{k}"] = np.array([r["l32"][key] for key in LOSS_KEYS], dtype=np.float64)
        out[f"g32_dist_{k}"] = np.array(r["g32_dist"], dtype=np.float64)
        out[f"d32_dist_{k}"] = np.array(r["d32_dist"], dtype=np.float64)
        out[f"g32_agg_{k}"], out[f"d32_agg_{k}"] = r["g32_agg"], r["d32_agg"]
        for i, g in enumerate(r["g64"]):
            if g is not None:
                out[f"g64_{k}_{i}"] = g.to(torch.float32).numpy()
        for i, g in enumerate(r["d64"]):
            if g is not None:
                out[f"d64_{k}_{i}"] = g.to(torch.float32).numpy()
        for name, v in r["uv"].items():
            out[f"uv_{k}_{name}"] = v.to(torch.float32).numpy()
    # How far is a CORRECT fp32 evaluation from fp64 in general?  The draw above was selected because the reference's fp32 run
    # happens to be close on it (that is what makes the fixture a sharp pin for step 0 and the optimizer plumbing), so its
    # distances say little about what another fp32 implementation should reach on the same draw.  The distribution of the
    # reference's own forced fp32 distance over many draws of this shape does: it is stored beside the vectors and the GPU test
    # bounds the build by its median.
    n_scan = int(os.environ.get("MSTG_GOLDEN_SCAN", "80"))
    samples_g, samples_d = [], []
    for s0 in range(300, 300 + 4 * n_scan, 4):
        _, _, _, rec_s = run_forced(eg, et, C, shape, (s0, s0 + 1, s0 + 2, s0 + 3), 1000 + s0)
        samples_g += [r["g32_agg"] for r in rec_s]
        samples_d += [r["d32_agg"] for r in rec_s]
    out["scan_g32_agg"] = np.array(samples_g, dtype=np.float64)
    out["scan_d32_agg"] = np.array(samples_d, dtype=np.float64)
    q = np.percentile(out["scan_g32_agg"], [10, 50, 90])
    print(f"  reference fp32 vs fp64, forced steps, {n_scan} draws x {len(rec)} steps, generator gradients: "
          f"10 % {q[0]:.2e}  median {q[1]:.2e}  90 % {q[2]:.2e}")
    np.savez_compressed(os.path.join(GOLD, "train_step_fp64_c8.npz"), **out)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(GOLD, exist_ok=True)
    eg, et, pt = import_reference()
    print("reference imported from", REF, "| torch", torch.__version__)
    gen_ops(eg)
    gen_generator(eg)
    gen_discriminator(eg)
    gen_plain_generator(pt)
    gen_train_step(eg, et)
    gen_train_step_fp64(eg, et)
    with open(os.path.join(GOLD, "PROVENANCE.txt"), "w") as f:
        f.write(f"generated by oracle/make_golden.py from {REF} with torch {torch.__version__}, "
                f"{torch.get_num_threads()} threads, fp32 CPU; weights/inputs: oracle.restatement.make_state_dict/make_input\n")
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
