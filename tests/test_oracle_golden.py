"""CPU: the oracle (oracle/restatement.py) against the committed golden vectors, which were produced by the
reference's own modules (oracle/make_golden.py).  This is what pins the oracle on a box without /root/reference."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import restatement as R

TOL = 1e-5


def _load(gold_dir, name):
    return np.load(os.path.join(gold_dir, name))


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_local_attention(gold_dir, tag):
    g = _load(gold_dir, "ops.npz")
    ch, shape, seed = int(g[f"attn_{tag}_ch"]), tuple(g[f"attn_{tag}_shape"]), int(g[f"attn_{tag}_seed"])
    spec = [("qkv.weight", (3 * ch, ch, 1, 1)), ("qkv.bias", (3 * ch,)), ("proj.weight", (ch, ch, 1, 1)), ("proj.bias", (ch,))]
    sd = {"p." + k: v.requires_grad_(True) for k, v in R.make_state_dict(spec, seed).items()}
    x = R.make_input(shape, seed + 100).requires_grad_(True)
    y = R.local_attention(x, sd, "p", 4)
    gy = R.make_input(tuple(y.shape), seed + 200)
    grads = torch.autograd.grad((y * gy).sum(), [x] + list(sd.values()))
    assert rel_l2(y, _t(g[f"attn_{tag}_y"])) <= TOL
    assert rel_l2(grads[0], _t(g[f"attn_{tag}_dx"])) <= TOL
    for k, gr in zip(sd, grads[1:]):
        assert rel_l2(gr, _t(g[f"attn_{tag}_d_{k[2:]}"])) <= TOL, k


@pytest.mark.parametrize("tag", ["a", "b"])
def test_multi_scale_block(gold_dir, tag):
    g = _load(gold_dir, "ops.npz")
    ch, shape, seed = int(g[f"msb_{tag}_ch"]), tuple(g[f"msb_{tag}_shape"]), int(g[f"msb_{tag}_seed"])
    spec = [("branch1.0.weight", (ch // 4, ch, 1, 1)), ("branch1.0.bias", (ch // 4,))]
    for b in (2, 3, 4):
        spec += [(f"branch{b}.0.weight", (ch // 4, ch, 3, 3)), (f"branch{b}.0.bias", (ch // 4,))]
    spec += [("fusion.0.weight", (ch, ch, 1, 1)), ("fusion.0.bias", (ch,))]
    sd = {"p." + k: v.requires_grad_(True) for k, v in R.make_state_dict(spec, seed).items()}
    x = R.make_input(shape, seed + 100).requires_grad_(True)
    y = R.multi_scale_block(x, sd, "p")
    gy = R.make_input(tuple(y.shape), seed + 200)
    grads = torch.autograd.grad((y * gy).sum(), [x] + list(sd.values()))
    assert rel_l2(y, _t(g[f"msb_{tag}_y"])) <= TOL
    assert rel_l2(grads[0], _t(g[f"msb_{tag}_dx"])) <= 2e-5
    for k, gr in zip(sd, grads[1:]):
        if k.endswith("weight"):
            assert rel_l2(gr, _t(g[f"msb_{tag}_d_{k[2:]}"])) <= 2e-5, k


@pytest.mark.parametrize("tag", ["c8_32x48", "c16_64x64"])
def test_generator(gold_dir, tag):
    g = _load(gold_dir, f"generator_{tag}.npz")
    C, shape, seed = int(g["C"]), tuple(g["shape"]), int(g["seed"])
    sd = {k: v.requires_grad_(True) for k, v in R.make_state_dict(R.generator_spec(C), seed).items()}
    x = R.make_input(shape, seed + 100).requires_grad_(True)
    taps = {}
    y = R.generator_forward(sd, x, taps)
    for k in ("initial", "down1", "down2", "up1", "up2"):
        assert rel_l2(taps[k], _t(g["tap_" + k])) <= TOL, k
    assert rel_l2(taps["pre_tanh"], _t(g["pre_tanh"])) <= TOL
    assert rel_l2(y, _t(g["out"])) <= TOL
    names = [k for k in sd if not k.startswith("style_encoder")]
    grads = torch.autograd.grad(y.abs().mean(), [x] + [sd[k] for k in names])
    assert rel_l2(grads[0], _t(g["dx"])) <= 5e-5
    for k, gr in zip(names, grads[1:]):
        if k.endswith("weight"):
            assert rel_l2(gr, _t(g["d_" + k])) <= 1e-4, k


def test_discriminator(gold_dir):
    g = _load(gold_dir, "discriminator_c8_64x64.npz")
    C, shape, seed = int(g["C"]), tuple(g["shape"]), int(g["seed"])
    sd = R.make_state_dict(R.discriminator_spec(C), seed)
    names = [k for k in sd if not k.endswith(("_u", "_v"))]
    for k in names:
        sd[k].requires_grad_(True)
    x = R.make_input(shape, seed + 100)
    for it in (1, 2):
        xi = x.clone().requires_grad_(True)
        s, st = R.discriminator_forward(sd, xi, train=True)
        loss = ((s - 1.0) ** 2).mean() + st.abs().mean()
        grads = torch.autograd.grad(loss, [xi] + [sd[k] for k in names])
        assert rel_l2(s, _t(g[f"t{it}_score"])) <= TOL
        assert rel_l2(st, _t(g[f"t{it}_struct"])) <= TOL
        assert rel_l2(grads[0], _t(g[f"t{it}_dx"])) <= 5e-5
        for k, gr in zip(names, grads[1:]):
            if k.endswith("weight_orig"):
                assert rel_l2(gr, _t(g[f"t{it}_d_{k}"])) <= 1e-4, k
        for k in sd:
            if k.endswith(("_u", "_v")):
                assert rel_l2(sd[k], _t(g[f"t{it}_{k}"])) <= TOL, k
    with torch.no_grad():
        s, st = R.discriminator_forward(sd, x, train=False)
    assert rel_l2(s, _t(g["eval_score"])) <= TOL and rel_l2(st, _t(g["eval_struct"])) <= TOL
    with torch.no_grad():
        s1, _ = R.discriminator_forward(sd, x[:1], train=False)
    assert s1.dim() == 0  # .squeeze() of a batch of one (reference :274)


def test_plain_generator(gold_dir):
    g = _load(gold_dir, "plain_generator_c8_32x32.npz")
    C, shape, seed = int(g["C"]), tuple(g["shape"]), int(g["seed"])
    sd = R.make_state_dict(R.plain_generator_spec(C), seed)
    names = [k for k in sd if k.endswith(("weight", "bias"))]
    for k in names:
        sd[k].requires_grad_(True)
    x = R.make_input(shape, seed + 100).requires_grad_(True)
    y = R.plain_generator_forward(sd, x, train=True)
    grads = torch.autograd.grad(y.abs().mean(), [x] + [sd[k] for k in names])
    assert rel_l2(y, _t(g["train_out"])) <= TOL
    assert rel_l2(grads[0], _t(g["train_dx"])) <= 5e-5
    for k, gr in zip(names, grads[1:]):
        if k.endswith("weight"):
            assert rel_l2(gr, _t(g["d_" + k])) <= 1e-4, k
    for k in sd:
        if "running" in k:
            assert rel_l2(sd[k], _t(g["after_" + k])) <= TOL, k
        if "num_batches" in k:
            assert int(sd[k]) == int(g["after_" + k])
    with torch.no_grad():
        ye = R.plain_generator_forward(sd, x.detach(), train=False)
    assert rel_l2(ye, _t(g["eval_out"])) <= TOL


def test_train_step(gold_dir):
    """Three consecutive steps of the restated train step against the losses the reference's unmodified train_step
    produced.  Step 0 is tight; later steps inherit Adam's +-lr sign noise on zero-gradient elements (see
    oracle/make_golden.py) and are compared at 2e-3."""
    g = _load(gold_dir, "train_step_c8_64x64.npz")
    C, shape = int(g["C"]), tuple(g["shape"])
    seeds = [int(s) for s in g["seeds"]]
    sds = [R.make_state_dict(R.generator_spec(C), seeds[0]), R.make_state_dict(R.generator_spec(C), seeds[1]),
           R.make_state_dict(R.discriminator_spec(C), seeds[2]), R.make_state_dict(R.discriminator_spec(C), seeds[3])]
    model = R.CycleGANOracle(*sds)
    keys = ("d_loss", "g_loss", "cycle_loss", "identity_loss", "structure_loss")
    for step in range(3):
        out = model.train_step(R.make_input(shape, 700 + 2 * step), R.make_input(shape, 701 + 2 * step))
        ref = g[f"losses_{step}"]
        tol = 2e-5 if step == 0 else 2e-3
        for k, r in zip(keys, ref):
            assert abs(out[k] - r) <= tol * max(1.0, abs(r)), (step, k, out[k], r)
    w = model.G_AB["output.0.weight"].detach()
    assert float((w - _t(g["final_G_AB_output.0.weight"])).abs().max()) <= 2.1 * 5e-5 * 3


def test_train_step_fp64_pin(gold_dir):
    """The restated train step run in fp64, free-running for three steps, against the reference's unmodified train_step run in
    fp64 (tests/golden/train_step_fp64_c8.npz): losses to 1e-9, every gradient both optimizers see to 1e-6 (the fixture stores
    the fp64 gradients rounded to fp32), and the Adam state derived by tests/fp64_fixture.py equals the oracle's own."""
    from fp64_fixture import Fp64TrainStepFixture, LOSS_KEYS, dead_bias
    fx = Fp64TrainStepFixture(gold_dir)
    C, shape, seeds = fx.C, fx.shape, fx.seeds
    sds = [R.make_state_dict(R.generator_spec(C), seeds[0]), R.make_state_dict(R.generator_spec(C), seeds[1]),
           R.make_state_dict(R.discriminator_spec(C), seeds[2]), R.make_state_dict(R.discriminator_spec(C), seeds[3])]
    model = R.CycleGANOracle(*[{k: v.double() for k, v in sd.items()} for sd in sds])
    assert [k for _, k in model.g_keys] == fx.names["g"] and [k for _, k in model.d_keys] == fx.names["d"]
    got = {}
    for which, opt in (("d", model.d_opt), ("g", model.g_opt)):
        orig = opt.step

        def step(grads, _w=which, _orig=orig):
            got[_w] = grads
            _orig(grads)
        opt.step = step
    p0 = {"g": [sd[k].detach().clone() for sd, k in model.g_keys], "d": [sd[k].detach().clone() for sd, k in model.d_keys]}
    for k in range(fx.steps):
        a, b = R.make_input(shape, fx.in_seed + 2 * k).double(), R.make_input(shape, fx.in_seed + 1 + 2 * k).double()
        for which, keys in (("g", model.g_keys), ("d", model.d_keys)):  # state derived from the stored gradients == oracle's state
            p, _, _ = fx.state_at(which, k, p0[which])
            for (sd, key), pe in zip(keys, p):
                assert float((sd[key].detach() - pe).abs().max()) <= 1e-9, (k, which, key)
        out = model.train_step(a, b)
        ref = fx.losses64(k)
        for key in LOSS_KEYS:
            assert abs(out[key] - ref[key]) <= 1e-9 * max(1.0, abs(ref[key])), (k, key, out[key], ref[key])
        for which in ("g", "d"):
            for n, mine, r in zip(fx.names[which], got[which], fx.grads(which, k)):
                assert (mine is None) == (r is None), (which, n)
                if r is not None and not dead_bias(n):
                    assert rel_l2(mine, r.view_as(mine)) <= 1e-6, (k, which, n)
