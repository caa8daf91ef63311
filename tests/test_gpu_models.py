"""GPU parity of the drop-in modules and the training step against (a) the golden vectors produced by the
reference's own modules and (b) the oracle run live on the CPU with the same seeded weights/inputs.

Tolerance: north star = 1e-3 relative (fp32).  Forward taps are held to 1e-4, parameter gradients to 1e-3."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mstg_hip import _lib
    _lib.load()


def report(name, err, tol):
    print(f"  [parity] {name:60s} rel-L2 {err:.2e} (tol {tol:.0e})")
    assert err <= tol, f"{name}: {err:.3e} > {tol:.0e}"


def _t(a):
    return torch.from_numpy(np.asarray(a))


def dead_bias(name: str) -> bool:
    """Bias of a convolution that feeds an InstanceNorm: its gradient is exactly zero in exact arithmetic (IN removes
    the mean), so what any fp32 implementation -- the reference included -- produces there is rounding noise."""
    if not name.endswith(".bias"):
        return False
    stem = name[:-5]
    g_dead = stem == "initial.0" or stem.endswith((".branch1.0", ".branch2.0", ".branch3.0", ".branch4.0", ".fusion.0")) \
        or stem in ("down1.0", "down2.0", "up1.0", "up2.0")
    d_dead = stem in ("main.2", "main.5", "main.8", "structure_head.0")
    plain_dead = stem in ("encoder.2", "encoder.5", "encoder.8", "decoder.0", "decoder.3", "decoder.6")  # conv -> BatchNorm
    return g_dead or d_dead or plain_dead


def check_grads(tag, names, ours, refs, tol_each=5e-3, tol_all=1e-3, fp32_refs=None):
    """Per-tensor and aggregate gradient parity.

    Every tensor must be within ``tol_each`` and the concatenation of all of them within ``tol_all`` (the north-star
    1e-3).  The per-tensor bound is looser because ReLU / |.| are discontinuous in their derivative: one pre-activation
    within fp32 rounding of zero flips its mask between two correct fp32 implementations and moves that one channel's
    gradient by a few 1e-3 (tools/diag_grad_conditioning.py shows the HIP path and the CPU fp32 path both sit at the
    same distance from an fp64 run everywhere else).

    ``fp32_refs`` (the same gradients from the checker run in fp32, when ``refs`` are fp64): a tensor may then be as far from
    ``refs`` as 1.5x the fp32 checker itself is -- the bar for a network whose fp32 evaluation is ill-conditioned."""
    num = den = num32 = 0.0
    worst, worst_name = 0.0, ""
    scale = max(float(r.abs().max()) for n, r in zip(names, refs) if not dead_bias(n))
    for n, a, r in zip(names, ours, refs):
        a, r = a.detach().double().cpu(), r.detach().double().cpu()
        if dead_bias(n):
            assert float(a.abs().max()) <= 1e-4 * scale + 1e-6, (tag, n, float(a.abs().max()))
            continue
        num += float((a - r).pow(2).sum())
        den += float(r.pow(2).sum())
        e = float((a - r).norm() / r.norm().clamp_min(1e-30))
        if e > worst:
            worst, worst_name = e, n
        bound = tol_each
        if fp32_refs is not None:
            r32 = fp32_refs[names.index(n)].detach().double().cpu()
            num32 += float((r32 - r).pow(2).sum())
            bound = max(bound, 1.5 * float((r32 - r).norm() / r.norm().clamp_min(1e-30)))
        assert e <= bound, (tag, n, e, bound)
    total = (num / max(den, 1e-300)) ** 0.5
    if fp32_refs is not None:
        tol_all = max(tol_all, 1.5 * (num32 / max(den, 1e-300)) ** 0.5)
    print(f"  [parity] {tag:40s} all-gradients rel-L2 {total:.2e} (tol {tol_all:.0e}); worst tensor {worst_name} {worst:.2e} (tol {tol_each:.0e})")
    assert total <= tol_all, (tag, total)


def aggregate_distance(names, ours, refs):
    num = den = 0.0
    for n, a, r in zip(names, ours, refs):
        if dead_bias(n):
            continue
        a, r = a.detach().double().cpu().reshape(-1), r.detach().double().cpu().reshape(-1)
        num += float((a - r).pow(2).sum())
        den += float(r.pow(2).sum())
    return (num / max(den, 1e-300)) ** 0.5


def check_grads_over_draws(tag, run, extra_draws=6, tol_all=1e-3, **kw):
    """Gradient parity that does not depend on the luck of one draw.

    ``run(k)`` evaluates draw ``k`` (``k = 0``: the committed draw; others: the same weights on other inputs) and returns
    ``(names, ours, refs)`` or ``(names, ours, refs, fp32_refs)``.  The committed draw is held to ``check_grads``'s strict per-tensor and
    aggregate bars.  If it misses them, that is either an error or one ReLU mask flipped between two correct evaluations (a
    pre-activation within fp32 rounding of zero: the aggregate distance then jumps from ~1e-5 to 1e-3 .. 1e-2, and whether it happens
    on a given draw changes with ANY reordering of a sum -- measured in test_generator_same_with_every_norm_folding_switched_off: the
    distances are bimodal, about half of the draws flip).  The two are told apart over ``extra_draws`` more draws: a flip leaves at least
    two of the draws in the flip-free mode, agreeing within ``tol_all``, and none beyond 2e-2; an error shows on every draw."""
    out = run(0)
    names, ours, refs = out[:3]
    try:
        check_grads(tag, names, ours, refs, tol_all=tol_all, fp32_refs=out[3] if len(out) > 3 else None, **kw)
        return
    except AssertionError as first:
        msg = str(first)[:160]
    ds = [aggregate_distance(names, ours, refs)]
    for k in range(1, extra_draws + 1):
        out = run(k)
        ds.append(aggregate_distance(*out[:3]))
    print(f"  [parity] {tag}: the committed draw misses its strict bars ({msg}); aggregate distances over {len(ds)} draws: "
          + " ".join(f"{d:.1e}" for d in ds))
    srt = sorted(ds)
    assert srt[1] <= tol_all, (tag, "fewer than two draws agree in the flip-free mode", ds)
    assert srt[-1] <= 2e-2, (tag, "a draw is beyond what one flipped mask moves", ds)


def nchw(x):
    return x.permute(0, 3, 1, 2)


@pytest.mark.parametrize("tag", ["c8_32x48", "c16_64x64"])
@pytest.mark.parametrize("checkpointing", [False, True])
def test_generator_vs_reference_golden(gold_dir, tag, checkpointing):
    import enhanced_generator as eg
    from oracle import restatement as R
    g = np.load(os.path.join(gold_dir, f"generator_{tag}.npz"))
    C, shape, seed = int(g["C"]), tuple(g["shape"]), int(g["seed"])
    m = eg.EnhancedGenerator(channels=C, num_transformer_blocks=0)
    m.load_state_dict(R.make_state_dict(R.generator_spec(C), seed))
    m.to(DEV)
    if checkpointing:
        m.gradient_checkpointing_enable()
    x = R.make_input(shape, seed + 100).to(DEV).requires_grad_(True)
    taps = {}
    y = m.forward_taps(x, taps)
    for k in ("initial", "down1", "down2", "up1", "up2"):
        report(f"G[{tag}] ckpt={int(checkpointing)} tap {k}", rel_l2(nchw(taps[k]), _t(g["tap_" + k])), 1e-4)
    report(f"G[{tag}] pre_tanh", rel_l2(taps["pre_tanh"], _t(g["pre_tanh"])), 1e-4)
    report(f"G[{tag}] out", rel_l2(y, _t(g["out"])), 1e-4)
    names = [k for k, _ in m.named_parameters() if not k.startswith("style_encoder")]
    params = [p for k, p in m.named_parameters() if not k.startswith("style_encoder")]
    loss = y.abs().mean()
    grads = torch.autograd.grad(loss, [x] + params)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * float(g["loss"])
    sd = R.make_state_dict(R.generator_spec(C), seed)

    def run(k):  # k = 0: the reference's golden vectors; k > 0: the oracle (pinned to them) on another input, same weights
        if k == 0:
            return ["dx"] + names, grads, [_t(g["dx"])] + [_t(g["d_" + k_]) for k_ in names]
        xk = R.make_input(shape, seed + 100 + 1000 * k)
        sd_r = {k_: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k_, v in sd.items()}
        xr = xk.clone().requires_grad_(True)
        refs = torch.autograd.grad(R.generator_forward(sd_r, xr).abs().mean(), [xr] + [sd_r[k_] for k_ in names])
        xg = xk.to(DEV).requires_grad_(True)
        return ["dx"] + names, torch.autograd.grad(m(xg).abs().mean(), [xg] + params), refs

    check_grads_over_draws(f"G[{tag}] ckpt={int(checkpointing)}", run)


def test_generator_default_width_vs_oracle(monkeypatch):
    """The class-default width (channels=64, enhanced_generator.py:107): LocalAttention at C = 64, 128, 256 takes the
    row-blocked core above 64 channels.  Forward and every gradient against the oracle's autograd on the CPU."""
    import enhanced_generator as eg
    from oracle import restatement as R
    C, shape = 64, (1, 3, 32, 32)
    sd = R.make_state_dict(R.generator_spec(C), 31)
    m = eg.EnhancedGenerator(num_transformer_blocks=0)
    assert m.initial[0].out_channels == C
    m.load_state_dict(sd)
    m.to(DEV)
    names = [k for k, _ in m.named_parameters() if not k.startswith("style_encoder")]
    params = [p for k, p in m.named_parameters() if not k.startswith("style_encoder")]
    from mstg_hip import ops

    # The checker runs in fp64 here: at this width the fp32 CPU run itself sits 1.4e-2 from the exact gradients (ReLU-mask
    # flips, tools/diag_grad_c64.py), so fp32-vs-fp32 would measure the checker's noise.  The same holds between two correct
    # fp32 implementations: a 1e-7 change of one attention output (another summation order, tools/diag_attn_blk4_model.py)
    # grows to 1e-5 at the output and moves dx by 1e-2 through one flipped mask.  The bar is therefore "within 5e-3 of the
    # exact gradient, or no further from it than 1.5x the fp32 run of the reference's own arithmetic", per tensor -- on the
    # committed draw, and over more draws when that one happens to flip (check_grads_over_draws).
    def run(k, one_wave=False):
        x = R.make_input(shape, 32 + 1000 * k)

        def oracle_grads(dt):
            sd_r = {k_: (v.to(dt).requires_grad_(True) if v.is_floating_point() else v) for k_, v in sd.items()}
            xr = x.to(dt).requires_grad_(True)
            yr = R.generator_forward(sd_r, xr)
            return yr.detach(), torch.autograd.grad(yr.abs().mean(), [xr] + [sd_r[k_] for k_ in names])
        yr, refs = oracle_grads(torch.float64)
        if one_wave:  # forward through the one-wave attention core, default backward kernels (the switch is read per call)
            monkeypatch.setenv("MSTG_ATTN_BLK4", "0")
            ops.refresh_env()
        xg = x.to(DEV).requires_grad_(True)
        y = m(xg)
        if one_wave:
            monkeypatch.delenv("MSTG_ATTN_BLK4")
            ops.refresh_env()
        grads = torch.autograd.grad(y.abs().mean(), [xg] + params)
        report(f"G[c64_32x32] draw {k} out vs oracle(fp64)", rel_l2(y, yr), 1e-4)
        if one_wave:
            return ["dx"] + names, grads, refs
        return ["dx"] + names, grads, refs, oracle_grads(torch.float32)[1]

    check_grads_over_draws("G[c64_32x32] vs oracle(fp64)", run, extra_draws=4)
    check_grads_over_draws("G[c64_32x32] one-wave fwd, default bwd", lambda k: run(k, True), extra_draws=4)


def test_generator_no_grad_blocks1_eval_and_errors():
    """What the inference callers do: num_transformer_blocks=1, .eval(), torch.no_grad() (direct_transform.py:35-63)."""
    import enhanced_generator as eg
    from oracle import restatement as R
    sd = R.make_state_dict(R.generator_spec_with_blocks(16, 1), 7)
    m = eg.EnhancedGenerator(channels=16, num_transformer_blocks=1)
    m.load_state_dict(sd)  # strict, build-defined block included
    m.to(DEV).eval()
    x = R.make_input((1, 3, 128, 128), 8)
    with torch.no_grad():
        y = m(x.to(DEV))
        yr = R.generator_forward(sd, x, num_blocks=1)
    report("G blocks=1 eval 128x128 vs oracle", rel_l2(y, yr), 1e-4)
    assert y.shape == (1, 3, 128, 128) and float(y.abs().max()) <= 1.0
    for bad in ((1, 3, 250, 250), (1, 3, 248, 248), (1, 3, 128, 136 + 4)):
        with pytest.raises(RuntimeError):
            m(torch.zeros(bad, device=DEV))
    with pytest.raises(RuntimeError):
        m(torch.zeros((1, 4, 64, 64), device=DEV))
    # two runs, same bits (fixed-order reductions everywhere)
    with torch.no_grad():
        y2 = m(x.to(DEV))
    assert torch.equal(y, y2)


def test_generator_backward_is_deterministic():
    import enhanced_generator as eg
    from oracle import restatement as R
    m = eg.EnhancedGenerator(channels=8, num_transformer_blocks=0)
    m.load_state_dict(R.make_state_dict(R.generator_spec(8), 9))
    m.to(DEV)
    x = R.make_input((2, 3, 32, 32), 10).to(DEV)
    outs = []
    for _ in range(2):
        grads = torch.autograd.grad(m(x).abs().mean(), list(m.parameters()), allow_unused=True)
        outs.append([g.clone() for g in grads if g is not None])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_discriminator_vs_reference_golden(gold_dir):
    import enhanced_generator as eg
    from oracle import restatement as R
    g = np.load(os.path.join(gold_dir, "discriminator_c8_64x64.npz"))
    C, shape, seed = int(g["C"]), tuple(g["shape"]), int(g["seed"])
    m = eg.EnhancedDiscriminator(channels=C)
    m.load_state_dict(R.make_state_dict(R.discriminator_spec(C), seed))
    m.to(DEV).train()
    x = R.make_input(shape, seed + 100).to(DEV)
    names = [k for k, _ in m.named_parameters()]
    for it in (1, 2):
        xi = x.clone().requires_grad_(True)
        s, st = m(xi)
        loss = ((s - 1.0) ** 2).mean() + st.abs().mean()
        grads = torch.autograd.grad(loss, [xi] + list(m.parameters()))
        report(f"D train fwd#{it} score", rel_l2(s, _t(g[f"t{it}_score"])), 1e-4)
        report(f"D train fwd#{it} struct", rel_l2(st, _t(g[f"t{it}_struct"])), 1e-4)
        check_grads(f"D train fwd#{it}", ["dx"] + names, grads, [_t(g[f"t{it}_dx"])] + [_t(g[f"t{it}_d_{k}"]) for k in names])
        for k, v in m.state_dict().items():
            if k.endswith(("_u", "_v")):
                assert rel_l2(v, _t(g[f"t{it}_{k}"])) <= 1e-5, k
    m.eval()
    with torch.no_grad():
        s, st = m(x)
        s1, _ = m(x[:1])
    report("D eval score", rel_l2(s, _t(g["eval_score"])), 1e-4)
    report("D eval struct", rel_l2(st, _t(g["eval_struct"])), 1e-4)
    assert s.shape == (2,) and st.shape == (2, 1, 3, 3) and s1.dim() == 0


def test_plain_generator_vs_reference_golden(gold_dir):
    import plain_generator
    from oracle import restatement as R
    g = np.load(os.path.join(gold_dir, "plain_generator_c8_32x32.npz"))
    C, shape, seed = int(g["C"]), tuple(g["shape"]), int(g["seed"])
    m = plain_generator.Generator(channels=C)
    m.load_state_dict(R.make_state_dict(R.plain_generator_spec(C), seed))
    m.to(DEV).train()
    x = R.make_input(shape, seed + 100).to(DEV).requires_grad_(True)
    y = m(x)
    names = [k for k, _ in m.named_parameters()]
    grads = torch.autograd.grad(y.abs().mean(), [x] + list(m.parameters()))
    report("plain G train out", rel_l2(y, _t(g["train_out"])), 1e-4)
    check_grads("plain G train", ["dx"] + names, grads, [_t(g["train_dx"])] + [_t(g["d_" + k]) for k in names])
    for k, v in m.state_dict().items():
        if "running" in k:
            report(f"plain G {k}", rel_l2(v, _t(g["after_" + k])), 1e-4)
        if "num_batches" in k:
            assert int(v) == int(g["after_" + k])
    m.eval()
    with torch.no_grad():
        ye = m(x.detach())
    report("plain G eval out", rel_l2(ye, _t(g["eval_out"])), 1e-4)


def _build_cyclegan(C, seeds):
    import enhanced_train
    from oracle import restatement as R
    model = enhanced_train.EnhancedCycleGAN(channels=C, num_transformer_blocks=0, device=DEV)
    sds = [R.make_state_dict(R.generator_spec(C), seeds[0]), R.make_state_dict(R.generator_spec(C), seeds[1]),
           R.make_state_dict(R.discriminator_spec(C), seeds[2]), R.make_state_dict(R.discriminator_spec(C), seeds[3])]
    for m, sd in zip((model.G_AB, model.G_BA, model.D_A, model.D_B), sds):
        m.load_state_dict(sd)
    return model, sds


def test_train_step_vs_reference_golden(gold_dir):
    """Step 0 of the reference's unmodified fp32 train_step (C=8, 64x64, batch 2) against ours: a pure function of the inputs
    (1e-4; measured 3e-7).  Multi-step behaviour is pinned in fp64 by test_train_step_teacher_forced_vs_reference_fp64."""
    from oracle import restatement as R
    g = np.load(os.path.join(gold_dir, "train_step_c8_64x64.npz"))
    C, shape = int(g["C"]), tuple(g["shape"])
    model, sds = _build_cyclegan(C, [int(s) for s in g["seeds"]])
    keys = ("d_loss", "g_loss", "cycle_loss", "identity_loss", "structure_loss")
    a, b = R.make_input(shape, 700).to(DEV), R.make_input(shape, 701).to(DEV)
    out = model.train_step(a, b)
    ref = g["losses_0"]
    print(f"  [parity] train_step 0: ours {[round(out[k], 6) for k in keys]}  reference {[round(float(r), 6) for r in ref]}")
    for k, r in zip(keys, ref):
        assert abs(out[k] - r) <= 1e-4 * max(1.0, abs(r)), (k, out[k], r)
    for name, m, sd0 in (("G_AB", model.G_AB, sds[0]), ("D_A", model.D_A, sds[2])):
        st = m.state_dict()
        for k in st:
            if k.startswith("style_encoder"):
                assert torch.equal(st[k].cpu(), sd0[k]), k  # no gradient -> untouched, as with torch's Adam
        assert rel_l2(torch.tensor(float(sum((st[k].cpu() - sd0[k]).double().pow(2).sum() for k in st)) ** 0.5),
                      torch.tensor(float(g[f"delta_norm_0_{name}"]))) <= 2e-2


def test_train_step_teacher_forced_vs_reference_fp64(gold_dir):
    """Three steps of the train step against the reference's unmodified train_step run in fp64 (oracle/make_golden.py::
    gen_train_step_fp64), TEACHER-FORCED: every step starts from the fp64 trajectory's state (parameters and Adam moments
    re-derived in fp64 from the stored gradients, spectral-norm vectors from the fixture), so each step is held to its own
    conditioning instead of to the chaos of Adam's lr*sign(g) updates.

    What a correct fp32 implementation can reach here is a distribution, not a number: two chained generators flip ReLU masks,
    and the reference's OWN fp32 evaluation of a forced step sits 2e-4 (10th percentile) ... 5e-3 (median) ... 2e-2 (90th) from
    its fp64 evaluation over 80 draws x 3 steps of this shape (stored in the fixture, oracle/make_golden.py).  The fixture's draw
    was chosen because the reference is lucky on it (< 1e-3 at all three steps).  Bars:
      losses, every step            max(1e-3, 1.5 x the reference's own fp32 distance on this draw)
      gradients at step 0           aggregate max(1e-3, 1.5 x the reference's own fp32 distance on this draw); each tensor
                                    max(1e-3, 1.5 x that tensor's / the worst live tensor's reference distance on this draw)
      gradients at steps 1, 2       one draw cannot tell flip noise from a systematic error (the build need not share the
                                    reference's luck on this draw): the aggregate only has to lie inside the reference's own
                                    distribution (95th percentile of the scan); the sharp statement about these steps is
                                    test_train_step_gradient_distance_distribution below (32 fresh draws x 3 forced steps)
    and the test prints where the build's aggregate falls in the reference's distribution.  Also checked per step: Adam's first
    moment after the step is b1*m + (1-b1)*g for the gradient the optimizer actually saw (exactly linear: 1e-5), and the
    parameter update against the fp64 update on every element whose fp64 gradient is not rounding noise."""
    from fp64_fixture import BETAS, LOSS_KEYS, Fp64TrainStepFixture, dead_bias
    from oracle import restatement as R
    fx = Fp64TrainStepFixture(gold_dir)
    model, sds = _build_cyclegan(fx.C, fx.seeds)
    opts = {"g": model.g_optimizer, "d": model.d_optimizer}
    mods = {"g": (model.G_AB, model.G_BA), "d": (model.D_A, model.D_B)}
    for which in ("g", "d"):
        assert [n for m in mods[which] for n, _ in m.named_parameters()] == fx.names[which]
    p0 = {w: [p.detach().cpu().clone() for p in opts[w].params] for w in opts}
    captured = {}
    for which, opt in opts.items():
        orig = opt.step

        def step(_w=which, _opt=opt, _orig=orig):
            captured[_w] = _opt.grad.clone()
            _orig()
        opt.step = step

    def scatter(opt, flat, tensors):
        with torch.no_grad():
            for off, p, t in zip(opt.offsets, opt.params, tensors):
                flat[off:off + p.numel()].copy_(t.reshape(-1).to(torch.float32))

    worst = 0.0
    for k in range(fx.steps):
        expect = {}
        for which, opt in opts.items():  # force the fp64 trajectory's state
            p, m, v = fx.state_at(which, k, p0[which])
            scatter(opt, opt.flat, p)
            scatter(opt, opt.exp_avg, m)
            scatter(opt, opt.exp_avg_sq, v)
            opt.step_count = k
            expect[which] = (p, m, v)
        if k > 0:
            for name, vec in fx.uv(k).items():
                dn, key = name.split(".", 1)
                getattr(model, dn).state_dict()[key].copy_(vec)
        a = R.make_input(fx.shape, fx.in_seed + 2 * k).to(DEV)
        b = R.make_input(fx.shape, fx.in_seed + 1 + 2 * k).to(DEV)
        out = model.train_step(a, b)
        ref, ref32 = fx.losses64(k), fx.losses32(k)
        for key in LOSS_KEYS:
            bound = max(1e-3, 1.5 * abs(ref32[key] - ref[key]) / max(1.0, abs(ref[key])))
            err = abs(out[key] - ref[key]) / max(1.0, abs(ref[key]))
            assert err <= bound, (k, key, out[key], ref[key], bound)
        for which, opt in opts.items():
            ours = captured[which].cpu().double()
            dist32, agg32 = fx.ref32_dist(which, k)
            # the reference's own fp32 run is ONE draw of a flip-driven error (a 16-element bias moves by 1e-3 when one ReLU mask
            # flips upstream): a tensor may be as far out as 1.5x the reference's own worst live tensor of this step
            ref_worst = max(d for n_, d in zip(fx.names[which], dist32) if d >= 0 and not dead_bias(n_))
            scan = fx.scan(which)
            strict = k == 0 or which == "d"
            agg_bound = max(1e-3, 1.5 * agg32) if strict else max(1e-3, 1.5 * agg32, float(np.percentile(scan, 95)))
            num = den = 0.0
            rows = []
            p_before, m_before, _ = expect[which]
            m_flat = opt.exp_avg.detach().cpu().double()
            m_num = m_den = 0.0
            for i, (n, off, prm, r) in enumerate(zip(fx.names[which], opt.offsets, opt.params, fx.grads(which, k))):
                mine = ours[off:off + prm.numel()]
                if r is None:
                    assert float(mine.abs().max()) == 0.0, (k, n)
                    continue
                r = r.reshape(-1)
                if dead_bias(n):
                    continue
                e = float((mine - r).norm() / r.norm().clamp_min(1e-300))
                bound = max(1e-3, 1.5 * dist32[i], 1.5 * ref_worst) if strict else float("inf")
                worst = max(worst, e / bound)
                rows.append((e / bound, n, e, dist32[i], bound))
                num += float((mine - r).pow(2).sum())
                den += float(r.pow(2).sum())
                # Adam's first moment after the step: linear in the gradient the optimizer saw
                m_exp = BETAS[0] * m_before[i].reshape(-1) + (1 - BETAS[0]) * mine
                m_num += float((m_flat[off:off + prm.numel()] - m_exp).pow(2).sum())
                m_den += float(m_exp.pow(2).sum())
                # parameter update (checked below) on elements whose gradient is not rounding noise (|g| > 1e-2 rms)
                live = r.abs() > 1e-2 * float(r.pow(2).mean().sqrt())
                expect.setdefault(("live", which), []).append((i, off, prm.numel(), live))
            agg = (num / max(den, 1e-300)) ** 0.5
            pct = 100.0 * float(np.searchsorted(scan, agg)) / len(scan)
            print(f"  [parity] fp64-forced step {k} {which}-gradients: aggregate {agg:.2e} = percentile {pct:.0f} of the reference's own fp32 "
                  f"distances (this draw: {agg32:.2e}, its worst tensor {ref_worst:.2e}); bound {agg_bound:.2e}")
            for ratio, n, e, d32, bound in sorted(rows, reverse=True)[:4]:
                print(f"  [parity]     {n:28s} ours {e:.2e}  reference-fp32 {d32:.2e}  bound {bound:.2e}")
            for ratio, n, e, d32, bound in rows:
                assert e <= bound, (k, which, n, e, bound, d32)
            assert agg <= agg_bound, (k, which, agg, agg_bound)
            assert (m_num / max(m_den, 1e-300)) ** 0.5 <= 1e-5, (k, which, "exp_avg")
        # update direction / size: the state after this step vs the fp64 state at the start of the next one
        for which, opt in opts.items():
            p_next, _, _ = fx.state_at(which, k + 1, p0[which])
            after_flat = opt.flat.detach().cpu().double()
            p_before = expect[which][0]
            dn = dd = 0.0
            for i, off, n_el, live in expect[("live", which)]:
                d_ours = (after_flat[off:off + n_el] - p_before[i].reshape(-1))[live]
                d_ref = (p_next[i].reshape(-1) - p_before[i].reshape(-1))[live]
                dn += float((d_ours - d_ref).pow(2).sum())
                dd += float(d_ref.pow(2).sum())
            rel_upd = (dn / max(dd, 1e-300)) ** 0.5
            print(f"  [parity] fp64-forced step {k} {which}-update on live elements: rel-L2 {rel_upd:.2e}")
            assert rel_upd <= 5e-2, (k, which, rel_upd)
            expect.pop(("live", which))
    print(f"  [parity] fp64-forced: worst tensor at {worst:.2f} of its bound")


def test_train_step_gradient_distance_distribution():
    """Is the HIP train step's backward as close to exact arithmetic as a correct fp32 implementation can be?  One draw cannot
    answer that (ReLU / |.| masks flip under ANY change of fp32 summation order, and how many flip depends on the draw), so this
    test compares DISTRIBUTIONS: 32 fresh draws of weights and inputs (C=8, 64x64, batch 2) x 3 teacher-forced steps.  For every
    (draw, step) the oracle (oracle/restatement.py, pinned to the reference's own modules at 1e-5) runs the step in fp64 -- that
    trajectory supplies the state each step starts from (parameters, Adam moments, spectral-norm vectors) and the exact
    gradients -- and then, from that same state, (a) the oracle in fp32 on the host and (b) the HIP step; the sample is the
    aggregate relative L2 distance of the generator gradients from the fp64 ones (enhanced_train.py:59-131).

    Assertions (generator gradients; the discriminator gradients are held to the same floor / median / 90th-percentile statistics):
      * systematic error: a small systematic backward error would put a FLOOR under the HIP distances that a correct fp32
        implementation's well-conditioned draws do not have (the oracle-fp32's 10th percentile is ~2e-4: no mask flips there).
        The third-smallest of the 96 HIP samples must be <= max(3e-4, 1.5 x the oracle-fp32's third-smallest): a pin three times
        below the 1e-3 north-star tolerance that flip noise cannot fake and that does not depend on the luck of one draw;
      * spread: median and 90th percentile of HIP <= 1.25 x the oracle-fp32's, pooled (n = 96) and per step (n = 32) -- or inside
        the sampling noise of that ratio where 1.25 is below it: the two implementations flip DIFFERENT masks on the same draw,
        and the ratio of two sample quantiles of this heavy-tailed distribution (p90 / p10 ~ 100) scatters by more than 1.25x at
        these n.  The allowance is the 99.9th percentile of the same ratio under random relabelling of the paired samples (a
        paired permutation test with a fixed seed; every number is printed)."""
    from fp64_fixture import LOSS_KEYS
    from oracle import restatement as R
    C, shape, ndraw, nstep = 8, (2, 3, 64, 64), 32, 3
    model, _ = _build_cyclegan(C, [1, 2, 3, 4])
    opts = {"g": model.g_optimizer, "d": model.d_optimizer}
    nets = {"G_AB": model.G_AB, "G_BA": model.G_BA, "D_A": model.D_A, "D_B": model.D_B}
    names = {"g": [n for m in (model.G_AB, model.G_BA) for n, _ in m.named_parameters()],
             "d": [n for m in (model.D_A, model.D_B) for n, _ in m.named_parameters()]}
    captured = {}
    for which, opt in opts.items():
        def step(_w=which, _opt=opt):  # gradients only: every step starts from a forced state, nothing is updated here ...
            captured[_w] = _opt.grad.clone()
            if _w == "d":
                type(_opt).step(_opt)  # ... except the discriminator update INSIDE the step, which the generator loss sees
        opt.step = step

    def scatter(opt, flat, tensors):
        with torch.no_grad():
            for off, prm, t in zip(opt.offsets, opt.params, tensors):
                flat[off:off + prm.numel()].copy_(t.reshape(-1).to(torch.float32))

    def oracle_grads(orc, a, b, apply_g=False):
        got = {}
        d_step, g_step = orc.d_opt.step, orc.g_opt.step
        orc.d_opt.step = lambda grads: (got.__setitem__("d", [None if t is None else t.detach().clone() for t in grads]), d_step(grads))
        orc.g_opt.step = lambda grads: (got.__setitem__("g", [None if t is None else t.detach().clone() for t in grads]),
                                        g_step(grads) if apply_g else None)
        try:
            losses = orc.train_step(a, b)
        finally:
            orc.d_opt.step, orc.g_opt.step = d_step, g_step
        return got, losses

    def distance(which, ours, refs):
        num = den = 0.0
        for n, mine, r in zip(names[which], ours, refs):
            if r is None or dead_bias(n):
                continue
            num += float((mine.double().reshape(-1) - r.double().reshape(-1)).pow(2).sum())
            den += float(r.double().pow(2).sum())
        return (num / max(den, 1e-300)) ** 0.5

    dist = {("g", "hip"): [], ("g", "f32"): [], ("d", "hip"): [], ("d", "f32"): []}
    loss_err = 0.0
    for d in range(ndraw):
        seeds = [5000 + 10 * d + j for j in range(4)]
        sds = [R.make_state_dict(R.generator_spec(C), seeds[0]), R.make_state_dict(R.generator_spec(C), seeds[1]),
               R.make_state_dict(R.discriminator_spec(C), seeds[2]), R.make_state_dict(R.discriminator_spec(C), seeds[3])]
        o64 = R.CycleGANOracle(*[{k: v.double() for k, v in sd.items()} for sd in sds])
        assert [k for _, k in o64.g_keys] == names["g"] and [k for _, k in o64.d_keys] == names["d"]
        for k in range(nstep):
            a, b = R.make_input(shape, 7000 + 100 * d + 2 * k), R.make_input(shape, 7001 + 100 * d + 2 * k)
            # the state this step starts from, as the fp64 trajectory holds it
            sd64 = [{key: v.detach().clone() for key, v in sd.items()} for sd in (o64.G_AB, o64.G_BA, o64.D_A, o64.D_B)]
            adam = {w: ([t.clone() for t in o.m], [t.clone() for t in o.v], list(o.t)) for w, o in (("g", o64.g_opt), ("d", o64.d_opt))}
            # (a) oracle fp32 from that state
            o32 = R.CycleGANOracle(*[{key: v.float() for key, v in sd.items()} for sd in sd64])
            for w, o in (("g", o32.g_opt), ("d", o32.d_opt)):
                o.m, o.v, o.t = [t.float() for t in adam[w][0]], [t.float() for t in adam[w][1]], list(adam[w][2])
            g32, _ = oracle_grads(o32, a, b)
            # (b) HIP from that state
            for (name, net), sd in zip(nets.items(), sd64):
                st = net.state_dict()
                with torch.no_grad():
                    for key, v in sd.items():
                        st[key].copy_(v.to(torch.float32))
            for w, opt in opts.items():
                scatter(opt, opt.exp_avg, adam[w][0])
                scatter(opt, opt.exp_avg_sq, adam[w][1])
                opt.step_count = k
            out = model.train_step(a.to(DEV), b.to(DEV))
            # exact: this advances the fp64 trajectory
            g64, l64 = oracle_grads(o64, a.double(), b.double(), apply_g=True)
            loss_err = max(loss_err, max(abs(out[key] - l64[key]) / max(1.0, abs(l64[key])) for key in LOSS_KEYS))
            for w, opt in opts.items():
                ours = captured[w].cpu()
                mine = [ours[off:off + prm.numel()] for off, prm in zip(opt.offsets, opt.params)]
                dist[(w, "hip")].append((k, distance(w, mine, g64[w])))
                dist[(w, "f32")].append((k, distance(w, g32[w], g64[w])))
        print(f"  [parity] draw {d:2d}: g-gradient distance from fp64 per step  HIP " + " ".join(f"{v:.1e}" for _, v in dist[("g", "hip")][-nstep:])
              + "   oracle-fp32 " + " ".join(f"{v:.1e}" for _, v in dist[("g", "f32")][-nstep:]), flush=True)

    def q(vals, pct):
        return float(np.percentile(np.asarray(vals, dtype=np.float64), pct))

    def allowance(hip, f32, pct, rng):
        """99.9th percentile of quantile(hip') / quantile(f32') over random swaps of the paired samples"""
        hip, f32 = np.asarray(hip), np.asarray(f32)
        r = []
        for _ in range(5000):
            sw = rng.random(len(hip)) < 0.5
            r.append(np.percentile(np.where(sw, f32, hip), pct) / np.percentile(np.where(sw, hip, f32), pct))
        return float(np.percentile(r, 99.9))

    rng = np.random.default_rng(1234)
    print(f"  [parity] distribution test: worst loss distance from fp64 over {ndraw * nstep} forced steps {loss_err:.2e}")
    assert loss_err <= 5e-3
    hip_all, f32_all = sorted(v for _, v in dist[("g", "hip")]), sorted(v for _, v in dist[("g", "f32")])
    print(f"  [parity] g-gradient distance from fp64, three smallest of {len(hip_all)}: HIP {hip_all[:3]}  oracle-fp32 {f32_all[:3]}")
    assert hip_all[2] <= max(3e-4, 1.5 * f32_all[2]), ("floor under the HIP distances", hip_all[:3], f32_all[:3])
    for sel, tag in [(None, "pooled")] + [(k, f"step {k}") for k in range(nstep)]:
        hip = [v for kk, v in dist[("g", "hip")] if sel is None or kk == sel]
        f32 = [v for kk, v in dist[("g", "f32")] if sel is None or kk == sel]
        line = f"  [parity] g-gradient distance from fp64, {tag:7s} (n={len(hip)}):"
        for pct in (10, 50, 90):
            line += f"  p{pct} HIP {q(hip, pct):.2e} / oracle-fp32 {q(f32, pct):.2e}"
        print(line)
        for pct in (50, 90):
            ratio, allow = q(hip, pct) / q(f32, pct), max(1.25, allowance(hip, f32, pct, rng))
            print(f"  [parity]     p{pct} ratio {ratio:.2f} (bar 1.25; sampling allowance at this n {allow:.2f})")
            assert ratio <= allow, (tag, pct, ratio, allow)
    # the discriminator gradients: no ReLU masks of their own to flip at these sizes (LeakyReLU's two slopes, a 16x16 bottleneck),
    # but the fakes they are evaluated on come out of the generators' fp32 forwards, so a rare sample still moves: same statistics
    dh, d3 = [v for _, v in dist[("d", "hip")]], [v for _, v in dist[("d", "f32")]]
    line = "  [parity] d-gradient distance from fp64 (n=%d):" % len(dh)
    for pct in (10, 50, 90):
        line += f"  p{pct} HIP {q(dh, pct):.2e} / oracle-fp32 {q(d3, pct):.2e}"
    print(line + f"   largest three HIP {sorted(dh)[-3:]}  oracle-fp32 {sorted(d3)[-3:]}")
    for pct in (50, 90):
        ratio, allow = q(dh, pct) / q(d3, pct), max(1.25, allowance(dh, d3, pct, rng))
        print(f"  [parity]     d p{pct} ratio {ratio:.2f} (bar 1.25; sampling allowance at this n {allow:.2f})")
        assert ratio <= allow, ("d", pct, ratio, allow)
    assert sorted(dh)[2] <= max(3e-5, 1.5 * sorted(d3)[2]), ("floor under the HIP d-distances", sorted(dh)[:3], sorted(d3)[:3])
    assert max(dh) <= 2e-2  # sanity: a flipped sample, not a broken one


def test_train_step_gradients_vs_oracle():
    """First-step gradients of both optimizers against the oracle's autograd on the CPU (C=8, 32x32, batch 2) --
    isolates the backward path from Adam's sign amplification."""
    from oracle import restatement as R
    C, shape = 8, (2, 3, 32, 32)

    def run(k):
        model, sds = _build_cyclegan(C, [81, 82, 83, 84])
        a, b = R.make_input(shape, 90 + 1000 * k), R.make_input(shape, 91 + 1000 * k)
        captured = {}
        model.d_optimizer.step = lambda: captured.__setitem__("d", model.d_optimizer.grad.clone())
        model.g_optimizer.step = lambda: captured.__setitem__("g", model.g_optimizer.grad.clone())
        losses = model.train_step(a.to(DEV), b.to(DEV))
        # oracle, with optimizer steps disabled the same way
        oracle = R.CycleGANOracle(*[{k_: v.clone() for k_, v in sd.items()} for sd in sds])
        og = {}
        oracle.d_opt.step = lambda grads: og.__setitem__("d", grads)
        oracle.g_opt.step = lambda grads: og.__setitem__("g", grads)
        lo = oracle.train_step(a, b)
        for k_ in lo:
            assert abs(losses[k_] - lo[k_]) <= 1e-4 * max(1.0, abs(lo[k_])), (k_, losses[k_], lo[k_])

        def flat_of(grads, keys):
            return [torch.zeros_like(sd[k_]).flatten() if gr is None else gr.flatten() for (sd, k_), gr in zip(keys, grads)]

        res = {}
        for which, opt, keys in (("d", model.d_optimizer, oracle.d_keys), ("g", model.g_optimizer, oracle.g_keys)):
            ref_list = flat_of(og[which], keys)
            ours = captured[which].cpu()
            # our flat layout follows module.parameters() order == state-dict parameter order used by the oracle
            names = [n for m in ((model.D_A, model.D_B) if which == "d" else (model.G_AB, model.G_BA)) for n, _ in m.named_parameters()]
            assert names == [k_ for _, k_ in keys], "parameter order differs between the module and the oracle"
            mine = [ours[off:off + p.numel()] for off, p in zip(opt.offsets, opt.params)]
            keep = [i for i, n in enumerate(names) if not n.startswith("style_encoder")]  # no gradient at all (reference: None)
            for i, n in enumerate(names):
                if n.startswith("style_encoder"):
                    assert float(mine[i].abs().max()) == 0.0, n
            res[which] = ([names[i] for i in keep], [mine[i] for i in keep], [ref_list[i] for i in keep])
        return res

    cache = {}

    def run_which(which):
        def f(k):
            if k not in cache:
                cache[k] = run(k)
            return cache[k][which]
        return f

    for which in ("d", "g"):
        check_grads_over_draws(f"train_step first-step {which}-gradients", run_which(which))


def test_train_step_with_style_loss_vs_oracle():
    """Extended step (build-defined multi-style Gram loss attached, parity unpinned vs the reference): generator gradients
    against the oracle's autograd of the same definition."""
    from oracle import restatement as R
    C, shape, div = 8, (2, 3, 32, 32), 4  # Gram needs channel counts that are multiples of 16

    def run(k):
        model, sds = _build_cyclegan(C, [81, 82, 83, 84])
        refs = [R.make_input(shape, 95 + kk + 1000 * k) for kk in range(3)]
        model.attach_style_loss(refs, (0.5, 0.3, 0.2), lambda_style=3.0, width_div=div)
        vgg_sd = {k_: v.detach().cpu().clone() for k_, v in model.style_loss.features.state_dict().items()}
        a, b = R.make_input(shape, 90 + 1000 * k), R.make_input(shape, 91 + 1000 * k)
        captured = {}
        model.g_optimizer.step = lambda: captured.__setitem__("g", model.g_optimizer.grad.clone())
        model.d_optimizer.step = lambda: None
        losses = model.train_step(a.to(DEV), b.to(DEV))
        assert "style_loss" in losses and losses["style_loss"] > 0
        # oracle: same step with the extra term on fake_A = G_BA(real_B)
        g_ab = {k_: v.clone().requires_grad_(True) for k_, v in sds[0].items()}
        g_ba = {k_: v.clone().requires_grad_(True) for k_, v in sds[1].items()}
        d_a, d_b = {k_: v.clone() for k_, v in sds[2].items()}, {k_: v.clone() for k_, v in sds[3].items()}
        fake_B, fake_A = R.generator_forward(g_ab, a), R.generator_forward(g_ba, b)
        for sd_, inp in ((d_a, a), (d_b, b), (d_a, fake_A.detach()), (d_b, fake_B.detach())):  # D phase: 4 power iterations
            R.discriminator_forward(sd_, inp)
        idt = (R.l1(R.generator_forward(g_ba, a), a) + R.l1(R.generator_forward(g_ab, b), b)) * 2.0
        fa, _ = R.discriminator_forward(d_a, fake_A)
        fb, _ = R.discriminator_forward(d_b, fake_B)
        gl = R.mse(fa, 1.0) + R.mse(fb, 1.0)
        cyc = (R.l1(R.generator_forward(g_ba, fake_B), a) + R.l1(R.generator_forward(g_ab, fake_A), b)) * 10.0
        _, ras = R.discriminator_forward(d_a, a)
        _, fas = R.discriminator_forward(d_a, fake_A)
        _, rbs = R.discriminator_forward(d_b, b)
        _, fbs = R.discriminator_forward(d_b, fake_B)
        st = (R.l1(ras, fas) + R.l1(rbs, fbs)) * 0.5
        sty = R.multi_style_gram_loss(vgg_sd, fake_A, refs, [0.5, 0.3, 0.2]) * 3.0
        names = [k_ for k_ in g_ab if not k_.startswith("style_encoder")]
        grads = torch.autograd.grad(gl + cyc + idt + st + sty, [g_ab[k_] for k_ in names] + [g_ba[k_] for k_ in names])
        assert abs(losses["style_loss"] - float(sty)) <= 1e-4 * max(1.0, abs(float(sty)))
        ours = captured["g"].cpu()
        all_names = [n for m in (model.G_AB, model.G_BA) for n, _ in m.named_parameters()]
        mine = {}
        for off, p, n, idx in zip(model.g_optimizer.offsets, model.g_optimizer.params, all_names, range(len(all_names))):
            mine[(idx >= len(all_names) // 2, n)] = ours[off:off + p.numel()].view(p.shape)
        keys = [(False, k_) for k_ in names] + [(True, k_) for k_ in names]
        return [k_ for _, k_ in keys], [mine[k_] for k_ in keys], list(grads)

    check_grads_over_draws("train_step + style loss g-gradients", run)


def test_train_step_schedules_are_bit_identical():
    """Two side streams vs one, gradient checkpointing on vs off: same losses and same parameters, bit for bit, after three
    steps (every reduction has a fixed order; each parameter's gradient is only ever written from one stream)."""
    import enhanced_train

    def run(ckpt, streams):
        torch.manual_seed(3)
        m = enhanced_train.EnhancedCycleGAN(channels=16, num_transformer_blocks=0, device=torch.device(DEV), gradient_checkpointing=ckpt)
        m.two_streams = streams
        g = torch.Generator().manual_seed(5)
        a = (torch.rand((4, 3, 64, 64), generator=g) * 2 - 1).to(DEV)
        b = (torch.rand((4, 3, 64, 64), generator=g) * 2 - 1).to(DEV)
        out = [m.train_step(a, b) for _ in range(3)]
        return out, m.g_optimizer.flat.clone(), m.d_optimizer.flat.clone()

    ref = run(False, False)
    for ckpt, streams in ((False, True), (True, False), (True, True)):
        o = run(ckpt, streams)
        for i in range(3):
            assert o[0][i] == ref[0][i], (ckpt, streams, i, o[0][i], ref[0][i])
        assert torch.equal(o[1], ref[1]) and torch.equal(o[2], ref[2]), (ckpt, streams)


@pytest.mark.parametrize("env", ["MSTG_ATTN_UNFUSED=1", "MSTG_TORCH_SPECTRAL_NORM=1", "MSTG_SN_GROUP=0", "MSTG_D_SKIP_DEAD_HEADS=1", "MSTG_STREAMS=4",
                                 "MSTG_STREAMS=0",
                                 "MSTG_IGEMM=l", "MSTG_IGEMM=h", "MSTG_STREAM=1"])
def test_module_level_switches_keep_parity(env, gold_dir, monkeypatch):
    """The module-level switches of INTEGRATION.md section 3 (unfused attention, torch's spectral-norm hook, stream counts, conv
    kernel families) against the same golden vectors as the defaults."""
    k, v = env.split("=")
    monkeypatch.setenv(k, v)
    from mstg_hip import ops
    ops.refresh_env()
    test_generator_vs_reference_golden(gold_dir, "c16_64x64", False)
    test_discriminator_vs_reference_golden(gold_dir)
    test_train_step_vs_reference_golden(gold_dir)


def test_save_models_round_trip(tmp_path):
    """save_models writes the reference's three checkpoint files (enhanced_train.py:133-152) holding ONLY each model's own
    tensors (parameters are views of the optimizers' flat buffers: the files must not carry the whole buffer), a fresh model
    strict-loads them, and the files load the way the reference's inference scripts do (advanced_transform.py:14-32)."""
    import enhanced_generator as eg
    import enhanced_train
    model, _ = _build_cyclegan(8, [11, 12, 13, 14])
    g = torch.Generator().manual_seed(3)
    a = (torch.rand((1, 3, 32, 32), generator=g) * 2 - 1).to(DEV)
    b = (torch.rand((1, 3, 32, 32), generator=g) * 2 - 1).to(DEV)
    model.train_step(a, b)
    model.save_models(tmp_path, 20)
    n_g = sum(v.numel() for v in model.G_AB.state_dict().values())
    size = (tmp_path / "G_AB_epoch_20.pth").stat().st_size
    assert size < 4 * n_g + 65536, f"G_AB checkpoint is {size} bytes for {n_g} floats: it carries more than its own tensors"
    fresh = enhanced_train.EnhancedCycleGAN(channels=8, num_transformer_blocks=0, device=torch.device(DEV))
    assert fresh.load_models(tmp_path, 20) == 20
    for m_old, m_new in ((model.G_AB, fresh.G_AB), (model.G_BA, fresh.G_BA), (model.D_A, fresh.D_A), (model.D_B, fresh.D_B)):
        for (k, v), (k2, v2) in zip(m_old.state_dict().items(), m_new.state_dict().items()):
            assert k == k2 and torch.equal(v, v2), k
    fresh.g_optimizer.check_views()
    fresh.d_optimizer.check_views()
    la, lb = model.train_step(a, b), fresh.train_step(a, b)  # Adam moments are not part of the reference's checkpoints
    assert la["cycle_loss"] == lb["cycle_loss"] and la["identity_loss"] == lb["identity_loss"]
    ckpt = torch.load(tmp_path / "G_AB_epoch_20.pth", map_location="cpu", weights_only=True)
    assert set(ckpt) == {"epoch", "G_AB_state_dict"} and ckpt["epoch"] == 20
    g2 = eg.EnhancedGenerator(channels=8, num_transformer_blocks=0)
    g2.load_state_dict(ckpt["G_AB_state_dict"])
    d = torch.load(tmp_path / "discriminators_epoch_20.pth", map_location="cpu", weights_only=True)
    assert set(d) == {"epoch", "D_A_state_dict", "D_B_state_dict"}
    # moving a module after its optimizer was built is caught, not silently ignored
    model.G_AB.half()
    with pytest.raises(RuntimeError, match="flat buffer"):
        model.g_optimizer.step()
