"""BUILD-DEFINED StructuralTransformerBlock (structural_transformer.py; csrc/transformer.hip) through the C ABI, against the
oracle's restatement of the same definition and against plain torch for the generic pieces (LayerNorm, softmax attention).
PARITY UNPINNED: the reference's file is missing from its snapshot; what is checked is kernel == definition."""
import math

import pytest
import torch
import torch.nn.functional as F

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
    print(f"  [parity] {name:64s} rel-L2 {err:.2e} (tol {tol:.0e})")
    assert err <= tol, f"{name}: {err:.3e} > {tol:.0e}"


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def test_structure_map_vs_oracle():
    from mstg_hip import ops
    from oracle import restatement as R
    x = R.make_input((2, 3, 32, 48), 1)
    s = ops.structure_map(x.to(DEV))
    report("structure_map", rel_l2(s.reshape(2, -1, 4), R.structure_map(x)), 1e-6)


@pytest.mark.parametrize("N,L,dim,mod", [(2, 64, 32, True), (1, 200, 64, True), (2, 96, 256, False), (1, 33, 128, True)])
def test_layer_norm_mod_vs_torch(N, L, dim, mod):
    from mstg_hip import ops
    x, ga, be = rnd((N, L, dim), 1) * 2 + 0.3, 1 + 0.1 * rnd((dim,), 2), 0.1 * rnd((dim,), 3)
    gm, bm = (0.2 * rnd((N, dim), 4), 0.2 * rnd((N, dim), 5)) if mod else (None, None)
    gy = rnd((N, L, dim), 6)
    leaves = [t.clone().requires_grad_(True) for t in (x, ga, be)] + ([t.clone().requires_grad_(True) for t in (gm, bm)] if mod else [])
    ref = F.layer_norm(leaves[0], (dim,), leaves[1], leaves[2], 1e-5)
    if mod:
        ref = ref * (1 + leaves[3][:, None]) + leaves[4][:, None]
    gr = torch.autograd.grad((ref * gy).sum(), leaves)
    dl = [t.to(DEV).requires_grad_(True) for t in (x, ga, be)] + ([t.to(DEV).requires_grad_(True) for t in (gm, bm)] if mod else [])
    y = ops.layer_norm_mod(dl[0], dl[1], dl[2], dl[3] if mod else None, dl[4] if mod else None)
    gg = torch.autograd.grad((y * gy.to(DEV)).sum(), dl)
    report(f"ln_mod dim{dim} y", rel_l2(y, ref), 1e-5)
    for name, a, b in zip(("dx", "dgamma", "dbeta", "dgmod", "dbmod"), gg, gr):
        report(f"ln_mod dim{dim} {name}", rel_l2(a, b), 2e-5)


@pytest.mark.parametrize("N,L,heads,D", [(2, 64, 4, 8), (1, 96, 4, 16), (1, 200, 2, 32), (1, 130, 4, 64), (1, 1024, 4, 16)])
def test_flash_attention_vs_torch(N, L, heads, D):
    from mstg_hip import ops
    dim = heads * D
    qkv = rnd((N, L, 3 * dim), 1, 0.8)
    gy = rnd((N, L, dim), 2)

    def ref_attn(t):
        q, k, v = (u.reshape(N, L, heads, D).transpose(1, 2) for u in t.chunk(3, dim=-1))
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), dim=-1) @ v
        return a.transpose(1, 2).reshape(N, L, dim)

    tr = qkv.clone().double().requires_grad_(True)
    yr = ref_attn(tr)
    (gr,) = torch.autograd.grad((yr * gy.double()).sum(), [tr])
    tg = qkv.to(DEV).requires_grad_(True)
    y = ops.flash_attention(tg, heads)
    (gg,) = torch.autograd.grad((y * gy.to(DEV)).sum(), [tg])
    report(f"flash attention L{L} heads{heads} D{D} out", rel_l2(y, yr), 1e-5)
    report(f"flash attention L{L} heads{heads} D{D} dqkv", rel_l2(gg, gr), 2e-5)


@pytest.mark.parametrize("C_,shape", [(8, (2, 3, 32, 32)), (16, (1, 3, 64, 48))])
def test_block_vs_oracle(C_, shape):
    from oracle import restatement as R
    import structural_transformer as stx
    dim = 4 * C_
    sd = R.make_state_dict(R.transformer_block_spec(dim, "b"), 11)
    sd["b.style_mod.weight"] = 0.05 * rnd(tuple(sd["b.style_mod.weight"].shape), 12)  # the module zero-initialises it
    blk = stx.StructuralTransformerBlock(dim)
    blk.load_state_dict({k[2:]: v for k, v in sd.items()})
    blk.to(DEV)
    N, L = shape[0], (shape[2] // 4) * (shape[3] // 4)
    x, style, img = rnd((N, L, dim), 13), rnd((N, dim), 14).abs(), R.make_input(shape, 15)
    gy = rnd((N, L, dim), 16)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr, sr = x.clone().requires_grad_(True), style.clone().requires_grad_(True)
    yr = R.structural_transformer_block(sdr, "b", xr, sr, img)
    names = list(sd)
    gr = torch.autograd.grad((yr * gy).sum(), [xr, sr] + [sdr[k] for k in names])
    xg, sg = x.to(DEV).requires_grad_(True), style.to(DEV).requires_grad_(True)
    y = blk(xg, sg, img.to(DEV))
    params = dict(blk.named_parameters())
    gg = torch.autograd.grad((y * gy.to(DEV)).sum(), [xg, sg] + [params[k[2:]] for k in names])
    report(f"block dim{dim} out", rel_l2(y, yr), 2e-5)
    for name, a, b in zip(["dx", "dstyle"] + names, gg, gr):
        report(f"block dim{dim} d {name}", rel_l2(a, b), 1e-4)


@pytest.mark.parametrize("C_,shape", [(8, (2, 3, 32, 32)), (16, (1, 3, 64, 64))])
def test_generator_with_block_vs_oracle(C_, shape):
    """EnhancedGenerator(num_transformer_blocks=1) -- what every caller of the reference builds -- forward and all gradients
    (style encoder included: it is live now, enhanced_generator.py:142-147,216) against the oracle; strict state_dict round trip."""
    import enhanced_generator as eg
    from oracle import restatement as R
    spec = R.generator_spec_with_blocks(C_, 1)
    sd = R.make_state_dict(spec, 21)
    sd["transformer_blocks.0.style_mod.weight"] = 0.05 * rnd(tuple(sd["transformer_blocks.0.style_mod.weight"].shape), 22)
    m = eg.EnhancedGenerator(channels=C_, num_transformer_blocks=1)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == spec
    m.load_state_dict(sd)  # strict
    m.to(DEV)
    x = R.make_input(shape, 23)
    xg = x.to(DEV).requires_grad_(True)
    taps = {}
    y = m.forward_taps(xg, taps)
    names = [k for k, _ in m.named_parameters()]
    grads = torch.autograd.grad(y.abs().mean(), [xg] + list(m.parameters()))
    sdr = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
    xr = x.double().requires_grad_(True)
    tr = {}
    yr = R.generator_forward(sdr, xr, tr, num_blocks=1)
    gr = torch.autograd.grad(yr.abs().mean(), [xr] + [sdr[k] for k in names])
    report(f"G[blocks=1, C={C_}] pre_tanh", rel_l2(taps["pre_tanh"], tr["pre_tanh"]), 1e-4)
    report(f"G[blocks=1, C={C_}] out", rel_l2(y, yr), 1e-4)
    with torch.no_grad():
        sv = m._style_vector(taps["down2"])
    report(f"G[blocks=1, C={C_}] style vector (a9)", rel_l2(sv, R.style_encoder(sdr, tr["down2"])), 1e-5)
    # the same gradients from the oracle run in fp32: at these sizes the fp32 evaluation of the network sits percents away from
    # fp64 (ReLU-mask flips); the bar is "within 2e-3 of fp64, or no further from it than 1.5x the fp32 oracle"
    sd32 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x32 = x.clone().requires_grad_(True)
    g32 = torch.autograd.grad(R.generator_forward(sd32, x32, None, num_blocks=1).abs().mean(), [x32] + [sd32[k] for k in names])
    num = den = num32 = numo = 0.0
    for n, a, b, c in zip(["dx"] + names, grads, gr, g32):
        if n.endswith(".bias") and not n.startswith(("transformer_blocks", "style_encoder")) and not n.endswith(("qkv.bias", "proj.bias", "output.0.bias")):
            continue  # conv biases in front of an InstanceNorm: exactly-zero gradient (rounding noise in any implementation)
        num += float((a.cpu().double() - b).pow(2).sum())
        num32 += float((c.double() - b).pow(2).sum())
        numo += float((a.cpu().double() - c.double()).pow(2).sum())
        den += float(b.pow(2).sum())
    d32 = (num32 / den) ** 0.5
    print(f"  [parity] G[blocks=1, C={C_}] fp32 oracle vs fp64 oracle: {d32:.2e}")
    # One draw of a flip-driven error: tests/test_gpu_models.py::test_train_step_gradient_distance_distribution measures how two
    # correct fp32 evaluations of this network family scatter around fp64 ON THE SAME DRAW -- they flip different masks, the ratio of
    # their distances ranges over 0.2 ... 4 across 96 samples while the distributions agree (medians within 4 %).  The round-2 bar
    # "1.5 x the fp32 oracle's distance" on a single draw asserted more than that scatter allows; the bar is the scatter's edge.
    # The distance between the two fp32 evaluations follows from the two fp64 distances (triangle inequality) and is printed.
    print(f"  [parity] G[blocks=1, C={C_}] all gradients vs oracle (fp32): {(numo / den) ** 0.5:.2e}")
    report(f"G[blocks=1, C={C_}] all gradients vs oracle (fp64)", (num / den) ** 0.5, max(2e-3, 4.0 * d32))
    # round trip: a state_dict written by this module loads back strictly, inference callers' pattern (eval + no_grad)
    m2 = eg.EnhancedGenerator(channels=C_, num_transformer_blocks=1)
    m2.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    m2.to(DEV).eval()
    with torch.no_grad():
        assert torch.equal(m2(x.to(DEV)), m.eval()(x.to(DEV)))


def test_train_step_with_blocks_runs_and_is_deterministic():
    import enhanced_train

    def run():
        torch.manual_seed(5)
        m = enhanced_train.EnhancedCycleGAN(channels=8, device=torch.device(DEV))  # reference default: num_transformer_blocks=1
        assert len(m.G_AB.transformer_blocks) == 1 and not m.G_AB.transformer_blocks[0].is_identity
        g = torch.Generator().manual_seed(6)
        a = (torch.rand((2, 3, 64, 64), generator=g) * 2 - 1).to(DEV)
        b = (torch.rand((2, 3, 64, 64), generator=g) * 2 - 1).to(DEV)
        out = [m.train_step(a, b) for _ in range(2)]
        return out, m.g_optimizer.flat.clone()

    o1, f1 = run()
    o2, f2 = run()
    assert all(math.isfinite(v) for d in o1 for v in d.values())
    assert o1 == o2 and torch.equal(f1, f2)
    # the block's parameters received gradients and moved
    torch.manual_seed(5)
    fresh = enhanced_train.EnhancedCycleGAN(channels=8, device=torch.device(DEV))
    moved = (fresh.g_optimizer.flat - f1).abs().max()
    assert float(moved) > 0
