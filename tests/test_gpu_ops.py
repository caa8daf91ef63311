"""GPU parity of every C-ABI kernel against plain fp32 PyTorch on the CPU (the definition the reference's layers
dispatch to) and against the golden vectors of the reference's LocalAttention / MultiScaleBlock.

Tolerance: the north star asks for 1e-3 relative; the fp32 MFMA path is an exact fmaf chain, so these tests hold the
kernels to 2e-5 relative L2 (1e-4 for gradients that sum ~1e5 products) and print the measured error.
"""
import os

import numpy as np
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
    _lib.load()  # raises if the HIP library is missing -- there is no fallback to test instead


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def report(name, err, tol):
    print(f"  [parity] {name:60s} rel-L2 {err:.2e} (tol {tol:.0e})")
    assert err <= tol, f"{name}: {err:.3e} > {tol:.0e}"


# (name, N, H, W, Cin, Cout, k, stride, pad, dil, transposed, x_nchw, y_nchw, act)
CONV_CASES = [
    ("stem7x7 3->8 nchw-in", 2, 32, 48, 3, 8, 7, 1, 3, 1, 0, 1, 0, 0),
    ("stem7x7 3->16 nchw-in", 1, 64, 64, 3, 16, 7, 1, 3, 1, 0, 1, 0, 0),
    ("head7x7 8->3 nchw-out tanh", 2, 32, 48, 8, 3, 7, 1, 3, 1, 0, 0, 1, 3),
    ("head7x7 16->3 nchw-out tanh", 1, 64, 64, 16, 3, 7, 1, 3, 1, 0, 0, 1, 3),
    ("k4s2 8->16", 2, 32, 48, 8, 16, 4, 2, 1, 1, 0, 0, 0, 0),
    ("k4s2 16->32", 1, 64, 64, 16, 32, 4, 2, 1, 1, 0, 0, 0, 0),
    ("k4s2 32->64", 2, 16, 16, 32, 64, 4, 2, 1, 1, 0, 0, 0, 0),
    ("k4s2 64->128", 1, 8, 8, 64, 128, 4, 2, 1, 1, 0, 0, 0, 0),
    ("k4s2 3->8 nchw-in (D stem)", 2, 64, 64, 3, 8, 4, 2, 1, 1, 0, 1, 0, 0),
    ("k4s2 ragged 20x36", 1, 20, 36, 16, 24, 4, 2, 1, 1, 0, 0, 0, 0),
    ("convT 32->16", 2, 8, 12, 32, 16, 4, 2, 1, 1, 1, 0, 0, 0),
    ("convT 64->32", 1, 16, 16, 64, 32, 4, 2, 1, 1, 1, 0, 0, 0),
    ("convT 16->8", 2, 16, 24, 16, 8, 4, 2, 1, 1, 1, 0, 0, 0),
    ("convT 8->3 nchw-out tanh", 2, 16, 16, 8, 3, 4, 2, 1, 1, 1, 0, 1, 3),
    ("convT ragged 5x7", 1, 5, 7, 16, 12, 4, 2, 1, 1, 1, 0, 0, 0),
    ("k3 d1 16->4", 2, 16, 24, 16, 4, 3, 1, 1, 1, 0, 0, 0, 0),
    ("k3 d2 32->8", 1, 32, 32, 32, 8, 3, 1, 2, 2, 0, 0, 0, 0),
    ("k3 d4 64->16", 1, 16, 16, 64, 16, 3, 1, 4, 4, 0, 0, 0, 0),
    ("k3 d4 8->2", 2, 8, 12, 8, 2, 3, 1, 4, 4, 0, 0, 0, 0),
    ("k3 d1 64->64 (D structure head)", 2, 4, 4, 64, 64, 3, 1, 1, 1, 0, 0, 0, 0),
    ("k3 d1 128->128", 1, 16, 16, 128, 128, 3, 1, 1, 1, 0, 0, 0, 0),
    ("1x1 16->48", 2, 16, 24, 16, 48, 1, 1, 0, 1, 0, 0, 0, 0),
    ("1x1 64->192", 1, 8, 8, 64, 192, 1, 1, 0, 1, 0, 0, 0, 0),
    ("1x1 8->24", 2, 8, 12, 8, 24, 1, 1, 0, 1, 0, 0, 0, 0),
    ("1x1 128->384 (wgrad in 2x2 channel blocks)", 1, 12, 16, 128, 384, 1, 1, 0, 1, 0, 0, 0, 0),
    ("1x1 72->200 (uneven channel blocks)", 2, 8, 12, 72, 200, 1, 1, 0, 1, 0, 0, 0, 0),
    ("1x1 256->32", 1, 8, 8, 256, 32, 1, 1, 0, 1, 0, 0, 0, 0),
    ("k4s1p1 64->1 (D head)", 2, 4, 4, 64, 1, 4, 1, 1, 1, 0, 0, 0, 0),
    ("k4s1p1 128->1 16x16", 2, 16, 16, 128, 1, 4, 1, 1, 1, 0, 0, 0, 0),
    ("k3 d1 ragged 13x21 4->5", 1, 13, 21, 4, 5, 3, 1, 1, 1, 0, 0, 0, 0),
    # the persistent kernel of the 4x4 stride-2 family (conv_p32.hip): every channel pairing class, ragged tiles, several images
    ("p32 k4s2 16->16 ragged 20x36 N3", 3, 20, 36, 16, 16, 4, 2, 1, 1, 0, 0, 0, 0),
    ("p32 k4s2 64->64 24x40", 2, 24, 40, 64, 64, 4, 2, 1, 1, 0, 0, 0, 0),
    ("p32 k4s2 64->16 70x34", 1, 70, 34, 64, 16, 4, 2, 1, 1, 0, 0, 0, 0),
    ("p32 k4s2 32->32 128x128", 1, 128, 128, 32, 32, 4, 2, 1, 1, 0, 0, 0, 0),
    ("p32 k4s2 16->64 2x2", 2, 2, 2, 16, 64, 4, 2, 1, 1, 0, 0, 0, 0),
    ("p32 convT 16->64 ragged 5x7", 1, 5, 7, 16, 64, 4, 2, 1, 1, 1, 0, 0, 0),
    ("p32 convT 64->64 9x11", 2, 9, 11, 64, 64, 4, 2, 1, 1, 1, 0, 0, 0),
    ("p32 convT 16->16 17x33 N3", 3, 17, 33, 16, 16, 4, 2, 1, 1, 1, 0, 0, 0),
    ("p32 convT 32->32 64x64", 1, 64, 64, 32, 32, 4, 2, 1, 1, 1, 0, 0, 0),
    ("p32 convT 64->16 1x1", 2, 1, 1, 64, 16, 4, 2, 1, 1, 1, 0, 0, 0),
    ("p32 1x1 32->32 ragged 37x53 N3", 3, 37, 53, 32, 32, 1, 1, 0, 1, 0, 0, 0, 0),
    ("p32 1x1 64->64 24x40", 2, 24, 40, 64, 64, 1, 1, 0, 1, 0, 0, 0, 0),
    ("p32 1x1 32->64 128x128", 1, 128, 128, 32, 64, 1, 1, 0, 1, 0, 0, 0, 0),
    ("p32 1x1 64->16 5x7", 2, 5, 7, 64, 16, 1, 1, 0, 1, 0, 0, 0, 0),
    # the persistent 7x7 weight-gradient kernel (wgrad7_kernel): ragged tiles, several images, more tiles than workgroups
    ("w7 stem7x7 3->16 nchw-in ragged 37x53 N3", 3, 37, 53, 3, 16, 7, 1, 3, 1, 0, 1, 0, 0),
    ("w7 head7x7 16->3 nchw-out ragged 37x53 N3", 3, 37, 53, 16, 3, 7, 1, 3, 1, 0, 0, 1, 0),
    ("w7 stem7x7 3->16 nchw-in 256x256 N4", 4, 256, 256, 3, 16, 7, 1, 3, 1, 0, 1, 0, 0),
    ("w7 head7x7 16->3 nchw-out tanh 256x256 N4", 4, 256, 256, 16, 3, 7, 1, 3, 1, 0, 0, 1, 3),
    ("w7 stem7x7 1->16 nchw-in 16x16", 2, 16, 16, 1, 16, 7, 1, 3, 1, 0, 1, 0, 0),
    # the discriminator's image-side layer: its input gradient runs on the vector pipe (conv_img.hip); widths 8 ... 64, ragged, 1-3 channels
    ("img k4s2 3->16 nchw-in 64x64 N2", 2, 64, 64, 3, 16, 4, 2, 1, 1, 0, 1, 0, 0),
    ("img k4s2 3->64 nchw-in ragged 36x20 N3", 3, 36, 20, 3, 64, 4, 2, 1, 1, 0, 1, 0, 0),
    ("img k4s2 3->24 nchw-in 2x2", 2, 2, 2, 3, 24, 4, 2, 1, 1, 0, 1, 0, 0),
    ("img k4s2 1->16 nchw-in 18x30", 1, 18, 30, 1, 16, 4, 2, 1, 1, 0, 1, 0, 0),
    ("img k4s2 3->16 nchw-in 256x256 N4", 4, 256, 256, 3, 16, 4, 2, 1, 1, 0, 1, 0, 0),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_bwd(case):
    from mstg_hip import ops
    name, N, H, W, Cin, Cout, k, s, p, d, tr, x_nchw, y_nchw, act = case
    seed = sum(ord(c) for c in name) % 10000  # stable across processes
    x = rnd((N, Cin, H, W), seed)
    wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
    w = rnd(wshape, seed + 1, (2.0 / (Cin * k * k)) ** 0.5 * 1.7)
    b = rnd((Cout,), seed + 2, 0.3)
    # ---- CPU reference
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, br, stride=s, padding=p) if tr else F.conv2d(xr, wr, br, stride=s, padding=p, dilation=d)
    if act == 3:
        yr = torch.tanh(yr)
    gy = rnd(tuple(yr.shape), seed + 3)
    gxr, gwr, gbr = torch.autograd.grad((yr * gy).sum(), [xr, wr, br])
    # ---- HIP
    xg = (x if x_nchw else nhwc(x)).to(DEV).requires_grad_(True)
    wg, bg = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    yg = ops.conv2d(xg, wg, bg, k, s, p, d, transposed=bool(tr), x_nchw=bool(x_nchw), y_nchw=bool(y_nchw), act=act)
    gyg = (gy if y_nchw else nhwc(gy)).to(DEV)
    gxg, gwg, gbg = torch.autograd.grad((yg * gyg).sum(), [xg, wg, bg])
    y_cmp = yg if y_nchw else nchw(yg)
    gx_cmp = gxg if x_nchw else nchw(gxg)
    report(f"conv {name} y", rel_l2(y_cmp, yr), 2e-5)
    report(f"conv {name} dx", rel_l2(gx_cmp, gxr), 2e-5)
    report(f"conv {name} dw", rel_l2(gwg, gwr), 1e-4)
    report(f"conv {name} db", rel_l2(gbg, gbr), 1e-4)


STREAM_CASES = [
    ("stream k3 d4 16->4", 2, 64, 80, 16, 4, 3, 1, 4, 4, 0, 0, 0, 0),
    ("stream k3 d1 16->4 (dpack)", 2, 64, 80, 16, 4, 3, 1, 1, 1, 0, 0, 0, 0),
    ("stream k3 d2 32->8 (two chunks)", 2, 48, 64, 32, 8, 3, 1, 2, 2, 0, 0, 0, 0),
    ("stream head7x7 16->3 nchw-out tanh", 2, 64, 80, 16, 3, 7, 1, 3, 1, 0, 0, 1, 3),
    ("stream stem7x7 3->16 nchw-in", 2, 64, 80, 3, 16, 7, 1, 3, 1, 0, 1, 0, 0),
    ("stream 1x1 16->16", 2, 64, 80, 16, 16, 1, 1, 0, 1, 0, 0, 0, 0),
    ("stream convT 32->16", 2, 32, 40, 32, 16, 4, 2, 1, 1, 1, 0, 0, 0),
    ("stream k4s2 16->32 (dgrad by phases)", 2, 64, 80, 16, 32, 4, 2, 1, 1, 0, 0, 0, 0),
    ("stream ragged 37x53 16->20", 1, 37, 53, 16, 20, 3, 1, 1, 1, 0, 0, 0, 0),
]


@pytest.mark.parametrize("case", STREAM_CASES, ids=[c[0] for c in STREAM_CASES])
def test_conv_stream_kernel(case, monkeypatch):
    """The opt-in persistent streaming igemm kernel (MSTG_STREAM=1; "1f" also lifts its minimum tile count) on the same
    checks as the default kernels: interior fast path, border tiles, dpack exchange, channel chunks, parity classes."""
    from mstg_hip import _lib
    monkeypatch.setenv("MSTG_STREAM", "1f")
    monkeypatch.setenv("MSTG_P32", "0")  # the persistent 4x4 stride-2 kernel (conv_p32.hip) otherwise takes two of these shapes
    from mstg_hip import ops as _ops
    _ops.refresh_env()
    name, N, H, W, Cin, Cout, k, s, p, d, tr, x_nchw, y_nchw, act = case
    Ho, Wo = (2 * H, 2 * W) if tr else ((H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1)
    from mstg_hip import ops
    desc = ops.make_desc(N, H, W, Cin, Ho, Wo, Cout, k, s, p, d, tr, x_nchw, y_nchw, act=act)
    import ctypes
    names = {_lib.load().mstg_conv2d_kernel_name(ctypes.byref(desc), ps).decode() for ps in (0, 1)}
    assert any("stream" in n for n in names), names
    test_conv_fwd_bwd(case)


@pytest.mark.parametrize("shape", [(16, 3, 4, 4), (128, 64, 4, 4), (128, 128, 3, 3), (1, 128, 4, 4), (24, 8, 1, 1)])
def test_spectral_norm_fused_vs_torch(shape):
    """One-launch spectral norm against torch.nn.utils.spectral_norm on the CPU: normalised weight, sigma path gradient,
    and the evolution of weight_u / weight_v over three training-mode forwards and one eval-mode forward."""
    import torch.nn as nn
    from mstg_hip import ops
    torch.manual_seed(sum(shape))
    ref = nn.utils.spectral_norm(nn.Conv2d(shape[1], shape[0], (shape[2], shape[3])))
    w0, u0, v0 = ref.weight_orig.detach().clone(), ref.weight_u.clone(), ref.weight_v.clone()
    w = w0.to(DEV).requires_grad_(True)
    u, v = u0.to(DEV), v0.to(DEV)
    x = rnd((2, shape[1], shape[2] + 3, shape[3] + 3), 5)
    for it in range(4):
        training = it < 3
        ref.train(training)
        ref.zero_grad()
        y = ref(x)
        gy = rnd(tuple(y.shape), 6 + it)
        (y * gy).sum().backward()
        wn_ref, dw_ref = ref.weight.detach(), ref.weight_orig.grad
        wn = ops.SpectralNormFn.apply(w, u, v, 1e-12, training)
        yr = F.conv2d(x.to(DEV), wn, ref.bias.detach().to(DEV))
        (dw,) = torch.autograd.grad((yr * gy.to(DEV)).sum(), [w])
        report(f"spectral_norm {shape} it{it} weight", rel_l2(wn, wn_ref), 1e-5)
        report(f"spectral_norm {shape} it{it} u", rel_l2(u, ref.weight_u), 1e-5)
        report(f"spectral_norm {shape} it{it} v", rel_l2(v, ref.weight_v), 1e-5)
        report(f"spectral_norm {shape} it{it} dweight_orig", rel_l2(dw, dw_ref), 1e-4)


def test_spectral_norm_group_vs_single():
    """The grouped call (one discriminator forward's seven weights in three launches) against the one-weight call on the same
    weights and vectors: normalised weights, the in-place evolution of u / v over three training-mode calls and one eval-mode call,
    and the gradients -- with one output left out of the loss (the discriminator update never uses the structure head), whose
    weight must then get no gradient, and with gradients accumulated straight into .grad slots (ops.direct_param_grads)."""
    from mstg_hip import ops
    shapes = [(16, 3, 4, 4), (32, 16, 4, 4), (64, 32, 4, 4), (128, 64, 4, 4), (1, 128, 4, 4), (128, 128, 3, 3), (1, 128, 4, 4)]
    n = len(shapes)
    w0 = [rnd(sh, 300 + j, 0.2) for j, sh in enumerate(shapes)]
    u0 = [F.normalize(rnd((sh[0],), 320 + j), dim=0, eps=1e-12) for j, sh in enumerate(shapes)]
    v0 = [F.normalize(rnd((sh[1] * sh[2] * sh[3],), 340 + j), dim=0, eps=1e-12) for j, sh in enumerate(shapes)]
    wa = [t.to(DEV).requires_grad_(True) for t in w0]
    wb = [torch.nn.Parameter(t.to(DEV)) for t in w0]
    ua, va = [t.to(DEV) for t in u0], [t.to(DEV) for t in v0]
    ub, vb = [t.to(DEV) for t in u0], [t.to(DEV) for t in v0]
    skip = n - 2  # this output takes no part in the loss
    for it in range(4):
        training = it < 3
        gy = [rnd(sh, 360 + 10 * it + j).to(DEV) for j, sh in enumerate(shapes)]
        single = [ops.SpectralNormFn.apply(wa[j], ua[j], va[j], 1e-12, training) for j in range(n)]
        group = ops.SpectralNormGroupFn.apply(1e-12, training, n, *wb, *ub, *vb)
        ga = torch.autograd.grad(sum((single[j] * gy[j]).sum() for j in range(n) if j != skip), wa, allow_unused=True)
        gb = torch.autograd.grad(sum((group[j] * gy[j]).sum() for j in range(n) if j != skip), wb, allow_unused=True)
        assert ga[skip] is None and gb[skip] is None
        for j in range(n):
            report(f"spectral group it{it} weight {shapes[j]}", rel_l2(group[j], single[j]), 2e-6)
            report(f"spectral group it{it} u {shapes[j]}", rel_l2(ub[j], ua[j]), 2e-6)
            report(f"spectral group it{it} v {shapes[j]}", rel_l2(vb[j], va[j]), 2e-6)
            if j != skip:
                report(f"spectral group it{it} dweight_orig {shapes[j]}", rel_l2(gb[j], ga[j]), 1e-5)
    # gradients straight into .grad (accumulating): what the train step does
    seeds = [rnd(sh, 500 + j).to(DEV) for j, sh in enumerate(shapes)]
    for w_, s0 in zip(wb, seeds):
        w_.grad = s0.clone()
    gy = [rnd(sh, 600 + j).to(DEV) for j, sh in enumerate(shapes)]
    ref = torch.autograd.grad(sum((o * g_).sum() for o, g_ in zip(ops.SpectralNormGroupFn.apply(1e-12, False, n, *wb, *ub, *vb), gy)), wb)
    with ops.direct_param_grads():
        sum((o * g_).sum() for o, g_ in zip(ops.SpectralNormGroupFn.apply(1e-12, False, n, *wb, *ub, *vb), gy)).backward()
    for w_, s0, r in zip(wb, seeds, ref):
        assert torch.equal(w_.grad, s0 + r)


@pytest.mark.parametrize("N,H,W,ch", [(2, 16, 24, 16), (2, 21, 37, 16), (1, 32, 32, 32), (2, 9, 20, 64), (3, 40, 56, 16), (1, 16, 16, 8),
                                      (1, 8, 8, 128)])
def test_conv_channel_slices_and_accumulate(N, H, W, ch):
    """The multi-scale block's four branches: output into channel slices of one buffer, input gradients accumulated, and
    (16 / 32 / 64 channels) the eight parameter gradients from the fused one-pass kernel; other widths per branch."""
    from mstg_hip import ops
    x = rnd((N, ch, H, W), 5)
    ws = [rnd((ch // 4, ch, k, k), 6 + j, 0.2) for j, k in enumerate((1, 3, 3, 3))]
    bs = [rnd((ch // 4,), 10 + j, 0.2) for j in range(4)]
    xr = x.clone().requires_grad_(True)
    wr = [t.clone().requires_grad_(True) for t in ws]
    br = [t.clone().requires_grad_(True) for t in bs]
    outs = [F.conv2d(xr, wr[0], br[0])] + [F.conv2d(xr, wr[j], br[j], padding=d, dilation=d) for j, d in ((1, 1), (2, 2), (3, 4))]
    yr = torch.cat(outs, 1)
    gy = rnd(tuple(yr.shape), 20)
    gres = rnd(tuple(x.shape), 21)  # gradient arriving over the block's residual connection (second output = x itself)
    gr = torch.autograd.grad((yr * gy).sum() + (xr * gres).sum(), [xr] + wr + br)
    xg = nhwc(x).to(DEV).requires_grad_(True)
    wg = [t.to(DEV).requires_grad_(True) for t in ws]
    bg = [t.to(DEV).requires_grad_(True) for t in bs]
    args = []
    for a, b in zip(wg, bg):
        args += [a, b]
    yg, xa = ops.MSBranchesFn.apply(xg, *args)
    assert xa.data_ptr() == xg.data_ptr()
    gg = torch.autograd.grad((yg * nhwc(gy).to(DEV)).sum() + (xa * nhwc(gres).to(DEV)).sum(), [xg] + wg + bg)
    report("msbranches y", rel_l2(nchw(yg), yr), 2e-5)
    report("msbranches dx (4 dgrads + residual gradient)", rel_l2(nchw(gg[0]), gr[0]), 2e-5)
    for j in range(4):
        report(f"msbranches dw{j + 1}", rel_l2(gg[1 + j], gr[1 + j]), 1e-4)
        report(f"msbranches db{j + 1}", rel_l2(gg[5 + j], gr[5 + j]), 1e-4)


@pytest.mark.parametrize("env", ["MSTG_MS_UNFUSED=1", "MSTG_MS_WGRAD_PACKED=0", "MSTG_MS_FWD4=0", "MSTG_MS_FWD4=1", "MSTG_MS_FWD4=2", "MSTG_WGLOB=0",
                                 "MSTG_WGRAD_1X1=0", "MSTG_PF=2", "MSTG_NO_DPACK=1", "MSTG_WGRAD_PLAIN=1", "MSTG_ATTN_BLK64=0", "MSTG_ATTN_BLK4=0",
                                 "MSTG_P32=0", "MSTG_P32_TH=4", "MSTG_P32_TH=8", "MSTG_P32_TH=16", "MSTG_P32_WLDS=0", "MSTG_P32_WLDS=1",
                                 "MSTG_ATTN_REG=0", "MSTG_ATTN_BIG32=1", "MSTG_CONV_IMG=0"])
def test_kernel_selection_switches_keep_parity(env, monkeypatch):
    """Every runtime switch of INTEGRATION.md section 3 selects another kernel for the same arithmetic: the fallbacks stay correct."""
    k, v = env.split("=")
    monkeypatch.setenv(k, v)
    from mstg_hip import ops as _ops
    _ops.refresh_env()
    test_conv_channel_slices_and_accumulate(2, 21, 37, 16)
    test_conv_channel_slices_and_accumulate(1, 32, 32, 32)
    for case in CONV_CASES:
        if case[0] in ("head7x7 16->3 nchw-out tanh", "k4s2 16->32", "1x1 16->48", "k3 d1 16->4") or (k.startswith("MSTG_P32") and (
                case[0].startswith(("p32", "convT", "k4s2", "w7", "stem7x7", "head7x7")))):
            test_conv_fwd_bwd(case)
    if k == "MSTG_CONV_IMG":  # the image-side 4x4 stride-2 layer back on the MFMA kernels
        for case in CONV_CASES:
            if case[0].startswith("img") or "D stem" in case[0]:
                test_conv_fwd_bwd(case)
    if k == "MSTG_ATTN_BLK64":
        test_window_attention_core(2, 8, 8, 64)
        test_window_attention_core(1, 8, 4, 48)
    if k == "MSTG_ATTN_BLK4":
        for shape in ((2, 8, 8, 128), (1, 4, 8, 256), (1, 8, 4, 96), (1, 4, 4, 200)):
            test_window_attention_core(*shape)
    if k == "MSTG_ATTN_REG":  # the LDS-tile fused attention kernels of rounds 1-2 behind the register-resident ones
        for shape in ((2, 8, 12, 16), (1, 16, 16, 32), (1, 64, 64, 16)):
            test_local_attention_fused_vs_oracle(*shape)
        import test_gpu_normfuse
        test_gpu_normfuse.test_norm_attention_fused_vs_chain_and_torch(16, 3, 16, 24)
        test_gpu_normfuse.test_norm_attention_fused_vs_chain_and_torch(32, 2, 24, 16)
    if k == "MSTG_ATTN_BIG32":  # C = 32 through the C = 64 implementation (filters in LDS, row-block transposes)
        for shape in ((1, 16, 16, 32), (1, 32, 64, 32), (4, 96, 64, 32)):
            test_local_attention_fused_vs_oracle(*shape)
        import test_gpu_normfuse
        test_gpu_normfuse.test_norm_attention_fused_vs_chain_and_torch(32, 2, 24, 16)
        test_gpu_normfuse.test_norm_attention_fused_vs_chain_and_torch(32, 3, 96, 64)


NORM_CASES = [(2, 16, 24, 8, 1), (1, 64, 64, 16, 1), (3, 7, 9, 32, 2), (2, 4, 4, 64, 2), (1, 128, 128, 16, 1), (2, 2, 2, 64, 2),
              (2, 16, 16, 4, 0), (1, 32, 32, 128, 1),
              # small batches of large maps: more than 32 partial rows per image, added once by the grouped statistics / sums kernels
              (1, 96, 96, 24, 1), (2, 128, 96, 64, 2), (1, 160, 128, 256, 1)]


@pytest.mark.parametrize("N,H,W,C,act", NORM_CASES)
@pytest.mark.parametrize("residual", [False, True])
def test_instnorm_act(N, H, W, C, act, residual):
    from mstg_hip import ops
    x = rnd((N, C, H, W), 31) * 3 + 5.0  # mean >> 0 exercises the pivoted variance
    res = rnd((N, C, H, W), 32) if residual else None
    xr = x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if residual else None
    z = F.instance_norm(xr, eps=1e-5)
    yr = {0: z, 1: F.relu(z), 2: F.leaky_relu(z, 0.2)}[act]
    if residual:
        yr = yr + rr
    gy = rnd(tuple(yr.shape), 33)
    gr = torch.autograd.grad((yr * gy).sum(), [xr] + ([rr] if residual else []))
    xg = nhwc(x).to(DEV).requires_grad_(True)
    rg = nhwc(res).to(DEV).requires_grad_(True) if residual else None
    yg = ops.instnorm_act(xg, act, rg)
    gg = torch.autograd.grad((yg * nhwc(gy).to(DEV)).sum(), [xg] + ([rg] if residual else []))
    report(f"instnorm act{act} res{int(residual)} {N}x{H}x{W}x{C} y", rel_l2(nchw(yg), yr), 2e-5)
    report(f"instnorm act{act} res{int(residual)} {N}x{H}x{W}x{C} dx", rel_l2(nchw(gg[0]), gr[0]), 1e-4)
    if residual:
        report("instnorm dres", rel_l2(nchw(gg[1]), gr[1]), 1e-6)


@pytest.mark.parametrize("N,H,W,C,act", [(2, 16, 16, 16, 2), (4, 8, 8, 64, 1), (2, 32, 32, 8, 1)])
def test_batchnorm_act(N, H, W, C, act):
    from mstg_hip import ops
    x = rnd((N, C, H, W), 41) * 2 + 1.0
    gamma, beta = 1 + 0.3 * rnd((C,), 42), 0.3 * rnd((C,), 43)
    rm, rv = 0.1 * rnd((C,), 44), 1 + 0.3 * rnd((C,), 45)
    xr, gr_, br_ = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rmr, rvr = rm.clone(), rv.clone()
    z = F.batch_norm(xr, rmr, rvr, gr_, br_, training=True, momentum=0.1, eps=1e-5)
    yr = F.relu(z) if act == 1 else F.leaky_relu(z, 0.2)
    gy = rnd(tuple(yr.shape), 46)
    grads = torch.autograd.grad((yr * gy).sum(), [xr, gr_, br_])
    xg = nhwc(x).to(DEV).requires_grad_(True)
    gg_, bg_ = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    rmg, rvg = rm.to(DEV), rv.to(DEV)
    yg = ops.BatchNormActFn.apply(xg, gg_, bg_, rmg, rvg, act, True)
    gg = torch.autograd.grad((yg * nhwc(gy).to(DEV)).sum(), [xg, gg_, bg_])
    report(f"batchnorm train {N}x{H}x{W}x{C} y", rel_l2(nchw(yg), yr), 2e-5)
    report("batchnorm dx", rel_l2(nchw(gg[0]), grads[0]), 1e-4)
    report("batchnorm dgamma", rel_l2(gg[1], grads[1]), 1e-4)
    report("batchnorm dbeta", rel_l2(gg[2], grads[2]), 1e-4)
    report("batchnorm running_mean", rel_l2(rmg, rmr), 1e-5)
    report("batchnorm running_var", rel_l2(rvg, rvr), 1e-5)
    with torch.no_grad():
        ye = ops.BatchNormActFn.apply(xg, gg_, bg_, rmg, rvg, act, False)
        zr = F.batch_norm(x, rmr, rvr, gamma, beta, training=False, eps=1e-5)
        yer = F.relu(zr) if act == 1 else F.leaky_relu(zr, 0.2)
    report("batchnorm eval y", rel_l2(nchw(ye), yer), 2e-5)


def _attn_core_ref(qkv, C):
    """o from qkv (NCHW, 3C channels) by the oracle's own formulation (window 4)."""
    B, _, H, W = qkv.shape
    q, k, v = qkv.chunk(3, dim=1)
    q = q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)
    k = k / k.norm(dim=1, keepdim=True).clamp_min(1e-12)

    def win(t):
        return t.reshape(B, C, H // 4, 4, W // 4, 4).permute(0, 2, 4, 1, 3, 5).reshape(B, H // 4, W // 4, C, 16)

    attn = torch.einsum("bhwcp,bhwdp->bhwcd", win(q), win(k)).softmax(dim=-1)
    ow = torch.einsum("bhwcd,bhwdp->bhwcp", attn, win(v))
    return ow.reshape(B, H // 4, W // 4, C, 4, 4).permute(0, 3, 1, 4, 2, 5).reshape(B, C, H, W)


@pytest.mark.parametrize("N,H,W,C", [(2, 8, 12, 8), (1, 4, 8, 16), (2, 16, 16, 16), (1, 16, 8, 32), (2, 8, 8, 64), (1, 64, 64, 16),
                                     (1, 4, 4, 4), (1, 8, 8, 24), (1, 8, 4, 48), (2, 8, 8, 128), (1, 4, 8, 256), (1, 8, 4, 96), (1, 4, 4, 200), (2, 96, 128, 128),
                                     (1, 64, 96, 256)])
def test_window_attention_core(N, H, W, C):
    from mstg_hip import ops
    qkv = rnd((N, 3 * C, H, W), 51 + C, 2.0)
    qr = qkv.clone().requires_grad_(True)
    orf = _attn_core_ref(qr, C)
    go = rnd(tuple(orf.shape), 52)
    (gqr,) = torch.autograd.grad((orf * go).sum(), [qr])
    qg = nhwc(qkv).to(DEV).requires_grad_(True)
    og = ops.WindowAttnCoreFn.apply(qg)
    (gqg,) = torch.autograd.grad((og * nhwc(go).to(DEV)).sum(), [qg])
    report(f"window-attn core {N}x{H}x{W} C={C} o", rel_l2(nchw(og), orf), 2e-5)
    report(f"window-attn core {N}x{H}x{W} C={C} dqkv", rel_l2(nchw(gqg), gqr), 1e-4)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_local_attention_golden(gold_dir, tag):
    """LocalAttention module (qkv conv -> core -> proj conv) against the reference's own LocalAttention outputs."""
    import enhanced_generator as eg
    from oracle import restatement as R
    g = np.load(os.path.join(gold_dir, "ops.npz"))
    ch, shape, seed = int(g[f"attn_{tag}_ch"]), tuple(g[f"attn_{tag}_shape"]), int(g[f"attn_{tag}_seed"])
    spec = [("qkv.weight", (3 * ch, ch, 1, 1)), ("qkv.bias", (3 * ch,)), ("proj.weight", (ch, ch, 1, 1)), ("proj.bias", (ch,))]
    m = eg.LocalAttention(ch, window_size=4)
    m.load_state_dict(R.make_state_dict(spec, seed))
    m.to(DEV)
    x = R.make_input(shape, seed + 100).to(DEV).requires_grad_(True)
    y = m(x)  # NCHW contract
    gy = R.make_input(tuple(y.shape), seed + 200).to(DEV)
    grads = torch.autograd.grad((y * gy).sum(), [x] + list(m.parameters()))
    report(f"LocalAttention[{tag}] y vs reference", rel_l2(y, torch.from_numpy(g[f"attn_{tag}_y"])), 2e-5)
    report(f"LocalAttention[{tag}] dx vs reference", rel_l2(grads[0], torch.from_numpy(g[f"attn_{tag}_dx"])), 1e-4)
    for (k, _), gr in zip(m.named_parameters(), grads[1:]):
        report(f"LocalAttention[{tag}] d{k} vs reference", rel_l2(gr, torch.from_numpy(g[f"attn_{tag}_d_{k}"])), 1e-4)


@pytest.mark.parametrize("ch,ws,shape", [(16, 8, (2, 16, 16, 24)), (32, 8, (1, 32, 8, 16)), (8, 2, (2, 8, 6, 10)), (64, 8, (1, 64, 16, 8)),
                                         (16, 16, (1, 16, 16, 32))])
def test_local_attention_other_window_sizes(ch, ws, shape):
    """LocalAttention with a window size other than 4 -- the constructor default is 8 (enhanced_generator.py:7), no caller uses it --
    through the general-window core (csrc/attention_ws.hip) against the oracle's local_attention(ws): forward and every gradient;
    H, W not multiples of the window raise like the reference's .view does; what does not fit a CU's LDS raises."""
    import enhanced_generator as eg
    from oracle import restatement as R
    spec = [("qkv.weight", (3 * ch, ch, 1, 1)), ("qkv.bias", (3 * ch,)), ("proj.weight", (ch, ch, 1, 1)), ("proj.bias", (ch,))]
    sd = R.make_state_dict(spec, 300 + ch + ws)
    m = eg.LocalAttention(ch) if ws == 8 else eg.LocalAttention(ch, window_size=ws)
    assert m.window_size == ws
    m.load_state_dict(sd)
    m.to(DEV)
    x = R.make_input(shape, 301 + ch)
    xg = x.to(DEV).requires_grad_(True)
    y = m(xg)
    gy = R.make_input(tuple(y.shape), 302)
    grads = torch.autograd.grad((y * gy.to(DEV)).sum(), [xg] + list(m.parameters()))
    sdr = {"p." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = R.local_attention(xr, sdr, "p", ws)
    gr = torch.autograd.grad((yr * gy).sum(), [xr] + [sdr["p." + k] for k, _ in m.named_parameters()])
    report(f"LocalAttention ws={ws} C={ch} y", rel_l2(y, yr), 2e-5)
    for name, a_, b_ in zip(["dx"] + [k for k, _ in m.named_parameters()], grads, gr):
        report(f"LocalAttention ws={ws} C={ch} d {name}", rel_l2(a_, b_), 1e-4)
    with pytest.raises(RuntimeError):
        m(torch.zeros((1, ch, ws + 1, ws), device=DEV))
    if ws == 16:
        with pytest.raises(RuntimeError, match="does not fit"):
            eg.LocalAttention(256, window_size=16).to(DEV)(torch.zeros((1, 256, 16, 16), device=DEV))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_multi_scale_block_golden(gold_dir, tag):
    import enhanced_generator as eg
    from oracle import restatement as R
    g = np.load(os.path.join(gold_dir, "ops.npz"))
    ch, shape, seed = int(g[f"msb_{tag}_ch"]), tuple(g[f"msb_{tag}_shape"]), int(g[f"msb_{tag}_seed"])
    spec = [("branch1.0.weight", (ch // 4, ch, 1, 1)), ("branch1.0.bias", (ch // 4,))]
    for b in (2, 3, 4):
        spec += [(f"branch{b}.0.weight", (ch // 4, ch, 3, 3)), (f"branch{b}.0.bias", (ch // 4,))]
    spec += [("fusion.0.weight", (ch, ch, 1, 1)), ("fusion.0.bias", (ch,))]
    m = eg.MultiScaleBlock(ch)
    m.load_state_dict(R.make_state_dict(spec, seed))
    m.to(DEV)
    x = R.make_input(shape, seed + 100).to(DEV).requires_grad_(True)
    y = m(x)
    gy = R.make_input(tuple(y.shape), seed + 200).to(DEV)
    grads = torch.autograd.grad((y * gy).sum(), [x] + list(m.parameters()))
    report(f"MultiScaleBlock[{tag}] y vs reference", rel_l2(y, torch.from_numpy(g[f"msb_{tag}_y"])), 2e-5)
    report(f"MultiScaleBlock[{tag}] dx vs reference", rel_l2(grads[0], torch.from_numpy(g[f"msb_{tag}_dx"])), 1e-4)
    for (k, _), gr in zip(m.named_parameters(), grads[1:]):
        if k.endswith("weight"):  # biases in front of InstanceNorm have an exactly-zero gradient (pure rounding noise)
            report(f"MultiScaleBlock[{tag}] d{k} vs reference", rel_l2(gr, torch.from_numpy(g[f"msb_{tag}_d_{k}"])), 1e-4)
        else:
            assert float(gr.abs().max()) < 1e-4


def test_losses_act_pool():
    from mstg_hip import ops
    a, b = rnd((4, 3, 32, 32), 61), rnd((4, 3, 32, 32), 62)
    b[0, 0, 0, :4] = a[0, 0, 0, :4]  # exact ties: sign(0) = 0 like torch
    for kind, fn in ((0, F.l1_loss), (1, F.mse_loss)):
        ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        lr_ = fn(ar, br) * 3.5
        gar, gbr = torch.autograd.grad(lr_, [ar, br])
        ag, bg = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        lg = (ops.l1_loss(ag, bg) if kind == 0 else ops.mse_loss(ag, bg)) * 3.5
        gag, gbg = torch.autograd.grad(lg, [ag, bg])
        report(f"loss kind{kind} value", abs(float(lg) - float(lr_)) / abs(float(lr_)), 1e-6)
        report(f"loss kind{kind} da", rel_l2(gag, gar), 1e-6)
        report(f"loss kind{kind} db", rel_l2(gbg, gbr), 1e-6)
    s = rnd((5,), 63)
    sr = s.clone().requires_grad_(True)
    sg = s.to(DEV).requires_grad_(True)
    for target in (0.0, 1.0):
        lr_ = F.mse_loss(sr, torch.full_like(sr, target))
        lg = ops.mse_to_const(sg, target)
        report(f"mse to const {target}", abs(float(lg) - float(lr_)) / abs(float(lr_)), 1e-6)
        report("mse to const grad", rel_l2(torch.autograd.grad(lg, sg)[0], torch.autograd.grad(lr_, sr)[0]), 1e-6)
    lg0 = ops.mse_to_const(torch.tensor(0.3, device=DEV), 1.0)  # 0-dim score (batch of one)
    assert abs(float(lg0) - 0.49) < 1e-6
    x = rnd((3, 17, 5, 7), 64)
    for act, fn in ((1, F.relu), (2, lambda t: F.leaky_relu(t, 0.2)), (3, torch.tanh)):
        xr = x.clone().requires_grad_(True)
        yr = fn(xr)
        gy = rnd(tuple(yr.shape), 65)
        (gxr,) = torch.autograd.grad((yr * gy).sum(), [xr])
        xg = x.to(DEV).requires_grad_(True)
        yg = ops.activation(xg, act)
        (gxg,) = torch.autograd.grad((yg * gy.to(DEV)).sum(), [xg])
        report(f"activation {act} y", rel_l2(yg, yr), 1e-6)
        report(f"activation {act} dx", rel_l2(gxg, gxr), 1e-6)
    p = rnd((3, 15, 15, 1), 66)
    pr = p.clone().requires_grad_(True)
    pg = p.to(DEV).requires_grad_(True)
    mr, mg = pr.mean(dim=(1, 2)), ops.spatial_mean(pg)
    report("spatial mean", rel_l2(mg, mr), 1e-6)
    gm = rnd((3, 1), 67)
    report("spatial mean grad", rel_l2(torch.autograd.grad((mg * gm.to(DEV)).sum(), pg)[0], torch.autograd.grad((mr * gm).sum(), pr)[0]), 1e-6)
    p2 = rnd((2, 8, 8, 64), 68)
    report("spatial mean C=64", rel_l2(ops.spatial_mean(p2.to(DEV)), p2.mean(dim=(1, 2))), 1e-6)


def test_flat_adam_matches_torch():
    from mstg_hip.optim import FlatAdam
    torch.manual_seed(0)
    shapes = [(16, 3, 7, 7), (16,), (5, 3), (1,), (33,)]
    ps_ref = [torch.nn.Parameter(rnd(s, 70 + i)) for i, s in enumerate(shapes)]
    ps_hip = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ps_ref]
    opt_ref = torch.optim.Adam(ps_ref, lr=5e-5, betas=(0.5, 0.999))
    opt_hip = FlatAdam(ps_hip, lr=5e-5, betas=(0.5, 0.999))
    for step in range(4):
        opt_ref.zero_grad(set_to_none=True)
        opt_hip.zero_grad()
        for i, (pr, ph) in enumerate(zip(ps_ref, ps_hip)):
            if i == 4:
                continue  # never receives a gradient: torch skips it, the flat step must leave it untouched
            g = rnd(tuple(pr.shape), 100 + 10 * step + i)
            pr.grad = g.clone()
            ph.grad.copy_(g)
        opt_ref.step()
        opt_hip.step()
    for i, (pr, ph) in enumerate(zip(ps_ref, ps_hip)):
        report(f"flat adam param {i}", rel_l2(ph, pr), 1e-6)
    assert torch.equal(ps_hip[4].detach().cpu(), rnd(shapes[4], 74))


def test_weighted_loss_sum_and_direct_attention_grads():
    """ops.weighted_sum (the loss combinations of enhanced_train.py:72-81, 95-131 in one launch each way) against torch, and the fused
    attention's parameter gradients accumulated straight into .grad (ops.direct_param_grads) against the returned-gradient path."""
    from mstg_hip import ops
    vals = [0.3, -1.25, 2.0, 0.5, 7.0]
    w = [1.0, 1.0, 10.0, 2.0, 0.5]
    t = [torch.tensor(v, device=DEV, requires_grad=True) for v in vals]
    total, parts = ops.weighted_sum(t, w, report=((1.0, 1.0, 0, 0, 0), (0, 0, 10.0, 0, 0)))
    tr = [torch.tensor(v, requires_grad=True) for v in vals]
    ref = sum(wi * ti for wi, ti in zip(w, tr))
    assert abs(float(total) - float(ref)) <= 1e-6 * abs(float(ref)) and parts.shape == (2,)
    assert abs(float(parts[0]) - (vals[0] + vals[1])) <= 1e-6 and abs(float(parts[1]) - 10.0 * vals[2]) <= 1e-6
    (total * 3.0).backward()
    (ref * 3.0).backward()
    for a_, b_ in zip(t, tr):
        assert abs(float(a_.grad) - float(b_.grad)) <= 1e-6
    assert not parts.requires_grad
    # fused attention: gradients added into existing .grad slots == returned gradients added by hand
    Cn, shape = 16, (2, 8, 12, 16)
    x = rnd(shape, 1).to(DEV)
    ps = [torch.nn.Parameter(v.to(DEV)) for v in (rnd((3 * Cn, Cn, 1, 1), 2, 0.25), rnd((3 * Cn,), 3, 0.1), rnd((Cn, Cn, 1, 1), 4, 0.25), rnd((Cn,), 5, 0.1))]
    gy = rnd(shape, 6).to(DEV)
    xg = x.clone().requires_grad_(True)
    g_ret = torch.autograd.grad((ops.LocalAttentionFusedFn.apply(xg, *ps) * gy).sum(), [xg] + ps)
    seeds = [rnd(tuple(p_.shape), 10 + i).to(DEV) for i, p_ in enumerate(ps)]
    for p_, s0 in zip(ps, seeds):
        p_.grad = s0.clone()
    xg2 = x.clone().requires_grad_(True)
    with ops.direct_param_grads():
        (ops.LocalAttentionFusedFn.apply(xg2, *ps) * gy).sum().backward()
    assert torch.equal(xg2.grad, g_ret[0])
    for p_, s0, gr in zip(ps, seeds, g_ret[1:]):
        assert torch.equal(p_.grad, s0 + gr)


@pytest.mark.parametrize("N,H,W,C", [(2, 8, 12, 16), (1, 16, 16, 32), (3, 4, 4, 16), (1, 32, 64, 32), (1, 64, 64, 16), (2, 8, 12, 64),
                                     (1, 4, 4, 64), (5, 64, 128, 64)])
def test_local_attention_fused_vs_oracle(N, H, W, C):
    """The one-kernel LocalAttention (C = 16 / 32 / 64) against the oracle's local_attention on the CPU; the last case gives every
    wave of the C = 64 kernels two or three windows."""
    from mstg_hip import ops
    from oracle import restatement as R
    spec = [("qkv.weight", (3 * C, C, 1, 1)), ("qkv.bias", (3 * C,)), ("proj.weight", (C, C, 1, 1)), ("proj.bias", (C,))]
    sd = R.make_state_dict(spec, 900 + C)
    x = rnd((N, C, H, W), 901 + C, 2.0)
    sdr = {"p." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = R.local_attention(xr, sdr, "p", 4)
    gy = rnd(tuple(yr.shape), 902)
    gr = torch.autograd.grad((yr * gy).sum(), [xr] + list(sdr.values()))
    xg = nhwc(x).to(DEV).requires_grad_(True)
    pg = [sd[k].to(DEV).requires_grad_(True) for k in ("qkv.weight", "qkv.bias", "proj.weight", "proj.bias")]
    assert ops.fused_attention_supported(C)
    yg = ops.LocalAttentionFusedFn.apply(xg, *pg)
    gg = torch.autograd.grad((yg * nhwc(gy).to(DEV)).sum(), [xg] + pg)
    report(f"fused attention {N}x{H}x{W} C={C} y", rel_l2(nchw(yg), yr), 2e-5)
    report(f"fused attention {N}x{H}x{W} C={C} dx", rel_l2(nchw(gg[0]), gr[0]), 1e-4)
    for name, a, b in zip(("dWqkv", "dbqkv", "dWproj", "dbproj"), gg[1:], gr[1:]):
        report(f"fused attention {N}x{H}x{W} C={C} {name}", rel_l2(a, b), 1e-4)


def test_maxpool_and_gram_vs_torch():
    """Build-defined style-loss pieces (parity unpinned: no reference implementation): max-pool values AND arg-max indices are
    bit-exact against torch, Gram forward/backward against a CPU einsum."""
    from mstg_hip import ops
    x = rnd((2, 16, 12, 20), 71)
    x[0, 0, 0, 0] = x[0, 0, 0, 1] = x[0, 0, 1, 0] = x[0, 0, 1, 1] = 0.25  # a 4-way tie: the first slot must win
    xr = x.clone().requires_grad_(True)
    yr, ir = F.max_pool2d(xr, 2, return_indices=True)
    gy = rnd(tuple(yr.shape), 72)
    (gxr,) = torch.autograd.grad((yr * gy).sum(), [xr])
    xg = nhwc(x).to(DEV).requires_grad_(True)
    yg, ig = ops.maxpool2x2(xg, return_indices=True)
    (gxg,) = torch.autograd.grad((yg * nhwc(gy).to(DEV)).sum(), [xg])
    assert torch.equal(nchw(yg).cpu(), yr.detach()), "max-pool values must be bit-exact"
    oy, ox = torch.meshgrid(torch.arange(6), torch.arange(10), indexing="ij")
    slot = nchw(ig).cpu().long()
    flat = (2 * oy + slot // 2) * 20 + (2 * ox + slot % 2)
    assert torch.equal(flat, ir), "arg-max indices must match torch exactly"
    assert torch.equal(nchw(gxg).cpu(), gxr), "max-pool backward must be bit-exact"
    for N, H, W, C in ((2, 8, 8, 16), (1, 32, 32, 64), (2, 16, 8, 128), (1, 128, 96, 32)):
        f = rnd((N, C, H, W), 73 + C)
        fr = f.clone().requires_grad_(True)
        m = fr.reshape(N, C, H * W)
        gr = torch.bmm(m, m.transpose(1, 2)) / (C * H * W)
        w = rnd(tuple(gr.shape), 74)  # deliberately NOT symmetric: exercises dG + dG^T
        (dfr,) = torch.autograd.grad((gr * w).sum(), [fr])
        fg = nhwc(f).to(DEV).requires_grad_(True)
        gg = ops.gram_matrix(fg)
        (dfg,) = torch.autograd.grad((gg * w.to(DEV)).sum(), [fg])
        report(f"gram {N}x{H}x{W} C={C} G", rel_l2(gg, gr), 2e-5)
        report(f"gram {N}x{H}x{W} C={C} dF", rel_l2(nchw(dfg), dfr), 1e-4)


def test_multi_style_loss_vs_oracle():
    """VGG-topology features + weighted multi-reference Gram loss against the oracle's CPU restatement of the SAME
    build-defined definition (parity unpinned against the reference: it has no such loss)."""
    import style_loss
    from oracle import restatement as R
    div, shape = 4, (2, 3, 32, 48)
    sd = R.make_state_dict(R.vgg_spec(div), 77)
    feats = style_loss.VGGFeatures(width_div=div)
    feats.load_state_dict(sd)
    feats.to(DEV)
    styles = [R.make_input(shape, 80 + k) for k in range(3)]
    weights = (0.5, 0.3, 0.2)
    y = R.make_input(shape, 90)
    yr = y.clone().requires_grad_(True)
    lr_ = R.multi_style_gram_loss(sd, yr, styles, list(weights))
    (gyr,) = torch.autograd.grad(lr_, [yr])
    loss_mod = style_loss.MultiStyleGramLoss(feats, [s.to(DEV) for s in styles], weights)
    yg = y.to(DEV).requires_grad_(True)
    fg = feats(yg)
    fr = R.vgg_features(sd, y)
    for l, (a, b) in enumerate(zip(fg, fr)):
        report(f"vgg tap {l}", rel_l2(nchw(a), b), 2e-5)
    lg = loss_mod(yg)
    (gyg,) = torch.autograd.grad(lg, [yg])
    report("multi-style loss value", abs(float(lg) - float(lr_)) / abs(float(lr_)), 1e-4)
    report("multi-style loss d/dy", rel_l2(gyg, gyr), 1e-3)
