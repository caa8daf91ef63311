"""fp16 inference path (BASELINE config #5; csrc/infer_f16.hip, mstg_hip/infer.py) through the C ABI.

Per kernel: against fp32 torch on the CPU evaluated on the SAME fp16-rounded inputs and filters, so that what is measured is the
kernel's arithmetic (fp16 products accumulated in fp32, one fp16 rounding at the store) -- tolerance 2e-3 relative L2 (fp16 has
11 significant bits: 4.9e-4 per rounding).  Whole generator: against this build's own fp32 HIP path (which the golden vectors
pin to the reference), with the tolerance the north star's survey measured for fp16 (1.4-2.3 % drift at the
pre-tanh tap for torch autocast; here statistics, softmax and accumulation stay fp32 and the measured drift is printed)."""
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
    _lib.load()


def report(name, err, tol):
    print(f"  [parity] {name:64s} rel-L2 {err:.2e} (tol {tol:.0e})")
    assert err <= tol, f"{name}: {err:.3e} > {tol:.0e}"


def h(t):
    """round to fp16 and back (CPU, fp32)"""
    return t.half().float()


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def stats_of(y_nchw):
    mu = y_nchw.mean(dim=(2, 3))
    var = y_nchw.var(dim=(2, 3), unbiased=False)
    return torch.stack([mu, torch.rsqrt(var + 1e-5)], dim=-1)  # (N, C, 2)


def norm_relu(x_nchw, st):
    return F.relu((x_nchw - st[..., 0][:, :, None, None]) * st[..., 1][:, :, None, None])


CONV_CASES = [
    # name, kind, N, H, W, Cin, Cout, K, stride, pad, normalise-on-load
    ("stem 7x7 3->16 (NCHW fp32 image)", 0, 2, 32, 48, 3, 16, 7, 1, 3, False),
    ("k4 s2 16->32 + norm on load", 0, 2, 32, 48, 16, 32, 4, 2, 1, True),
    ("k4 s2 32->64", 0, 1, 40, 24, 32, 64, 4, 2, 1, False),
    ("k4 s2 32->64 ragged output 10x6", 0, 1, 20, 12, 32, 64, 4, 2, 1, True),
    ("convT 64->32", 1, 2, 8, 12, 64, 32, 4, 2, 1, False),
    ("convT 32->16 ragged 20x36", 1, 1, 20, 36, 32, 16, 4, 2, 1, False),
    ("1x1 16->16 + norm on load", 0, 2, 32, 32, 16, 16, 1, 1, 0, True),
    ("1x1 64->64 + norm on load", 0, 1, 24, 20, 64, 64, 1, 1, 0, True),
    ("1x1 32->32", 0, 1, 16, 16, 32, 32, 1, 1, 0, False),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_f16_conv(case):
    from mstg_hip.infer import _PackedConv
    name, kind, N, H, W, Cin, Cout, K, s, p, norm = case
    image = Cin == 3
    w = rnd((Cin, Cout, K, K) if kind == 1 else (Cout, Cin, K, K), 1, (2.0 / (Cin * K * K)) ** 0.5)
    b = rnd((Cout,), 2, 0.1)
    x = rnd((N, Cin, H, W), 3) * 1.5 + 0.3
    pc = _PackedConv(kind, [w.to(DEV)], [b.to(DEV)], Cin, Cout, K, s, p, src_nchw_f32=int(image))
    if image:
        xin, xr = x.to(DEV), h(x)
    else:
        xin, xr = x.permute(0, 2, 3, 1).contiguous().half().to(DEV), h(x)
    st = None
    if norm:
        st = stats_of(xr)
        xr = h(norm_relu(xr, st))  # the kernel rounds the normalised activation to fp16 in LDS
    y, ost = pc(xin, in_stats=None if st is None else st.to(DEV).contiguous(), want_stats=True)
    if kind == 1:
        ref = F.conv_transpose2d(xr, h(w), b, stride=2, padding=1)
    else:
        ref = F.conv2d(xr, h(w), b, stride=s, padding=p)
    report(name + " y", rel_l2(y.float().permute(0, 3, 1, 2), ref), 2e-3)
    rst = stats_of(ref)
    report(name + " mean", float((ost[..., 0].cpu() - rst[..., 0]).abs().max() / rst[..., 0].abs().max().clamp_min(1e-3)), 2e-3)
    report(name + " rstd", rel_l2(ost[..., 1], rst[..., 1]), 2e-3)


@pytest.mark.parametrize("ch,N,H,W", [(16, 2, 32, 48), (32, 1, 24, 40), (64, 1, 16, 20), (16, 1, 8, 8)])
def test_f16_msblock_branches(ch, N, H, W):
    from mstg_hip.infer import _PackedConv
    c4 = ch // 4
    ws = [rnd((c4, ch, 1, 1), 11, (2.0 / ch) ** 0.5)] + [rnd((c4, ch, 3, 3), 12 + i, (2.0 / (9 * ch)) ** 0.5) for i in range(3)]
    bs = [rnd((c4,), 20 + i, 0.1) for i in range(4)]
    x = rnd((N, ch, H, W), 5)
    pc = _PackedConv(2, [w.to(DEV) for w in ws], [b.to(DEV) for b in bs], ch, ch, 3, 1, 4)
    y, ost = pc(x.permute(0, 2, 3, 1).contiguous().half().to(DEV), want_stats=True)
    xr = h(x)
    outs = [F.conv2d(xr, h(ws[0]), bs[0])] + [F.conv2d(xr, h(ws[i]), bs[i], padding=d, dilation=d) for i, d in ((1, 1), (2, 2), (3, 4))]
    ref = torch.cat(outs, dim=1)
    report(f"msblock branches ch{ch} y", rel_l2(y.float().permute(0, 3, 1, 2), ref), 2e-3)
    report(f"msblock branches ch{ch} rstd", rel_l2(ost[..., 1], stats_of(ref)[..., 1]), 2e-3)


def test_f16_head_tanh_nchw():
    from mstg_hip.infer import _PackedConv
    from mstg_hip.ops import ACT_TANH
    N, H, W, Cin = 2, 32, 48, 16
    w, b, x = rnd((3, Cin, 7, 7), 1, 0.05), rnd((3,), 2, 0.1), rnd((N, Cin, H, W), 3)
    pc = _PackedConv(0, [w.to(DEV)], [b.to(DEV)], Cin, 3, 7, 1, 3, dst_nchw=1, act=ACT_TANH)
    y, _ = pc(x.permute(0, 2, 3, 1).contiguous().half().to(DEV))
    ref = torch.tanh(F.conv2d(h(x), h(w), b, padding=3))
    assert y.shape == (N, 3, H, W) and y.dtype == torch.float16
    report("head 7x7 16->3 tanh NCHW", rel_l2(y.float(), ref), 2e-3)


@pytest.mark.parametrize("reg", ["1", "0"])
@pytest.mark.parametrize("C_,N,H,W,norm", [(16, 2, 16, 24, True), (32, 1, 8, 72, True), (64, 1, 8, 8, False), (16, 1, 4, 260, False),
                                           # more windows than persistent waves: every wave of the register-resident kernel walks several
                                           (64, 2, 128, 192, True), (32, 3, 192, 256, False), (16, 2, 256, 320, True)])
def test_f16_local_attention(C_, N, H, W, norm, reg, monkeypatch):
    """reg = 1: attn_f16r_kernel (register-resident chains, the default); 0: attn_f16_kernel (LDS tiles)."""
    monkeypatch.setenv("MSTG_F16_ATTN_REG", reg)
    from mstg_hip.infer import _PackedAttention
    from oracle import restatement as R
    import enhanced_generator as eg
    m = eg.LocalAttention(C_, window_size=4)
    sd = {"p.qkv.weight": rnd((3 * C_, C_, 1, 1), 1, (1.0 / C_) ** 0.5), "p.qkv.bias": rnd((3 * C_,), 2, 0.1),
          "p.proj.weight": rnd((C_, C_, 1, 1), 3, (1.0 / C_) ** 0.5), "p.proj.bias": rnd((C_,), 4, 0.1)}
    m.load_state_dict({k[2:]: v for k, v in sd.items()})
    m.to(DEV)
    pa = _PackedAttention(m)
    x = rnd((N, C_, H, W), 5) * 1.3 + 0.2
    xr, st = h(x), None
    if norm:
        st = stats_of(xr)
        xr = h(norm_relu(xr, st))
    y = pa(x.permute(0, 2, 3, 1).contiguous().half().to(DEV), in_stats=None if st is None else st.to(DEV).contiguous())
    sdr = {k: (h(v) if k.endswith("weight") else v) for k, v in sd.items()}
    ref = R.local_attention(xr, sdr, "p", 4)
    report(f"LocalAttention fp16 C{C_} {H}x{W}", rel_l2(y.float().permute(0, 3, 1, 2), ref), 5e-3)


def test_f16_norm_residual():
    from mstg_hip import infer
    x, r = rnd((2, 32, 24, 40), 1) * 2 + 0.5, rnd((2, 32, 24, 40), 2)
    st = stats_of(h(x))
    y = infer.norm_residual(x.permute(0, 2, 3, 1).contiguous().half().to(DEV), r.permute(0, 2, 3, 1).contiguous().half().to(DEV),
                            st.to(DEV).contiguous())
    report("norm + relu + residual fp16", rel_l2(y.float().permute(0, 3, 1, 2), norm_relu(h(x), st) + h(r)), 1e-3)


def _pair(seed, channels=16):
    import enhanced_generator as eg
    from oracle import restatement as R
    sd = R.make_state_dict(R.generator_spec(channels), seed)
    m = eg.EnhancedGenerator(channels=channels, num_transformer_blocks=0)
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


@pytest.mark.parametrize("shape", [(2, 3, 64, 64), (1, 3, 48, 80), (1, 3, 256, 256)])
def test_f16_generator_vs_fp32_path(shape):
    """Whole forward, fp16 path vs this build's fp32 path (same weights): every stage output, the pre-tanh tap and the image.
    Tolerance 3e-2 relative L2 at the pre-tanh tap (SURVEY.md section 7 measured 1.4-2.3 % for fp16 autocast of the reference),
    2e-2 on the image; the outputs must be finite (the survey saw NaNs from fp16 autocast at channels=16)."""
    from oracle import restatement as R
    m, _ = _pair(401)
    x = R.make_input(shape, 402).to(DEV)
    t32, t16 = {}, {}
    with torch.no_grad():
        y32 = m.forward_taps(x, t32)
        m.half_inference()
        y16 = m.forward_taps(x, t16)
        m.half_inference(False)
        y32b = m(x)
    assert y16.dtype == torch.float16 and y16.shape == y32.shape and torch.isfinite(y16).all()
    assert torch.equal(y32, y32b)  # switching the fast path off restores the fp32 path
    for k in ("down1", "down2", "up1", "up2"):
        report(f"fp16 vs fp32 {shape[2]}x{shape[3]} tap {k}", rel_l2(t16[k].float(), t32[k]), 3e-2)
    report(f"fp16 vs fp32 {shape[2]}x{shape[3]} pre_tanh", rel_l2(t16["pre_tanh"].float(), t32["pre_tanh"]), 3e-2)
    report(f"fp16 vs fp32 {shape[2]}x{shape[3]} out", rel_l2(y16.float(), y32), 2e-2)


@pytest.mark.parametrize("shape", [(2, 3, 64, 64), (1, 3, 256, 256)])
def test_f16_generator_with_transformer_block_vs_fp32_path(shape):
    """What every inference caller of the reference builds (num_transformer_blocks=1: direct_transform.py:35, advanced_transform.py:29,
    batch_process_images.py:95): the fp16 plan runs the block on the fp32 kernels between down2 and up1.  Same bars as the
    block-free model (3e-2 at the taps / pre-tanh, 2e-2 on the image), against this build's fp32 path."""
    import enhanced_generator as eg
    from oracle import restatement as R
    m = eg.EnhancedGenerator(channels=16, num_transformer_blocks=1)
    m.load_state_dict(R.make_state_dict(R.generator_spec_with_blocks(16, 1), 421))
    m.to(DEV).eval()
    x = R.make_input(shape, 422).to(DEV)
    t32, t16 = {}, {}
    with torch.no_grad():
        y32 = m.forward_taps(x, t32)
        m.half_inference()
        y16 = m.forward_taps(x, t16)
        y16b = m(x)
        m.half_inference(False)
    assert y16.dtype == torch.float16 and torch.isfinite(y16).all() and torch.equal(y16, y16b)
    for k in ("down2", "up1", "up2", "pre_tanh"):
        report(f"fp16+block vs fp32 {shape[2]}x{shape[3]} tap {k}", rel_l2(t16[k].float(), t32[k]), 3e-2)
    report(f"fp16+block vs fp32 {shape[2]}x{shape[3]} out", rel_l2(y16.float(), y32), 2e-2)


def test_f16_generator_contract():
    """The reference's callers: eval() + no_grad + strict load_state_dict (direct_transform.py:35-63).  The packed filters follow
    a load_state_dict; autograd-enabled calls keep the fp32 path; bad shapes raise like the reference; other widths raise."""
    import enhanced_generator as eg
    from oracle import restatement as R
    m, sd = _pair(411)
    m.half_inference()
    x = R.make_input((1, 3, 64, 64), 412).to(DEV)
    with torch.no_grad():
        y1 = m(x)
        y1b = m(x)
        assert torch.equal(y1, y1b)  # deterministic
        m.load_state_dict(R.make_state_dict(R.generator_spec(16), 413))
        y2 = m(x)
        m.load_state_dict(sd)
        y3 = m(x)
    assert not torch.equal(y1, y2) and torch.equal(y1, y3)
    yg = m(x)  # autograd on: fp32 training path
    assert yg.dtype == torch.float32 and yg.requires_grad
    with torch.no_grad():
        for bad in ((1, 3, 250, 250), (1, 3, 64, 72 + 4)):
            with pytest.raises(RuntimeError):
                m(torch.zeros(bad, device=DEV))
        with pytest.raises(RuntimeError, match="channels=16"):
            eg.EnhancedGenerator(channels=8, num_transformer_blocks=0).to(DEV).half_inference()  # raises at once, not at first use
        y16in = m(x.half())  # an fp16 image is accepted too
    assert torch.equal(y16in, m.half_inference()._half().forward(x.half().float()))


@pytest.mark.parametrize("shape", [(2, 3, 64, 64), (1, 3, 48, 80), (3, 3, 112, 16), (1, 3, 256, 256)])
def test_f16_residual_folded_into_next_layer_is_bit_identical(shape, monkeypatch):
    """A stage's closing relu(IN(fusion)) + x (enhanced_generator.py:84) is formed by the next layer (:106, :121, :128, :137)
    while it stages its input; MSTG_F16_FOLD_RESIDUAL=0 keeps the separate pass.  Same arithmetic, same rounding: equal bits,
    border tiles and batch > 1 included."""
    from oracle import restatement as R
    m, _ = _pair(441)
    x = R.make_input(shape, 442).to(DEV)
    m.half_inference()
    with torch.no_grad():
        y_fold = m(x)
        monkeypatch.setenv("MSTG_F16_FOLD_RESIDUAL", "0")
        y_pass = m(x)
        y_taps = m.forward_taps(x, {})
    assert torch.isfinite(y_fold).all() and torch.equal(y_fold, y_pass) and torch.equal(y_fold, y_taps)


@pytest.mark.parametrize("kind,cin,cout,k,s,p,hw", [(0, 32, 64, 4, 2, 1, (40, 24)), (1, 64, 32, 4, 2, 1, (12, 20)),
                                                    (1, 32, 16, 4, 2, 1, (16, 16)), (0, 16, 16, 3, 1, 1, (33, 17))])
def test_f16_conv_residual_operand_vs_separate_pass(kind, cin, cout, k, s, p, hw):
    """mstg_f16_conv_fwd_res against mstg_f16_norm_residual followed by mstg_f16_conv_fwd on one layer (ragged sizes)."""
    from mstg_hip import infer
    g = torch.Generator().manual_seed(451)
    N, (H, W) = 2, hw
    wshape = (cin, cout, k, k) if kind == 1 else (cout, cin, k, k)
    w = (torch.randn(wshape, generator=g) * 0.1).to(DEV)
    b = (torch.randn(cout, generator=g) * 0.1).to(DEV)
    f = torch.randn((N, H, W, cin), generator=g).half().to(DEV)
    a = torch.randn((N, H, W, cin), generator=g).half().to(DEV)
    ff = f.float()
    mean = ff.mean(dim=(1, 2))
    rstd = (ff.var(dim=(1, 2), unbiased=False) + 1e-5).rsqrt()
    stats = torch.stack([mean, rstd], dim=-1).contiguous()
    conv = infer._PackedConv(kind, [w], [b], cin, cout, k, s, p)
    y_fold, st_fold = conv(f, in_stats=stats, want_stats=True, residual=a)
    h = infer.norm_residual(f, a, stats)
    y_pass, st_pass = conv(h, want_stats=True)
    assert torch.equal(y_fold, y_pass) and torch.equal(st_fold, st_pass)
    with pytest.raises(RuntimeError):
        conv(f, residual=a)  # a residual operand comes with the statistics of x


def test_config5_forward_1024_fp16():
    """BASELINE config #5 itself: 1024x1024, fp16.  Batch 1 against the fp32 path; batch 64 (the benchmarked shape; activations of
    2.1 GB, > 2^31 elements): finite, and samples 0 / 63 equal the batch-1 results of the same images."""
    from oracle import restatement as R
    m, _ = _pair(421)
    x1 = R.make_input((1, 3, 1024, 1024), 422).to(DEV)
    with torch.no_grad():
        y32 = m(x1)
        m.half_inference()
        y1 = m(x1)
        report("config5 1024x1024 fp16 vs fp32 path, batch 1", rel_l2(y1.float(), y32), 2e-2)
        g = torch.Generator().manual_seed(423)
        x = torch.rand((64, 3, 1024, 1024), generator=g) * 2 - 1
        x[0], x[63] = x1[0].cpu(), x1[0].cpu().flip(-1)
        xd = x.to(DEV)
        y = m(xd)
        y63 = m(xd[63:64])
    assert torch.isfinite(y).all()
    report("config5 fp16 sample 0 of batch 64 vs batch 1", rel_l2(y[0:1].float(), y1.float()), 1e-3)
    report("config5 fp16 sample 63 of batch 64 vs batch 1", rel_l2(y[63:64].float(), y63.float()), 1e-3)


@pytest.mark.parametrize("half", [False, True])
def test_graph_inference_matches_eager(half):
    """hipGraph replay of the inference forward (BASELINE config #1 is launch-bound at batch 1): bit-identical to the eager
    forward, follows a load_state_dict, and a second shape gets its own graph."""
    import time
    from oracle import restatement as R
    m, sd = _pair(431)
    if half:
        m.half_inference()
    x = R.make_input((1, 3, 256, 256), 432).to(DEV)
    x2 = R.make_input((2, 3, 64, 80), 433).to(DEV)
    with torch.no_grad():
        y_eager, y2_eager = m(x), m(x2)
        m.graph_inference()
        y_graph, y2_graph = m(x), m(x2)
        assert torch.equal(y_eager, y_graph) and torch.equal(y2_eager, y2_graph)
        assert torch.equal(m(x), y_eager)  # replay
        m.load_state_dict(R.make_state_dict(R.generator_spec(16), 434))
        y_new = m(x)
        m.graph_inference(False)
        assert torch.equal(y_new, m(x)) and not torch.equal(y_new, y_eager)
        # latency (informational)
        for mode in (False, True):
            m.graph_inference(mode)
            for _ in range(3):
                m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                m(x)
            torch.cuda.synchronize()
            print(f"  [latency] 256x256 batch 1 {'fp16' if half else 'fp32'} {'hipGraph' if mode else 'eager   '}: "
                  f"{(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
