"""Device image pipeline (csrc/image.hip, mstg_hip/image.py) and the loops built on it (pretrain.py, enhanced_train.train), through
the C ABI.  Integer work is held bit-exact: against Pillow where it is importable (the reference's own dependency for these
steps), and always against the numpy restatement of Pillow's resampling that tests/test_image_cpu.py pins to Pillow."""
import random

import numpy as np
import pytest
import torch

from oracle import image_ref as IR

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mstg_hip import _lib
    _lib.load()


def _img(h, w, seed):
    rs = np.random.RandomState(seed)
    base = rs.randint(0, 256, size=(h // 4 + 2, w // 4 + 2, 3)).astype(np.uint8)
    img = np.kron(base, np.ones((4, 4, 1), dtype=np.uint8))[:h, :w]
    return np.ascontiguousarray((img.astype(np.int32) + rs.randint(-20, 21, size=img.shape)).clip(0, 255).astype(np.uint8))


def _pil_or_numpy():
    try:
        import PIL  # noqa: F401
        return IR.pil_resize
    except ImportError:
        return IR.resample_numpy


@pytest.mark.parametrize("filt", [IR.BILINEAR, IR.LANCZOS])
@pytest.mark.parametrize("shape,size", [((300, 400), (341, 256)), ((256, 256), (256, 192)), ((97, 301), (256, 82)), ((64, 48), (256, 341)),
                                        ((500, 333), (170, 256)), ((1200, 900), (192, 256)), ((256, 171), (333, 500))])
def test_resize_bit_exact(filt, shape, size):
    from mstg_hip import image as dimg
    img = _img(shape[0], shape[1], 7)
    out = dimg.resize_u8(torch.from_numpy(img).to(DEV), size, filt).cpu().numpy()
    ref = _pil_or_numpy()(img, size, filt)
    assert out.shape == ref.shape and np.array_equal(out, ref), f"{int((out != ref).sum())} bytes differ"


def test_dataset_item_bit_exact_and_mask_stream():
    """MonetPhotoDataset.__getitem__ on the device: (masked, image, mask) equal the CPU composition bit for bit (ToTensor /
    Normalize are two fp32 operations per byte), and a seeded run draws the same 64 cells as the reference's loop would."""
    import pretrain
    arrays = [_img(300, 420, 3), _img(512, 384, 4), _img(256, 256, 5)]
    ds = pretrain.MonetPhotoDataset(arrays=arrays, device=DEV)
    random.seed(42)
    got = [ds[i] for i in range(3)]
    rng = random.Random(42)
    for (masked, image, mask), arr in zip(got, arrays):
        grid = IR.draw_grid_mask(rng)
        m_ref, i_ref, k_ref = IR.dataset_item_ref(arr, grid, resize=_pil_or_numpy())
        assert np.array_equal(image.cpu().numpy(), i_ref)
        assert np.array_equal(mask.cpu().numpy(), k_ref)
        assert np.array_equal(masked.cpu().numpy(), m_ref)


def test_output_conversion_bit_exact():
    from mstg_hip import image as dimg
    y = (torch.randn((3, 96, 160), generator=torch.Generator().manual_seed(1)) * 0.8).clamp(-1.2, 1.2)
    y[0, 0, :4] = torch.tensor([-1.0, 1.0, 0.0, 0.99999994])
    out = dimg.to_u8(y.to(DEV)).cpu().numpy()
    assert np.array_equal(out, IR.output_to_u8(y.numpy()))


@pytest.mark.parametrize("shape", [(300, 420), (420, 300), (256, 256), (1100, 1000)])
def test_process_cyclegan_matches_reference_composition(shape):
    """process_cyclegan (batch_process_images.py:176-236) end to end on the device with the real generator in the middle: the
    bytes that reach the model and the bytes that leave equal the PIL / numpy composition, given the same forward."""
    import enhanced_generator as eg
    from mstg_hip import image as dimg
    from oracle import restatement as R
    m = eg.EnhancedGenerator(channels=16, num_transformer_blocks=1)
    m.load_state_dict(R.make_state_dict(R.generator_spec_with_blocks(16, 1), 77))
    m.to(DEV).eval()
    img = _img(shape[0], shape[1], 11)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = dimg.process_cyclegan(m, torch.from_numpy(img).to(DEV)).cpu().numpy()

        def model_fn(x):  # the SAME forward for the CPU composition: only the pre/post-processing is under test here
            with torch.no_grad():
                return m(torch.from_numpy(np.ascontiguousarray(x)).to(DEV)).cpu().numpy()
        ref = IR.process_cyclegan_ref(model_fn, img, resize=_pil_or_numpy())
    assert out.shape == ref.shape == ((shape[0], shape[1], 3) if shape[0] * shape[1] <= 1024 * 1024 else out.shape)
    assert np.array_equal(out, ref), f"{int((out != ref).sum())} of {out.size} bytes differ"


@pytest.mark.parametrize("strength", [0.8, 0.0, 1.0, 0.3, 0.55, 1.0 / 3.0, 1.2])
def test_blend_simple_bit_exact(strength):
    """mode 'simple' of process_local_style (batch_process_images.py:304-312): byte arithmetic, bit-exact against the numpy
    restatement (every (orig, styled) byte pair occurs: 256 x 256 pixels per channel)."""
    from mstg_hip import image as dimg
    a = np.repeat(np.arange(256, dtype=np.uint8)[:, None, None], 256, axis=1).repeat(3, axis=2)
    b = np.ascontiguousarray(a.transpose(1, 0, 2))
    b[..., 1] = b[::-1, :, 1]
    out = dimg.blend_u8(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), strength=strength).cpu().numpy()
    ref = IR.blend_simple(a, b, strength)
    assert np.array_equal(out, ref), f"{int((out != ref).sum())} bytes differ at strength {strength}"


def test_blend_weight_map_bit_exact():
    """The per-pixel blend of the 'enhanced' mode (batch_process_images.py:340-342, :352) with a weight map made the way the
    reference makes it (strength everywhere, stronger in one region, detail_weight in another), plus random float64 weights."""
    from mstg_hip import image as dimg
    rs = np.random.RandomState(5)
    a, b = _img(200, 312, 21), _img(200, 312, 22)
    strength, detail = 0.8, 0.7
    weight = np.ones((200, 312), dtype=float) * strength
    weight[:60] = min(strength + 0.2, 1.0)
    weight[rs.rand(200, 312) > 0.7] = max(strength - 0.3 * detail, 0.0)
    for wm in (weight, rs.rand(200, 312), np.zeros((200, 312)), np.ones((200, 312))):
        out = dimg.blend_u8(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), weight_map=torch.from_numpy(wm)).cpu().numpy()
        ref = IR.blend_weight_map(a, b, wm)
        assert np.array_equal(out, ref), f"{int((out != ref).sum())} bytes differ"
    with pytest.raises(RuntimeError):
        dimg.blend_u8(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV))
    with pytest.raises(RuntimeError):
        dimg.blend_u8(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), weight_map=torch.zeros(3, 3))


@pytest.mark.parametrize("shape,mode", [((300, 420), "simple"), ((420, 300), "weight_map"), ((256, 256), "simple"), ((200, 333), "styled")])
def test_process_local_style_matches_reference_composition(shape, mode):
    """process_local_style (batch_process_images.py:255-441) on the device for the byte-arithmetic modes, real generator in the
    middle, against the PIL / numpy composition given the same forward."""
    import warnings

    import enhanced_generator as eg
    from mstg_hip import image as dimg
    from oracle import restatement as R
    m = eg.EnhancedGenerator(channels=16, num_transformer_blocks=1)
    m.load_state_dict(R.make_state_dict(R.generator_spec_with_blocks(16, 1), 78))
    m.to(DEV).eval()
    img = _img(shape[0], shape[1], 12)
    wm = np.random.RandomState(3).rand(256, 256) if mode == "weight_map" else None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = dimg.process_local_style(m, torch.from_numpy(img).to(DEV), mode=mode, strength=0.65, weight_map=wm).cpu().numpy()

        def model_fn(x):
            with torch.no_grad():
                return m(torch.from_numpy(np.ascontiguousarray(x)).to(DEV)).cpu().numpy()
        ref = IR.process_local_style_ref(model_fn, img, mode=mode, strength=0.65, weight_map=wm, resize=_pil_or_numpy())
    assert out.shape == ref.shape == (shape[0], shape[1], 3)
    assert np.array_equal(out, ref), f"{int((out != ref).sum())} of {out.size} bytes differ"


def test_masked_l1_and_clip_grad_norm_vs_torch():
    from mstg_hip import ops
    g = torch.Generator().manual_seed(2)
    gen = (torch.rand((2, 3, 64, 64), generator=g) * 2 - 1)
    real = (torch.rand((2, 3, 64, 64), generator=g) * 2 - 1)
    mask = (torch.rand((2, 3, 64, 64), generator=g) < 0.6).float()
    a = gen.to(DEV).requires_grad_(True)
    loss = ops.masked_l1_loss(a, real.to(DEV), mask.to(DEV))
    (loss * 3.0).backward()
    ar = gen.clone().requires_grad_(True)
    lr = torch.nn.L1Loss()(ar * (1 - mask), real * (1 - mask))
    (lr * 3.0).backward()
    assert abs(float(loss) - float(lr)) <= 1e-6 * abs(float(lr))
    assert torch.allclose(a.grad.cpu(), ar.grad, rtol=0, atol=1e-9)
    for scale in (0.01, 10.0):
        flat = (torch.randn(100003, generator=g) * scale)
        p = torch.nn.Parameter(torch.zeros_like(flat))
        p.grad = flat.clone()
        nr = torch.nn.utils.clip_grad_norm_([p], max_norm=1.0)
        fd = flat.to(DEV)
        n = ops.clip_grad_norm_flat_(fd, 1.0)
        assert abs(float(n) - float(nr)) <= 1e-5 * float(nr)
        assert torch.allclose(fd.cpu(), p.grad, rtol=1e-5, atol=1e-9)


def test_pretrain_and_train_loops_run_and_checkpoints_round_trip(tmp_path):
    """pretrain.train (pretrain.py:99-230) and enhanced_train.train (enhanced_train.py:154-208) on synthetic image arrays: losses
    finite and moving, checkpoints carry the reference's keys and load back (strict) into fresh modules."""
    import enhanced_train
    import plain_generator
    import pretrain
    arrays_a = [_img(270 + 3 * i, 300 + 5 * i, 20 + i) for i in range(4)]
    arrays_b = [_img(300 + 2 * i, 260 + 7 * i, 40 + i) for i in range(4)]
    ds = (pretrain.MonetPhotoDataset(arrays=arrays_a, device=DEV, img_size=64), pretrain.MonetPhotoDataset(arrays=arrays_b, device=DEV, img_size=64))
    gen, hist = pretrain.train(None, tmp_path / "pre", num_epochs=50, batch_size=2, channels=8, datasets=ds, log_every=1000)
    first, last = np.mean([h[2] for h in hist[:4]]), np.mean([h[2] for h in hist[-4:]])
    assert np.isfinite(first) and np.isfinite(last) and last < first, (first, last)
    ck = torch.load(tmp_path / "pre" / "generator_pretrain_epoch_50.pth", map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "loss"} and ck["epoch"] == 49
    g2 = plain_generator.Generator(channels=8)
    g2.load_state_dict(ck["model_state_dict"])
    for (k, v), (_, v2) in zip(gen.state_dict().items(), g2.state_dict().items()):
        assert torch.equal(v.cpu(), v2), k
    ds256 = (pretrain.MonetPhotoDataset(arrays=arrays_a, device=DEV, img_size=64), pretrain.MonetPhotoDataset(arrays=arrays_b, device=DEV, img_size=64))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, hist2 = enhanced_train.train(None, tmp_path / "gan", num_epochs=2, batch_size=2, channels=8, datasets=ds256, log_every=1,
                                            save_every=2)
    assert len(hist2) == 4 and all(np.isfinite(v) for _, _, d in hist2 for v in d.values())
    fresh = enhanced_train.EnhancedCycleGAN(channels=8, num_transformer_blocks=1, device=torch.device(DEV))
    assert fresh.load_models(tmp_path / "gan", 2) == 2
    for k, v in model.G_AB.state_dict().items():
        assert torch.equal(v, fresh.G_AB.state_dict()[k]), k


def test_optimizer_and_scheduler_state_are_torch_shaped(tmp_path):
    """pretrain.py:210-216 saves torch.optim.Adam / CosineAnnealingLR state and pretrain_resume.py:146-150 loads it into those
    classes: FlatAdam / CosineLR must write that layout and read it back, in both directions, and a resumed step must be the step
    the un-interrupted run would have taken."""
    import pretrain
    from mstg_hip.optim import FlatAdam
    torch.manual_seed(0)
    shapes = [(8, 3, 4, 4), (8,), (5, 7), (3,)]
    base = [torch.randn(s) for s in shapes]
    grads = [[torch.randn(s) for s in shapes] for _ in range(4)]
    for g_ in grads:
        g_[3] = None  # a parameter that never receives a gradient (style_encoder at blocks = 0): torch keeps no state for it
    ours = [torch.nn.Parameter(t.clone().to(DEV)) for t in base]
    ref = [torch.nn.Parameter(t.clone()) for t in base]
    fo, to = FlatAdam(ours, lr=2e-4, betas=(0.5, 0.999)), torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999))

    def step_both(k, flat_opt, torch_opt, ps_ref):
        flat_opt.zero_grad()
        for p, p_r, g_ in zip(flat_opt.params, ps_ref, grads[k]):
            p_r.grad = None if g_ is None else g_.clone()
            if g_ is not None:
                p.grad.copy_(g_)
        flat_opt.step()
        torch_opt.step()

    for k in range(3):
        step_both(k, fo, to, ref)
    sd = fo.state_dict()
    assert set(sd) == {"state", "param_groups"} and set(sd["state"]) == {0, 1, 2} and sd["param_groups"][0]["params"] == [0, 1, 2, 3]
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 3.0
    torch.save({"optimizer_state_dict": sd}, tmp_path / "o.pth")
    sd = torch.load(tmp_path / "o.pth", map_location="cpu", weights_only=True)["optimizer_state_dict"]
    # (a) our state -> torch.optim.Adam on fresh CPU parameters holding our current values
    ref2 = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ours]
    to2 = torch.optim.Adam(ref2, lr=1.0)
    to2.load_state_dict(sd)
    assert to2.param_groups[0]["lr"] == 2e-4 and tuple(to2.param_groups[0]["betas"]) == (0.5, 0.999)
    # (b) torch's state -> FlatAdam on fresh GPU parameters
    ours2 = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref]
    fo2 = FlatAdam(ours2, lr=1.0, betas=(0.9, 0.9))
    fo2.load_state_dict(to.state_dict())
    assert fo2.step_count == 3 and fo2.param_groups[0]["lr"] == 2e-4 and fo2.param_groups[0]["betas"] == (0.5, 0.999)
    # the 4th step from every restored state equals the un-interrupted one
    step_both(3, fo, to, ref)
    step_both(3, fo2, to2, ref2)
    for a, b, c, d in zip(ours, ref, ours2, ref2):
        assert torch.allclose(a.detach().cpu(), b.detach(), rtol=1e-6, atol=1e-7)
        assert torch.allclose(c.detach().cpu(), b.detach(), rtol=1e-6, atol=1e-7)
        assert torch.allclose(d.detach(), b.detach(), rtol=1e-6, atol=1e-7)
    # round-2 flat layout still loads
    fo3 = FlatAdam([torch.nn.Parameter(p.detach().clone()) for p in ours], lr=1.0)
    fo3.load_state_dict({"step": 4, "exp_avg": fo.exp_avg.clone(), "exp_avg_sq": fo.exp_avg_sq.clone(), "param_groups": [{"lr": 3e-4}]})
    assert fo3.step_count == 4 and fo3.param_groups[0]["lr"] == 3e-4


def test_pretrain_resume_and_convert_model(tmp_path):
    """pretrain_resume.py:134-157 on a checkpoint this package wrote, and on one written the way the REFERENCE writes it (torch's own
    Adam + CosineAnnealingLR state); convert_model.py:12-29 wrapper flattening and the inference callers' load convention."""
    import convert_model
    import enhanced_generator as eg
    import plain_generator
    import pretrain
    import pretrain_resume
    from oracle import restatement as R
    arrays_a = [_img(90 + 3 * i, 100 + 5 * i, 60 + i) for i in range(2)]
    ds = (pretrain.MonetPhotoDataset(arrays=arrays_a, device=DEV, img_size=64), pretrain.MonetPhotoDataset(arrays=arrays_a, device=DEV, img_size=64))
    gen, _ = pretrain.train(None, tmp_path / "a", num_epochs=2, batch_size=2, channels=8, datasets=ds, log_every=1000, save_every=2)
    path = tmp_path / "a" / "generator_pretrain_epoch_2.pth"
    g2 = plain_generator.Generator(channels=8).to(DEV)
    step = pretrain.PretrainStep(g2)
    sched = pretrain.CosineLR(step.optimizer, T_max=2, eta_min=1e-6)
    assert pretrain.load_checkpoint(path, g2, step.optimizer, sched, DEV) == 2
    assert step.optimizer.step_count > 0 and sched.last_epoch == 2
    for (k, v), (_, v2) in zip(gen.state_dict().items(), g2.state_dict().items()):
        assert torch.equal(v, v2), k
    # a checkpoint the way the reference writes it: plain torch modules / optimizer / scheduler on the CPU
    g_ref = plain_generator.Generator(channels=8)
    opt = torch.optim.Adam(g_ref.parameters(), lr=2e-4, betas=(0.5, 0.999))
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10, eta_min=1e-6)
    for p in g_ref.parameters():
        p.grad = torch.full_like(p, 1e-3)
    opt.step()
    sch.step()
    torch.save({"epoch": 0, "model_state_dict": g_ref.state_dict(), "optimizer_state_dict": opt.state_dict(),
                "scheduler_state_dict": sch.state_dict(), "loss": 0}, tmp_path / "ref.pth")
    gen3, hist = pretrain_resume.train(None, tmp_path / "b", num_epochs=1, batch_size=2, datasets=ds, log_every=1000, channels=8,
                                       resume_path=tmp_path / "ref.pth", save_every=1)
    assert len(hist) == 2 and all(np.isfinite(h[2]) for h in hist)
    # convert_model: wrappers -> bare state dicts; the loading convention of the inference callers
    sd = R.make_state_dict(R.generator_spec_with_blocks(8, 1), 5)
    torch.save({"epoch": 20, "G_BA_state_dict": sd}, tmp_path / "G_BA_epoch_20.pth")
    assert convert_model.convert_model(tmp_path / "G_BA_epoch_20.pth", tmp_path / "flat.pth")
    flat = torch.load(tmp_path / "flat.pth", map_location="cpu", weights_only=True)
    assert list(flat) == list(sd) and all(torch.equal(flat[k], sd[k]) for k in sd)
    assert convert_model.flatten_checkpoint({"epoch": 3, "model_state_dict": {"w": 1}}) == {"w": 1}
    assert convert_model.flatten_checkpoint({"epoch": 3, "state_dict": {"w": 2}}) == {"w": 2}
    assert convert_model.flatten_checkpoint({"epoch": 3, "w": 4, "G_x": 5}) == {"w": 4}
    assert not convert_model.convert_model(tmp_path / "missing.pth", tmp_path / "x.pth")
    for path_ in (tmp_path / "G_BA_epoch_20.pth", tmp_path / "flat.pth"):
        m = convert_model.load_generator(path_, device=DEV, direction="BA")
        assert isinstance(m, eg.EnhancedGenerator) and not m.training and m.initial[0].weight.shape[0] == 8
        assert torch.equal(m.state_dict()["up2.0.weight"].cpu(), sd["up2.0.weight"])
