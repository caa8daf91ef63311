"""CPU: the image-pipeline oracle (oracle/image_ref.py) against Pillow itself -- the reference's own dependency for these steps
(pretrain.py:32-57, batch_process_images.py:183-233) -- and the library's host-side coefficient builder against the oracle.
Integer work: bit-exact."""
import ctypes as C

import numpy as np
import pytest

from oracle import image_ref as IR

PIL = pytest.importorskip("PIL")


def _img(h, w, seed):
    rs = np.random.RandomState(seed)
    base = rs.randint(0, 256, size=(h // 4 + 2, w // 4 + 2, 3)).astype(np.uint8)  # low-frequency content + noise, like a photo
    img = np.kron(base, np.ones((4, 4, 1), dtype=np.uint8))[:h, :w]
    return np.ascontiguousarray((img.astype(np.int32) + rs.randint(-20, 21, size=img.shape)).clip(0, 255).astype(np.uint8))


@pytest.mark.parametrize("filt", [IR.BILINEAR, IR.LANCZOS])
@pytest.mark.parametrize("shape,size", [((300, 400), (341, 256)), ((256, 256), (256, 192)), ((97, 301), (256, 82)), ((64, 48), (256, 341)),
                                        ((500, 333), (170, 256)), ((256, 171), (333, 500))])
def test_resample_restatement_matches_pillow(filt, shape, size):
    img = _img(shape[0], shape[1], 7)
    assert np.array_equal(IR.resample_numpy(img, size, filt), IR.pil_resize(img, size, filt))


def test_library_coefficient_tables_match_restatement():
    from mstg_hip import _lib, build
    build.build(verbose=False)
    lib = _lib.load()
    for filt in (IR.BILINEAR, IR.LANCZOS):
        for a, b in ((300, 256), (256, 341), (97, 256), (1024, 256), (256, 256), (171, 500)):
            ks, kk, bounds = IR.coeffs(a, b, filt)
            assert lib.mstg_resample_ksize(a, b, filt) == ks
            kk2, b2 = np.zeros_like(kk), np.zeros_like(bounds)
            assert lib.mstg_resample_coeffs(a, b, filt, kk2.ctypes.data, b2.ctypes.data) == 0
            assert np.array_equal(kk, kk2) and np.array_equal(bounds, b2), (filt, a, b)


def test_dataset_item_and_output_conversion_restatements():
    import random
    img = _img(300, 420, 3)
    grid = IR.draw_grid_mask(random.Random(42))
    masked, image, mask = IR.dataset_item_ref(img, grid)
    m2, i2, k2 = IR.dataset_item_ref(img, grid, resize=IR.resample_numpy)
    assert np.array_equal(masked, m2) and np.array_equal(image, i2) and np.array_equal(mask, k2)
    assert masked.shape == (3, 256, 256) and image.min() >= -1 and image.max() <= 1
    frac = 1.0 - mask.mean()
    assert 0.15 < frac < 0.65  # 40 % of 64 cells on average
    y = np.random.RandomState(0).standard_normal((3, 64, 64)).astype(np.float32)
    u = IR.output_to_u8(y)
    assert u.dtype == np.uint8 and u.shape == (64, 64, 3) and u.min() == 0 and u.max() == 255
    out = IR.process_cyclegan_ref(lambda x: -x, img)  # "model" = colour inversion: the geometry must survive the round trip
    assert out.shape == img.shape
    assert np.array_equal(out, IR.process_cyclegan_ref(lambda x: -x, img, resize=IR.resample_numpy))


def test_blend_restatements():
    """batch_process_images.py:304-312, :340-342: the numpy expressions themselves (float64 products, clip, truncation)."""
    a = np.array([[[0, 10, 255]], [[200, 100, 50]]], dtype=np.uint8)
    b = np.array([[[255, 20, 0]], [[100, 200, 250]]], dtype=np.uint8)
    out = IR.blend_simple(a, b, 0.8)
    assert out.dtype == np.uint8 and out.tolist() == [[[204, 18, 50]], [[120, 180, 210]]]  # 255 * (1 - 0.8) = 50.99999999999999 in float64: truncation gives 50
    assert np.array_equal(IR.blend_simple(a, b, 0.0), a) and np.array_equal(IR.blend_simple(a, b, 1.0), b)
    wm = np.array([[0.0], [1.0]])
    assert np.array_equal(IR.blend_weight_map(a, b, wm)[0], a[0]) and np.array_equal(IR.blend_weight_map(a, b, wm)[1], b[1])
    assert np.array_equal(IR.blend_weight_map(a, b, np.full((2, 1), 0.8)), out)
    img = _img(120, 200, 9)
    res = IR.process_local_style_ref(lambda x: -x, img, mode="simple", strength=0.5)
    assert res.shape == img.shape and res.dtype == np.uint8


def test_cosine_lr_state_is_torch_shaped():
    """pretrain.py:211-214 / pretrain_resume.py:149-150: the scheduler state travels through torch's CosineAnnealingLR both ways."""
    import torch
    import pretrain

    class _Opt:  # what CosineLR needs of an optimizer
        def __init__(self, lr):
            self.param_groups = [{"lr": lr}]

    ours = pretrain.CosineLR(_Opt(2e-4), T_max=7, eta_min=1e-6)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=2e-4)
    ref = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=7, eta_min=1e-6)
    for _ in range(3):
        ours.step()
        opt.step()
        ref.step()
    assert abs(ours.get_last_lr()[0] - ref.get_last_lr()[0]) <= 1e-12
    sd = ours.state_dict()
    assert {"T_max", "eta_min", "base_lrs", "last_epoch", "_last_lr", "_step_count"} <= set(sd)
    opt2 = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=2e-4)
    ref2 = torch.optim.lr_scheduler.CosineAnnealingLR(opt2, T_max=99, eta_min=0.0)
    ref2.load_state_dict(sd)
    assert ref2.T_max == 7 and ref2.last_epoch == 3 and abs(ref2.get_last_lr()[0] - ours.get_last_lr()[0]) <= 1e-12
    ours2 = pretrain.CosineLR(_Opt(1.0), T_max=1)
    ours2.load_state_dict(ref.state_dict())
    ours2.step()
    opt.step()
    ref.step()
    assert ours2.last_epoch == 4 and abs(ours2.get_last_lr()[0] - ref.get_last_lr()[0]) <= 1e-12
    ours3 = pretrain.CosineLR(_Opt(1.0), T_max=1)
    ours3.load_state_dict({"T_max": 7, "eta_min": 1e-6, "base_lr": 2e-4, "last_epoch": 3})  # the round-2 layout
    assert abs(ours3.get_last_lr()[0] - ours.get_last_lr()[0]) <= 1e-15
