"""CPU: the C-ABI library builds, loads, exports every symbol include/mstg_hip.h declares, validates arguments on
the host, and the Python boundary mirrors the reference's module surface.  No kernel is launched here."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import PKG, ROOT


@pytest.fixture(scope="module")
def lib():
    from mstg_hip import _lib, build
    build.build(verbose=False)
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mstg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mstg_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    from mstg_hip import _lib
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mstg_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.mstg_arch() == b"gfx950" and b"gfx950" in lib.mstg_version()


def test_conv_desc_layout():
    from mstg_hip._lib import ConvDesc
    assert C.sizeof(ConvDesc) == 21 * 4


def test_host_side_validation(lib):
    from mstg_hip import ops
    d = ops.make_desc(1, 16, 16, 16, 16, 16, 8, 3, 1, 1, 1)
    assert lib.mstg_conv2d_fwd(C.byref(d), None, None, None, None, None, 0, None) == -1  # null pointers
    assert lib.mstg_conv2d_workspace_bytes(C.byref(d)) >= 9 * 16 * 16 * 4
    d = ops.make_desc(1, 16, 16, 16, 15, 16, 8, 3, 1, 1, 1)                               # wrong Ho
    assert lib.mstg_conv2d_fwd(C.byref(d), 1, 1, None, 1, None, 0, None) == -1
    assert b"Ho" in lib.mstg_last_error()
    d = ops.make_desc(1, 16, 16, 16, 8, 8, 8, 3, 2, 1, 1)                                 # stride-2 3x3: not on the path
    assert lib.mstg_conv2d_fwd(C.byref(d), 1, 1, None, 1, None, 0, None) == -5
    d = ops.make_desc(1, 16, 16, 16, 32, 32, 8, 4, 2, 1, 1, transposed=1)
    assert lib.mstg_conv2d_wgrad_workspace_bytes(C.byref(d)) > 0
    assert lib.mstg_window_attn_core_fwd(1, 1, 1, 6, 8, 16, None) == -1                   # H not a multiple of 4
    assert lib.mstg_window_attn_core_fwd(1, 1, 1, 8, 8, 512, None) == -5                  # C > 256 not in this build
    assert lib.mstg_norm_workspace_bytes(2, 64 * 64, 16) > 0
    assert lib.mstg_adam_step_flat(1, 1, 1, 1, 4, 1e-3, 0.5, 0.999, 1e-8, 0, None, None) == -1  # step counts from 1


def test_no_cpu_fallback():
    import enhanced_generator as eg
    g = eg.EnhancedGenerator(8, 0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        g(torch.zeros(1, 3, 32, 32))


def test_module_surface_and_state_dict_keys():
    import enhanced_generator as eg
    import plain_generator
    from oracle import restatement as R
    for C_ in (8, 16):
        g = eg.EnhancedGenerator(channels=C_, num_transformer_blocks=0)
        assert [(k, tuple(v.shape)) for k, v in g.state_dict().items()] == R.generator_spec(C_)
        d = eg.EnhancedDiscriminator(channels=C_)
        assert {k: tuple(v.shape) for k, v in d.state_dict().items()} == dict(R.discriminator_spec(C_))
        p = plain_generator.Generator(channels=C_)
        assert {k: tuple(v.shape) for k, v in p.state_dict().items()} == dict(R.plain_generator_spec(C_))
    g = eg.EnhancedGenerator()  # reference defaults: channels=64, num_transformer_blocks=3
    assert len(g.transformer_blocks) == 3 and g.initial[0].weight.shape == (64, 3, 7, 7)
    g1 = eg.EnhancedGenerator(channels=16, num_transformer_blocks=1)  # what every reference caller builds
    block = sum(p.numel() for p in g1.transformer_blocks.parameters())
    assert block == sum(int(torch.tensor(s_).prod()) for _, s_ in R.transformer_block_spec(64)) > 0  # build-defined block (F1)
    assert sum(p.numel() for p in g1.parameters()) == 168611 + block  # SURVEY.md a1 for everything the reference defines
    assert [(k, tuple(v.shape)) for k, v in g1.state_dict().items()] == R.generator_spec_with_blocks(16, 1)
    g1.gradient_checkpointing_enable()
    assert g1.use_checkpointing
    for name in ("LocalAttention", "MultiScaleBlock", "EnhancedGenerator", "EnhancedDiscriminator"):
        assert hasattr(eg, name)
    att = eg.LocalAttention(16)
    assert att.window_size == 8 and att.qkv.weight.shape == (48, 16, 1, 1)


def test_init_matches_reference_distribution():
    """Same layer subclasses + same _init_weights => same RNG stream as the reference ctor for a given seed:
    kaiming-normal fan_out for every conv, zero biases (enhanced_generator.py:152-161)."""
    import enhanced_generator as eg
    torch.manual_seed(42)
    g = eg.EnhancedGenerator(channels=16, num_transformer_blocks=0)
    w = g.down1[0].weight
    fan_out = w.shape[0] * 16
    assert abs(float(w.std()) - (2.0 / fan_out) ** 0.5) < 0.1 * (2.0 / fan_out) ** 0.5
    assert float(g.down1[0].bias.abs().max()) == 0.0
