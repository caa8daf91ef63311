"""Inference-only forward of ``EnhancedGenerator`` in fp16 storage / fp16 MFMA / fp32 accumulation (BASELINE config #5).

What the reference's inference scripts run under ``torch.no_grad()`` (direct_transform.py:62-63, batch_process_images.py:210-211,
advanced_transform.py:99-100) is ``EnhancedGenerator.forward`` (enhanced_generator.py:179-228).  This module is that forward on
the kernels of ``csrc/infer_f16.hip``: activations NHWC fp16 between kernels, InstanceNorm statistics produced by the epilogue
of the convolution in front of the norm and applied (with the ReLU) by the consumer while it stages its input, so that per
stage only four tensors are written (conv, attention, branches, fusion) plus the residual sum.

The filters are packed once (inference: weights are frozen) into device blobs; ``EnhancedGenerator.half_inference()`` builds a
``HalfGeneratorPlan`` and re-builds it after a ``load_state_dict``.  Widths: the three stage widths must be 16 / 32 / 64 channels,
i.e. ``channels=16`` -- what every trainer / inference caller of the reference builds (enhanced_train.py:18,
batch_process_images.py:95, advanced_transform.py:12).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_TANH, F16ConvDesc
from .ops import KernelTimer, _p, _stream, _timed


def _desc(kind, N, H, W, Cin, Ho, Wo, Cout, K, stride, pad, dil=1, src_nchw_f32=0, dst_nchw=0, act=ACT_NONE) -> F16ConvDesc:
    return F16ConvDesc(kind, N, H, W, Cin, Ho, Wo, Cout, K, stride, pad, dil, src_nchw_f32, dst_nchw, act)


class _PackedConv:
    """One convolution (kind 0/1) or the four MultiScaleBlock branches (kind 2) with its filter packed for the fp16 kernel."""

    def __init__(self, kind, weights, biases, Cin, Cout, K, stride, pad, src_nchw_f32=0, dst_nchw=0, act=ACT_NONE):
        self.kind, self.Cin, self.Cout, self.K, self.stride, self.pad = kind, Cin, Cout, K, stride, pad
        self.src_nchw_f32, self.dst_nchw, self.act = src_nchw_f32, dst_nchw, act
        lib = _lib.load()
        d = self.desc(1, 64, 64)  # the packed layout depends on the layer, not on N / H / W
        nbytes = lib.mstg_f16_conv_plan_bytes(C.byref(d))
        if nbytes == 0:
            raise RuntimeError(f"mstg_hip fp16 inference: unsupported layer (kind {kind}, {Cin}->{Cout}, k{K} s{stride}): "
                               + lib.mstg_last_error().decode())
        dev = weights[0].device
        self.blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ws = [w.detach().float().contiguous() for w in weights] + [None] * (4 - len(weights))
        bs = [None if b is None else b.detach().float().contiguous() for b in biases] + [None] * (4 - len(biases))
        args = []
        for w, b in zip(ws, bs):
            args += [_p(w), _p(b)]
        _lib.check(lib.mstg_f16_conv_pack(C.byref(d), *args, _p(self.blob), nbytes, _stream()), "mstg_f16_conv_pack")
        self._keep = (ws, bs)  # the pack kernel reads them asynchronously

    def desc(self, N, H, W) -> F16ConvDesc:
        if self.kind == 1:
            Ho, Wo = 2 * H, 2 * W
        else:
            Ho = (H + 2 * self.pad - (self.K - 1) - 1) // self.stride + 1
            Wo = (W + 2 * self.pad - (self.K - 1) - 1) // self.stride + 1
        if self.kind == 2:
            Ho, Wo = H, W
        return _desc(self.kind, N, H, W, self.Cin, Ho, Wo, self.Cout, self.K, self.stride, self.pad, 1, self.src_nchw_f32,
                     self.dst_nchw, self.act)

    def __call__(self, x, in_stats=None, want_stats=False, residual=None):
        """x: NHWC fp16 (N,H,W,Cin) or the NCHW fp32 image -> (y, out_stats or None).  ``residual`` (NHWC fp16 like x, needs
        ``in_stats``): the layer reads relu(IN(x)) + residual, formed while it stages its input."""
        lib = _lib.load()
        if self.src_nchw_f32:
            N, _, H, W = x.shape
        else:
            N, H, W, _ = x.shape
        d = self.desc(N, H, W)
        shape = (N, self.Cout, d.Ho, d.Wo) if self.dst_nchw else (N, d.Ho, d.Wo, self.Cout)
        y = torch.empty(shape, dtype=torch.float16, device=x.device)
        stats = ws = None
        wsb = 0
        if want_stats:
            stats = torch.empty((N, self.Cout, 2), dtype=torch.float32, device=x.device)
            wsb = lib.mstg_f16_conv_partial_bytes(C.byref(d))
            ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=x.device)
        taps = {0: self.K * self.K, 1: 16, 2: 28}[self.kind]  # MultiScaleBlock: 1 + 3 * 9 taps of Cin -> Cin / 4
        cout_eff = self.Cout // 4 if self.kind == 2 else self.Cout
        pix = N * d.Ho * d.Wo / (4 if self.kind == 1 else 1)
        flops = 2.0 * pix * self.Cin * cout_eff * taps
        nbytes = x.numel() * x.element_size() * (2 if residual is not None else 1) + y.numel() * 2
        name = {0: "conv_f16_kernel", 1: "conv_f16_kernel<convT>", 2: "conv_f16_kernel<msblock>"}[self.kind]
        if self.kind == 0 and self.K == 1 and self.stride == 1 and not self.src_nchw_f32 and not self.dst_nchw:
            name = "conv1x1_f16_kernel"  # the LDS-free streaming variant (unless MSTG_F16_DIRECT=0)
        if residual is not None:
            if residual.shape != x.shape or residual.dtype != torch.float16 or not residual.is_contiguous() or in_stats is None:
                raise RuntimeError("mstg_hip fp16 conv: the residual operand must match x (NHWC fp16) and comes with in_stats")
            name += "+res"
        _timed(name, flops, nbytes, lambda: _lib.check(
            lib.mstg_f16_conv_fwd_res(C.byref(d), _p(self.blob), _p(x), _p(in_stats), _p(residual), _p(y), _p(stats), _p(ws), wsb,
                                      _stream()),
            "mstg_f16_conv_fwd_res"), f"k{self.kind} N{N} {H}x{W} {self.Cin}->{self.Cout} k{self.K} s{self.stride}")
        return y, stats


class _PackedAttention:
    def __init__(self, attn):
        self.C = attn.qkv.in_channels
        lib = _lib.load()
        nbytes = lib.mstg_f16_attn_plan_bytes(self.C)
        if nbytes == 0:
            raise RuntimeError(f"mstg_hip fp16 inference: LocalAttention with {self.C} channels is not supported (16 / 32 / 64)")
        if attn.window_size != 4:
            raise RuntimeError("mstg_hip fp16 inference: LocalAttention window_size must be 4")
        t = [attn.qkv.weight, attn.qkv.bias, attn.proj.weight, attn.proj.bias]
        self._keep = [v.detach().float().contiguous() for v in t]
        self.blob = torch.empty(nbytes, dtype=torch.uint8, device=t[0].device)
        _lib.check(lib.mstg_f16_attn_pack(*[_p(v) for v in self._keep], self.C, _p(self.blob), nbytes, _stream()), "mstg_f16_attn_pack")

    def __call__(self, x, in_stats=None):
        N, H, W, Cn = x.shape
        y = torch.empty_like(x)
        _timed(f"attn_f16_kernel<{Cn}>", 12.0 * Cn * Cn * N * H * W, 2.0 * 2 * x.numel(), lambda: _lib.check(
            _lib.load().mstg_f16_attn_fwd(_p(x), _p(in_stats), _p(self.blob), _p(y), N, H, W, Cn, _stream()), "mstg_f16_attn_fwd"),
            f"N{N} {H}x{W} C{Cn}")
        return y


def norm_residual(x, residual, stats):
    N, H, W, Cn = x.shape
    y = torch.empty_like(x)
    _timed("f16_norm_residual_kernel", 0, 2.0 * x.numel() * (3 if residual is not None else 2), lambda: _lib.check(
        _lib.load().mstg_f16_norm_residual(_p(x), _p(residual), _p(stats), _p(y), N, H * W, Cn, _stream()), "mstg_f16_norm_residual"))
    return y


class HalfGeneratorPlan:
    """Packed fp16 copy of an EnhancedGenerator's weights + the fused inference forward."""

    def __init__(self, gen):
        C0 = gen.initial[0].out_channels
        if C0 != 16:
            raise RuntimeError(f"mstg_hip fp16 inference is built for channels=16 (stage widths 16/32/64), got channels={C0}")
        # StructuralTransformerBlocks (every inference caller of the reference builds num_transformer_blocks=1:
        # direct_transform.py:35, advanced_transform.py:29, batch_process_images.py:95) run on the fp32 kernels between down2 and up1:
        # that tensor is H/4 x W/4 x 64 channels, 1/16 of the activation traffic, and the block's LayerNorm / softmax statistics
        # want fp32 anyway.  The fp16 features are widened for it and narrowed again behind it.
        self.blocks = [b for b in gen.transformer_blocks if not getattr(b, "is_identity", False)]
        self.style_vector = gen._style_vector
        if gen.initial[0].weight.device.type != "cuda":
            raise RuntimeError("mstg_hip fp16 inference: move the generator to the GPU first (no CPU path)")
        c = gen.initial[0]
        self.stem = _PackedConv(0, [c.weight], [c.bias], 3, C0, 7, 1, 3, src_nchw_f32=1)
        self.stages = []
        for st, transposed in ((gen.down1, False), (gen.down2, False), (gen.up1, True), (gen.up2, True)):
            conv, attn, ms = st[0], st[3], st[4]
            cin, ch = conv.in_channels, conv.out_channels
            first = _PackedConv(1 if transposed else 0, [conv.weight], [conv.bias], cin, ch, 4, 2, 1)
            branches = _PackedConv(2, [b[0].weight for b in (ms.branch1, ms.branch2, ms.branch3, ms.branch4)],
                                   [b[0].bias for b in (ms.branch1, ms.branch2, ms.branch3, ms.branch4)], ch, ch, 3, 1, 4)
            fusion = _PackedConv(0, [ms.fusion[0].weight], [ms.fusion[0].bias], ch, ch, 1, 1, 0)
            self.stages.append((first, _PackedAttention(attn), branches, fusion))
        c = gen.output[0]
        self.head = _PackedConv(0, [c.weight], [c.bias], C0, 3, 7, 1, 3, dst_nchw=1, act=ACT_TANH)
        self.head_pre = _PackedConv(0, [c.weight], [c.bias], C0, 3, 7, 1, 3, dst_nchw=1, act=ACT_NONE)  # parity taps only

    @torch.no_grad()
    def forward(self, x, taps=None):
        """x: (N,3,H,W) fp32 (or fp16) in [-1,1] on the GPU -> (N,3,H,W) fp16.  ``taps`` (dict) receives the stage outputs."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f"EnhancedGenerator expects (N,3,H,W), got {tuple(x.shape)}")
        if x.shape[2] % 16 or x.shape[3] % 16:
            raise RuntimeError(f"EnhancedGenerator: H and W must be multiples of 16, got {x.shape[2]}x{x.shape[3]}")
        if not x.is_cuda:
            raise RuntimeError("mstg_hip: input must live on the GPU (this package has no CPU path)")
        x = x.float().contiguous()
        h, stats = self.stem(x, want_stats=True)   # pre-norm stem output; its IN + ReLU is applied by down1's conv on load
        # A stage's closing relu(IN(f)) + a is formed by the NEXT layer while it stages its input (same arithmetic, same bits as
        # f16_norm_residual_kernel; MSTG_F16_FOLD_RESIDUAL=0 keeps the pass).  The pass stays where somebody needs the tensor
        # itself: parity taps, and the stage in front of the transformer blocks.
        fold = taps is None and os.environ.get("MSTG_F16_FOLD_RESIDUAL", "1") != "0"
        res = None
        for si, (first, attn, branches, fusion) in enumerate(self.stages):
            c, st_c = first(h, in_stats=stats, want_stats=True, residual=res)
            a = attn(c, in_stats=st_c)              # IN + ReLU of the stage's first norm: on load
            cat, st_cat = branches(a, want_stats=True)
            f, st_f = fusion(cat, in_stats=st_cat, want_stats=True)
            if fold and not (si == 1 and self.blocks):
                h, stats, res = f, st_f, a          # the consumer forms relu(IN(f)) + a
                continue
            h = norm_residual(f, a, st_f)           # relu(IN(f)) + a
            stats = res = None
            if taps is not None:
                taps[("down1", "down2", "up1", "up2")[si]] = h
            if si == 1 and self.blocks:             # enhanced_generator.py:216-225: style vector, tokens, blocks, back
                hf = h.float()
                N, H4, W4, C4 = hf.shape
                style = self.style_vector(hf)
                tokens = hf.reshape(N, H4 * W4, C4)
                for block in self.blocks:
                    tokens = block(tokens, style, x)
                h = tokens.reshape(N, H4, W4, C4).half()
        if taps is not None:
            taps["pre_tanh"] = self.head_pre(h)[0]
        return self.head(h, in_stats=stats, residual=res)[0]
