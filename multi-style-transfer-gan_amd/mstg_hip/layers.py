"""HIP-backed leaf layers.

Each class subclasses the torch.nn layer it replaces, so parameter names, shapes, default initialisation (same RNG
draws as the reference for the same seed), ``isinstance`` checks (the reference's ``_init_weights`` and its
``spectral_norm`` loop, enhanced_generator.py:152-161,269-271) and state_dict keys are inherited unchanged --
only ``forward`` is replaced by the gfx950 kernels.  ``forward(x)`` keeps the reference's NCHW contract; the fused
model paths call the layers with ``nhwc=True`` and stay in NHWC between ops.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_LEAKY02, ACT_NONE, ACT_RELU, ACT_TANH


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 2, 3, 1).contiguous()


def to_nchw(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2).contiguous()


def _single(v):
    return v[0] if isinstance(v, (tuple, list)) else v


class HipConv2d(nn.Conv2d):
    """nn.Conv2d on the implicit-GEMM MFMA kernel.  Square kernels, groups=1, zero padding."""

    def forward(self, x, nhwc: bool = False, x_nchw: bool = False, y_nchw: bool = False, act: int = ACT_NONE):
        if self.groups != 1 or self.padding_mode != "zeros":
            raise RuntimeError("HipConv2d: only groups=1, zero padding")
        k, s, p, d = _single(self.kernel_size), _single(self.stride), _single(self.padding), _single(self.dilation)
        if not nhwc:  # reference contract: NCHW in, NCHW out
            if x.dim() != 4:
                raise RuntimeError(f"HipConv2d expects a 4-D NCHW tensor, got {tuple(x.shape)}")
            if x.shape[1] <= 4:
                return to_nchw(ops.conv2d(x, self.weight, self.bias, k, s, p, d, x_nchw=True, act=act))
            return to_nchw(ops.conv2d(to_nhwc(x), self.weight, self.bias, k, s, p, d, act=act))
        return ops.conv2d(x, self.weight, self.bias, k, s, p, d, x_nchw=x_nchw, y_nchw=y_nchw, act=act)


class HipConvTranspose2d(nn.ConvTranspose2d):
    """nn.ConvTranspose2d(k=4, s=2, p=1) as four parity-class 2x2 convolutions (exact 2x upsampling)."""

    def forward(self, x, nhwc: bool = False, y_nchw: bool = False, act: int = ACT_NONE):
        k, s, p, d = _single(self.kernel_size), _single(self.stride), _single(self.padding), _single(self.dilation)
        if (k, s, p, d) != (4, 2, 1, 1) or _single(self.output_padding) != 0 or self.groups != 1:
            raise RuntimeError("HipConvTranspose2d: only kernel 4, stride 2, padding 1 is implemented")
        if not nhwc:
            return to_nchw(ops.conv2d(to_nhwc(x), self.weight, self.bias, k, s, p, d, transposed=True, act=act))
        return ops.conv2d(x, self.weight, self.bias, k, s, p, d, transposed=True, y_nchw=y_nchw, act=act)


class HipInstanceNorm2d(nn.InstanceNorm2d):
    """nn.InstanceNorm2d(affine=False, track_running_stats=False); fused with an activation via ``act``."""

    def forward(self, x, nhwc: bool = False, act: int = ACT_NONE, residual=None):
        if self.affine or self.track_running_stats:
            raise RuntimeError("HipInstanceNorm2d: affine / running-stat variants are not on the reference path")
        if not nhwc:
            return to_nchw(ops.instnorm_act(to_nhwc(x), act))
        return ops.instnorm_act(x, act, residual)


class HipBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (affine, momentum 0.1); fused with an activation via ``act``."""

    def forward(self, x, nhwc: bool = False, act: int = ACT_NONE):
        if not self.affine or not self.track_running_stats or self.momentum != 0.1:
            raise RuntimeError("HipBatchNorm2d: only the default affine / momentum 0.1 configuration")
        if self.training:
            self.num_batches_tracked += 1
        xin = x if nhwc else to_nhwc(x)
        y = ops.BatchNormActFn.apply(xin, self.weight, self.bias, self.running_mean, self.running_var, act, self.training)
        return y if nhwc else to_nchw(y)


class HipLeakyReLU(nn.LeakyReLU):
    def forward(self, x):
        if self.negative_slope != 0.2:
            raise RuntimeError("HipLeakyReLU: slope 0.2 only")
        return ops.activation(x, ACT_LEAKY02)


class HipReLU(nn.ReLU):
    def forward(self, x):
        return ops.activation(x, ACT_RELU)


class HipTanh(nn.Tanh):
    def forward(self, x):
        return ops.activation(x, ACT_TANH)
