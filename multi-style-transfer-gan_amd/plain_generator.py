"""The plain CycleGAN ``Generator`` of the reference (pretrain.py:60-97 == batch_process_images.py:20-58 ==
gan_login_gui.py:168-205 == pretrain_resume.py:60-97) on the MI355X kernels.

Same ``Generator(channels=64)`` constructor, ``.encoder`` / ``.decoder`` ``nn.Sequential`` members with the reference's
child indices (state_dict keys ``encoder.{0,2,5,8}.*``, BatchNorm at ``encoder.{3,6,9}`` / ``decoder.{1,4,7}``,
``decoder.{0,3,6,9}.*``) and NCHW ``forward``.  Under data parallelism BatchNorm statistics are per rank (replicas
only) -- the reference never trains this class on more than one device either.
"""
from __future__ import annotations

import torch.nn as nn

from mstg_hip.layers import HipBatchNorm2d, HipConv2d, HipConvTranspose2d, HipLeakyReLU, HipReLU, HipTanh
from mstg_hip.ops import ACT_LEAKY02, ACT_RELU, ACT_TANH
from mstg_hip import ops


class Generator(nn.Module):
    def __init__(self, channels=64):
        super().__init__()
        C = channels
        self.encoder = nn.Sequential(
            HipConv2d(3, C, 4, 2, 1), HipLeakyReLU(0.2),
            HipConv2d(C, C * 2, 4, 2, 1), HipBatchNorm2d(C * 2), HipLeakyReLU(0.2),
            HipConv2d(C * 2, C * 4, 4, 2, 1), HipBatchNorm2d(C * 4), HipLeakyReLU(0.2),
            HipConv2d(C * 4, C * 8, 4, 2, 1), HipBatchNorm2d(C * 8), HipLeakyReLU(0.2),
        )
        self.decoder = nn.Sequential(
            HipConvTranspose2d(C * 8, C * 4, 4, 2, 1), HipBatchNorm2d(C * 4), HipReLU(),
            HipConvTranspose2d(C * 4, C * 2, 4, 2, 1), HipBatchNorm2d(C * 2), HipReLU(),
            HipConvTranspose2d(C * 2, C, 4, 2, 1), HipBatchNorm2d(C), HipReLU(),
            HipConvTranspose2d(C, 3, 4, 2, 1), HipTanh(),
        )

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f"Generator expects (N,3,H,W), got {tuple(x.shape)}")
        if x.shape[2] % 16 or x.shape[3] % 16:
            raise RuntimeError(f"Generator: H and W must be multiples of 16 (four stride-2 stages), got {x.shape[2]}x{x.shape[3]}")
        e, d = self.encoder, self.decoder
        h = ops.activation(e[0](x, nhwc=True, x_nchw=True), ACT_LEAKY02)
        for ci, bi in ((2, 3), (5, 6), (8, 9)):
            h = e[bi](e[ci](h, nhwc=True), nhwc=True, act=ACT_LEAKY02)
        for ci, bi in ((0, 1), (3, 4), (6, 7)):
            h = d[bi](d[ci](h, nhwc=True), nhwc=True, act=ACT_RELU)
        return d[9](h, nhwc=True, y_nchw=True, act=ACT_TANH)
