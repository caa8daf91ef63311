"""BUILD-DEFINED multi-style perceptual loss -- parity unpinned (the reference has no VGG, no Gram matrix and no style
loss anywhere: SURVEY.md F2; "multi-style transfer with adjustable weights" is only a README bullet).

Definition (also restated for the CPU in oracle/restatement.py, which is what the tests compare against):
  * features: a frozen VGG16-topology stack up to relu4_3 -- 3x3 convolutions (pad 1) + ReLU, 2x2 max-pools after relu1_2,
    relu2_2, relu3_3 -- with taps after relu1_2, relu2_2, relu3_3, relu4_3.  torchvision and pretrained weights are not
    available offline, so the weights are seeded random (He-normal); throughput does not depend on their values;
  * Gram matrix of a tap: G = F F^T / (C H W);
  * style target per tap: sum_k w_k * mean_batch G(F(style_k)), sum_k w_k = 1 (three references by default);
  * loss: sum over taps of MSE(G(F(y)), target), added to the generator loss with weight ``lambda_style``.

Everything runs on the HIP kernels: the convolutions through the implicit-GEMM kernel (ReLU in its epilogue), max-pool
with arg-max bytes, Gram forward/backward as MFMA contractions.  Inputs are NCHW (N,3,H,W) in [-1,1] like the
generators' outputs; H and W must be multiples of 8.
"""
from __future__ import annotations

import math
from typing import List, Sequence

import torch
import torch.nn as nn

from mstg_hip import ops
from mstg_hip.layers import HipConv2d
from mstg_hip.ops import ACT_RELU

VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512]  # VGG16 up to relu4_3
VGG_TAPS = (1, 3, 6, 9)  # conv indices whose ReLU output is tapped


class VGGFeatures(nn.Module):
    """Frozen feature stack; state_dict keys ``conv{i}.weight`` / ``conv{i}.bias`` (i = 0..9)."""

    def __init__(self, width_div: int = 1, seed: int = 1234):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        cin, i = 3, 0
        self.plan = []
        for c in VGG_CFG:
            if c == "M":
                self.plan.append("M")
                continue
            co = c // width_div
            conv = HipConv2d(cin, co, 3, padding=1)
            with torch.no_grad():
                conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * math.sqrt(2.0 / (cin * 9)))
                conv.bias.zero_()
            self.add_module(f"conv{i}", conv)
            self.plan.append(i)
            cin, i = co, i + 1
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, x) -> List[torch.Tensor]:
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f"VGGFeatures expects (N,3,H,W), got {tuple(x.shape)}")
        if x.shape[2] % 8 or x.shape[3] % 8:
            raise RuntimeError("VGGFeatures: H and W must be multiples of 8 (three 2x2 max-pools)")
        feats, h, first = [], x, True
        for item in self.plan:
            if item == "M":
                h = ops.maxpool2x2(h)
                continue
            conv = getattr(self, f"conv{item}")
            h = conv(h, nhwc=True, x_nchw=first, act=ACT_RELU)  # NCHW image -> NHWC inside the first conv; ReLU fused
            first = False
            if item in VGG_TAPS:
                feats.append(h)
        return feats


class MultiStyleGramLoss(nn.Module):
    """sum_taps MSE(G(F(y)), sum_k w_k mean_batch G(F(style_k)))."""

    def __init__(self, features: VGGFeatures, styles: Sequence[torch.Tensor], weights: Sequence[float] = (0.5, 0.3, 0.2)):
        super().__init__()
        if len(styles) != len(weights):
            raise ValueError("one weight per style reference")
        if abs(sum(weights) - 1.0) > 1e-6:
            raise ValueError("style weights must sum to 1")
        self.features = features
        self.weights = tuple(float(w) for w in weights)
        with torch.no_grad():  # targets are computed once per set of style references
            per_style = [[ops.gram_matrix(f).mean(dim=0, keepdim=True) for f in features(s)] for s in styles]
            self.targets = [sum(w * gs[l] for w, gs in zip(self.weights, per_style)) for l in range(len(VGG_TAPS))]

    def forward(self, y):
        loss = None
        for f, t in zip(self.features(y), self.targets):
            g = ops.gram_matrix(f)
            term = ops.mse_loss(g, t.expand_as(g).contiguous())
            loss = term if loss is None else loss + term
        return loss
