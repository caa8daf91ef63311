"""Drop-in for the reference's ``enhanced_generator`` module, running on hand-written MI355X (gfx950) kernels.

Put this directory on ``PYTHONPATH`` and the reference's callers (``advanced_transform.py:8``,
``batch_process_images.py:18``, ``direct_transform.py``, ``enhanced_train.py:10`` ...) import these classes instead
of the torch.nn originals: same class names, constructor signatures, ``forward`` contracts (NCHW fp32 in/out),
``gradient_checkpointing_enable()`` and state_dict keys/shapes, so reference ``.pth``
files load unchanged.

What is different is everything below ``forward``: activations stay NHWC between ops, InstanceNorm+ReLU(+residual)
is one fused kernel pair, the four MultiScaleBlock branches write straight into the concatenated buffer, the
window partition/un-partition permutes of LocalAttention are index arithmetic inside the attention kernel, Tanh is
the head convolution's epilogue, and the 3-channel NCHW<->NHWC conversions are folded into the stem / head
convolutions.  There is no CPU or eager-PyTorch path: tensors must be on the GPU and ``libmstg_hip.so`` must be
built, otherwise the ops raise.

Mirrors /root/reference/enhanced_generator.py: LocalAttention :6-47, MultiScaleBlock :49-84,
EnhancedGenerator :86-228, EnhancedDiscriminator :230-274.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.utils.checkpoint

from mstg_hip import ops
from mstg_hip.layers import (HipConv2d, HipConvTranspose2d, HipInstanceNorm2d, HipLeakyReLU, HipReLU, HipTanh, to_nchw,
                             to_nhwc)
from mstg_hip.ops import ACT_LEAKY02, ACT_NONE, ACT_RELU, ACT_TANH
from structural_transformer import StructuralTransformerBlock


class LocalAttention(nn.Module):
    """Windowed channel attention (reference :6-47).  ``forward`` takes/returns NCHW like the reference."""

    def __init__(self, channels, window_size=8):
        super().__init__()
        self.window_size = window_size
        self.qkv = HipConv2d(channels, channels * 3, 1)
        self.proj = HipConv2d(channels, channels, 1)

    def forward_nhwc(self, x):
        ws = self.window_size
        H, W = x.shape[1], x.shape[2]
        if H % ws or W % ws:  # the reference's padding branch is broken and raises RuntimeError from .view (:15-23)
            raise RuntimeError(f"LocalAttention: H and W must be multiples of window_size={ws}, got {H}x{W}")
        if ws != 4:  # the constructor default (8) and any other size: qkv conv -> general-window core -> proj conv
            if not ops._lib.load().mstg_window_attn_ws_supported(int(x.shape[3]), int(ws)):
                raise RuntimeError(f"LocalAttention: window_size={ws} with {x.shape[3]} channels does not fit the general-window kernel "
                                   f"(a whole window is kept on chip); the reference's callers all use window_size=4")
            return self.proj(ops.WindowAttnCoreWsFn.apply(self.qkv(x, nhwc=True), ws), nhwc=True)
        if ops.fused_attention_supported(x.shape[3]) and os.environ.get("MSTG_ATTN_UNFUSED") != "1":
            # qkv conv + window attention + proj conv in one kernel: x read once, y written once
            return ops.LocalAttentionFusedFn.apply(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias)
        qkv = self.qkv(x, nhwc=True)
        o = ops.WindowAttnCoreFn.apply(qkv)
        return self.proj(o, nhwc=True)

    def forward(self, x):
        return to_nchw(self.forward_nhwc(to_nhwc(x)))


class MultiScaleBlock(nn.Module):
    """1x1 + three dilated 3x3 branches, IN+ReLU each, concat, 1x1 fusion + IN + ReLU, residual (reference :49-84)."""

    def __init__(self, channels):
        super().__init__()
        c4 = channels // 4
        self.branch1 = nn.Sequential(HipConv2d(channels, c4, 1), HipInstanceNorm2d(c4), HipReLU(True))
        self.branch2 = nn.Sequential(HipConv2d(channels, c4, 3, padding=1, dilation=1), HipInstanceNorm2d(c4), HipReLU(True))
        self.branch3 = nn.Sequential(HipConv2d(channels, c4, 3, padding=2, dilation=2), HipInstanceNorm2d(c4), HipReLU(True))
        self.branch4 = nn.Sequential(HipConv2d(channels, c4, 3, padding=4, dilation=4), HipInstanceNorm2d(c4), HipReLU(True))
        self.fusion = nn.Sequential(HipConv2d(channels, channels, 1), HipInstanceNorm2d(channels), HipReLU(True))

    def forward_nhwc(self, x):
        wb = []
        for br in (self.branch1, self.branch2, self.branch3, self.branch4):
            wb += [br[0].weight, br[0].bias]
        cat, xres = ops.MSBranchesFn.apply(x, *wb)    # 4 convs -> one (N,H,W,ch) buffer, no torch.cat; xres = x for the residual
        conv = self.fusion[0]
        if (os.environ.get("MSTG_NORM_FUSION", "1") != "0" and conv.kernel_size == (1, 1) and conv.out_channels == cat.shape[3]
                and ops.ms_fusion_supported(cat.shape[0], cat.shape[1], cat.shape[2], cat.shape[3])):
            # the concat's IN + ReLU folded into the fusion convolution: the normalised concat is never written
            # ... and the fusion output's statistics come out of the convolution's epilogue: its norm is the apply pass alone
            f, fstats = ops.MSFusionFn.apply(cat, conv.weight, conv.bias)
            return ops.instnorm_apply(f, fstats, ACT_RELU, residual=xres)
        else:
            cat = ops.instnorm_act(cat, ACT_RELU)      # per-channel IN: one launch covers all four branches
            f = conv(cat, nhwc=True)
        return ops.instnorm_act(f, ACT_RELU, residual=xres)

    def forward(self, x):
        return to_nchw(self.forward_nhwc(to_nhwc(x)))


class _Stage(nn.Sequential):
    """conv(T) -> IN -> ReLU -> LocalAttention -> MultiScaleBlock with the reference's child indices 0..4."""

    def forward_nhwc(self, x, x_raw=False):
        """x_raw: x is the RAW tensor in front of an InstanceNorm + ReLU still to be applied (the stem's, handed down by
        EnhancedGenerator): folded into this stage's convolution where the kernels can normalise on load, applied here otherwise."""
        conv, att = self[0], self[3]
        transposed = isinstance(conv, nn.ConvTranspose2d)
        cout = conv.out_channels
        pre = None
        if x_raw:
            if (not transposed and conv.kernel_size == (4, 4) and conv.stride == (2, 2) and conv.padding == (1, 1)
                    and os.environ.get("MSTG_NORM_STEM", "1") != "0"
                    and ops.norm_conv_supported(x.shape[0], x.shape[1], x.shape[2], x.shape[3], cout, 4, 2, 1, 1)):
                pre = ops.MSFusionFn.apply(x, conv.weight, conv.bias, (4, 2, 1, 1))  # (conv output, its statistics)
            else:
                x = ops.instnorm_act(x, ACT_RELU)
        fold = (att.window_size == 4 and ops.fused_attention_supported(cout) and os.environ.get("MSTG_ATTN_UNFUSED") != "1"
                and os.environ.get("MSTG_NORM_ATTN", "1") != "0")
        stats = None
        if pre is not None:
            x, stats = pre
            if not (fold and x.shape[1] % 4 == 0 and x.shape[2] % 4 == 0) or os.environ.get("MSTG_NORM_EPILOGUE", "1") == "0":
                stats = None
        elif (fold and os.environ.get("MSTG_NORM_EPILOGUE", "1") != "0" and conv.kernel_size == (4, 4) and conv.stride == (2, 2)
                and conv.padding == (1, 1) and ops.conv_stats_pays(x.shape[0], x.shape[1], x.shape[2], x.shape[3], cout, 4, 2, 1, 1,
                                                                   transposed)):
            # the norm's statistics come out of the convolution's epilogue: no pass over the tensor for them
            x, stats = ops.conv2d_stats(x, conv.weight, conv.bias, 4, 2, 1, 1, transposed=transposed)
        else:
            x = conv(x, nhwc=True)
        if fold and x.shape[1] % 4 == 0 and x.shape[2] % 4 == 0:
            # IN + ReLU folded into the attention kernels (its only consumer): the normalised tensor is never written
            x = ops.NormLocalAttentionFn.apply(x, att.qkv.weight, att.qkv.bias, att.proj.weight, att.proj.bias, stats)
        else:
            x = ops.instnorm_act(x, ACT_RELU)
            x = att.forward_nhwc(x)
        return self[4].forward_nhwc(x)

    def forward(self, x):
        return to_nchw(self.forward_nhwc(to_nhwc(x)))


class EnhancedGenerator(nn.Module):
    def __init__(self, channels=64, num_transformer_blocks=3):
        super().__init__()
        C = channels
        self.initial = nn.Sequential(HipConv2d(3, C, 7, 1, 3), HipInstanceNorm2d(C), HipReLU(True))
        self.down1 = _Stage(HipConv2d(C, C * 2, 4, 2, 1), HipInstanceNorm2d(C * 2), HipReLU(True),
                            LocalAttention(C * 2, window_size=4), MultiScaleBlock(C * 2))
        self.down2 = _Stage(HipConv2d(C * 2, C * 4, 4, 2, 1), HipInstanceNorm2d(C * 4), HipReLU(True),
                            LocalAttention(C * 4, window_size=4), MultiScaleBlock(C * 4))
        self.transformer_blocks = nn.ModuleList([StructuralTransformerBlock(dim=C * 4) for _ in range(num_transformer_blocks)])
        self.up1 = _Stage(HipConvTranspose2d(C * 4, C * 2, 4, 2, 1), HipInstanceNorm2d(C * 2), HipReLU(True),
                          LocalAttention(C * 2, window_size=4), MultiScaleBlock(C * 2))
        self.up2 = _Stage(HipConvTranspose2d(C * 2, C, 4, 2, 1), HipInstanceNorm2d(C), HipReLU(True),
                          LocalAttention(C, window_size=4), MultiScaleBlock(C))
        self.output = nn.Sequential(HipConv2d(C, 3, 7, 1, 3), HipTanh())
        self.style_encoder = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(C * 4, C * 4), nn.ReLU(True))
        self.apply(self._init_weights)

    def _init_weights(self, m):  # reference :152-161
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.InstanceNorm2d):
            if m.weight is not None:
                nn.init.constant_(m.weight, 1)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def gradient_checkpointing_enable(self):
        """Recompute each stage in the backward (reference :163-177).  Results are identical either way; with 288 GB of
        HBM it is off unless asked for."""
        for m in (self.down1, self.down2, self.transformer_blocks, self.up1, self.up2):
            m.requires_grad_(True)
        self.use_checkpointing = True

    def half_inference(self, enable: bool = True):
        """Inference-only fast path (BASELINE config #5): fp16 storage, fp16 MFMA, fp32 accumulation / statistics / softmax.
        Under ``torch.no_grad()`` ``forward`` then runs csrc/infer_f16.hip (mstg_hip/infer.py) and returns an fp16 (N,3,H,W)
        tensor; with autograd enabled the fp32 training path still runs.  The packed fp16 filters are rebuilt lazily after a
        ``load_state_dict``; call ``half_inference()`` again after changing weights in any other way."""
        if enable:  # fail here, not at the first forward: the fp16 kernels are built for the deployed width
            C0 = self.initial[0].out_channels
            if C0 != 16:
                raise RuntimeError(f"mstg_hip fp16 inference is built for channels=16 (stage widths 16/32/64, what every trainer and "
                                   f"inference caller of the reference uses), got channels={C0}; the fp32 forward serves other widths")
        self._half_enabled = bool(enable)
        self._half_plan = None
        if enable and not getattr(self, "_half_hook", False):
            self.register_load_state_dict_post_hook(lambda module, incompatible: setattr(module, "_half_plan", None))
            self._half_hook = True
        return self

    def graph_inference(self, enable: bool = True):
        """Replay the inference forward (under ``torch.no_grad()``) from a captured hipGraph, one per input shape: a forward is
        60-70 dependent launches (23 on the fp16 path) whose launch gaps dominate at batch 1 (BASELINE config #1).  The graph
        reads the parameters in place, so weight updates are picked up; it is re-captured after a ``load_state_dict``.  The
        returned tensor is a copy of the graph's output buffer."""
        self._graph_enabled = bool(enable)
        self._graphs = {}
        if enable and not getattr(self, "_graph_hook", False):
            self.register_load_state_dict_post_hook(lambda module, incompatible: setattr(module, "_graphs", {}))
            self._graph_hook = True
        return self

    def _graph_forward(self, x):
        key = (tuple(x.shape), x.dtype, bool(getattr(self, "_half_enabled", False)))
        entry = self._graphs.get(key)
        if entry is None:
            static_x = x.clone()
            for _ in range(2):  # warm-up outside the capture: lazy one-time set-up (kernel attributes, occupancy queries, fp16 plan)
                self._forward_impl(static_x, None)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_y = self._forward_impl(static_x, None)
            entry = self._graphs[key] = (graph, static_x, static_y)
        graph, static_x, static_y = entry
        static_x.copy_(x)
        graph.replay()
        return static_y.clone()

    def _half(self):
        if getattr(self, "_half_plan", None) is None:
            from mstg_hip.infer import HalfGeneratorPlan
            self._half_plan = HalfGeneratorPlan(self)
        return self._half_plan

    def _run(self, fn, *args):
        if getattr(self, "use_checkpointing", False) and torch.is_grad_enabled():
            return torch.utils.checkpoint.checkpoint(fn, *args, use_reentrant=False)
        return fn(*args)

    def _style_vector(self, feat_nhwc):
        pooled = ops.spatial_mean(feat_nhwc)  # AdaptiveAvgPool2d(1) + Flatten
        lin = self.style_encoder[2]        # Linear + ReLU (:144-146) as a 1x1 convolution over N "pixels", ReLU in its epilogue
        return ops.linear_tokens(pooled, lin.weight, lin.bias, act=ACT_RELU)

    def forward_taps(self, x, taps=None):
        """forward() that optionally records stage outputs (NHWC) into ``taps`` -- used by the parity tests."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f"EnhancedGenerator expects (N,3,H,W), got {tuple(x.shape)}")
        if x.shape[2] % 16 or x.shape[3] % 16:
            raise RuntimeError(f"EnhancedGenerator: H and W must be multiples of 16 (two stride-2 stages and 4x4 windows), "
                               f"got {x.shape[2]}x{x.shape[3]}")
        if getattr(self, "_graph_enabled", False) and taps is None and not torch.is_grad_enabled() and x.is_cuda:
            return self._graph_forward(x.contiguous())
        return self._forward_impl(x, taps)

    def _forward_impl(self, x, taps):
        if getattr(self, "_half_enabled", False) and not torch.is_grad_enabled():
            return self._half().forward(x, taps)
        orig_input = x
        h = self.initial[0](x, nhwc=True, x_nchw=True)          # NCHW image -> NHWC features inside the stem conv
        if taps is None:
            # the stem's IN + ReLU has one consumer, down1's convolution: handed down raw, the normalised tensor is never written
            h = self._run(self.down1.forward_nhwc, h, True)
        else:
            h = ops.instnorm_act(h, ACT_RELU)
            taps["initial"] = h
            h = self._run(self.down1.forward_nhwc, h)
        if taps is not None: taps["down1"] = h
        h = self._run(self.down2.forward_nhwc, h)
        if taps is not None: taps["down2"] = h
        if len(self.transformer_blocks) > 0:
            all_identity = all(getattr(b, "is_identity", False) for b in self.transformer_blocks)
            style = None if all_identity else self._style_vector(h)
            N, H4, W4, C4 = h.shape
            tokens = h.reshape(N, H4 * W4, C4)                    # NHWC is already (B, HW, C) token-major (:218-219)
            for block in self.transformer_blocks:
                tokens = self._run(block, tokens, style, orig_input)
            h = tokens.reshape(N, H4, W4, C4)
        h = self._run(self.up1.forward_nhwc, h)
        if taps is not None: taps["up1"] = h
        h = self._run(self.up2.forward_nhwc, h)
        if taps is not None: taps["up2"] = h
        if taps is not None:
            taps["pre_tanh"] = self.output[0](h, nhwc=True, y_nchw=True)
        return self.output[0](h, nhwc=True, y_nchw=True, act=ACT_TANH)  # tanh fused, NHWC -> NCHW inside the head conv

    def forward(self, x):
        return self.forward_taps(x, None)


class EnhancedDiscriminator(nn.Module):
    def __init__(self, channels=64):
        super().__init__()
        C = channels
        self.main = nn.Sequential(
            HipConv2d(3, C, 4, 2, 1), HipLeakyReLU(0.2),
            HipConv2d(C, C * 2, 4, 2, 1), HipInstanceNorm2d(C * 2), HipLeakyReLU(0.2),
            HipConv2d(C * 2, C * 4, 4, 2, 1), HipInstanceNorm2d(C * 4), HipLeakyReLU(0.2),
            HipConv2d(C * 4, C * 8, 4, 2, 1), HipInstanceNorm2d(C * 8), HipLeakyReLU(0.2),
        )
        self.batch_head = nn.Sequential(HipConv2d(C * 8, 1, 4, 1, 1), nn.AdaptiveAvgPool2d(1))
        self.structure_head = nn.Sequential(HipConv2d(C * 8, C * 8, 3, 1, 1), HipInstanceNorm2d(C * 8), HipLeakyReLU(0.2),
                                            HipConv2d(C * 8, 1, 4, 1, 1))
        fused = []
        for m in self.modules():  # reference :269-271 -- the hook recomputes W/sigma (one power iteration) per forward
            if isinstance(m, nn.Conv2d):
                nn.utils.spectral_norm(m)
                if os.environ.get("MSTG_TORCH_SPECTRAL_NORM", "0") != "1":
                    fused.append(ops.install_fused_spectral_norm(m))  # same parameters / buffers / state_dict; one launch instead of ~14
        # all seven weights of a forward in one grouped call (three launches) ahead of the first convolution; MSTG_SN_GROUP=0 leaves
        # every convolution to its own hook
        self._sn_grouped = bool(fused) and all(fused)

    def _sn_convs(self):
        return [self.main[0], self.main[2], self.main[5], self.main[8], self.batch_head[0], self.structure_head[0], self.structure_head[3]]

    def forward(self, x, outputs="both"):
        """(score, structure) like the reference (:256-274).  ``outputs`` = "score" / "struct" is a hint from the train step about the
        head whose result it discards; it is honoured only with MSTG_D_SKIP_DEAD_HEADS=1 (that head then comes back as None; every
        weight still takes its power iteration for this forward -- the grouped spectral norm normalises all seven -- so the other head
        and the state are exactly what a full forward gives)."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f"EnhancedDiscriminator expects (N,3,H,W), got {tuple(x.shape)}")
        grouped = self._sn_grouped and os.environ.get("MSTG_SN_GROUP", "1") != "0"
        if grouped:
            ops.spectral_norm_group(self._sn_convs())
        if not grouped or outputs not in ("score", "struct") or os.environ.get("MSTG_D_SKIP_DEAD_HEADS", "0") != "1":
            # default: both heads run, as in the reference, even where the train step discards one (the bench times the reference's
            # work); opt-in MSTG_D_SKIP_DEAD_HEADS=1 drops the discarded head (-38 launches, -0.4 ms per step, same losses / gradients /
            # state).  With per-module hooks a skipped convolution would also skip its power iteration: never skipped then.
            outputs = "both"
        m = self.main
        h = m[0](x, nhwc=True, x_nchw=True, act=ACT_LEAKY02)   # LeakyReLU in the convolution's epilogue (same arithmetic, one pass less)
        for ci in (2, 5, 8):
            h = ops.instnorm_act(m[ci](h, nhwc=True), ACT_LEAKY02)
        N = h.shape[0]
        score = st = None
        if outputs != "struct":
            score = ops.spatial_mean(self.batch_head[0](h, nhwc=True)).view(N, 1, 1, 1).squeeze()   # (N, 1)
        else:
            self.batch_head[0].__dict__.pop("_mstg_sn_fresh", None)  # its hook does not run: drop the marker the group call left
        if outputs != "score":
            s = ops.instnorm_act(self.structure_head[0](h, nhwc=True), ACT_LEAKY02)
            st = self.structure_head[3](s, nhwc=True).permute(0, 3, 1, 2)       # (N, h, w, 1) == NCHW (N, 1, h, w)
        else:
            for mod in (self.structure_head[0], self.structure_head[3]):
                mod.__dict__.pop("_mstg_sn_fresh", None)
        return score, st
