"""torch.autograd bindings of the HIP kernels.

PyTorch is used for what it is good at here -- device memory (caching allocator), streams and the autograd
graph -- while every arithmetic op of the hot path is a hand-written gfx950 kernel reached through the C ABI
of ``libmstg_hip.so`` (include/mstg_hip.h).  Activations between ops are fp32 NHWC-contiguous tensors of shape
(N, H, W, C); only the 3-channel image tensors at the module boundary are NCHW.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import os

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_LEAKY02, ACT_NONE, ACT_RELU, ACT_TANH, LOSS_L1, LOSS_MSE, ConvDesc  # noqa: F401

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------------------
# plumbing
# ----------------------------------------------------------------------------------------------------------
def refresh_env() -> None:
    """Make the library re-read the MSTG_* switches (it reads them once, at load).  Python-side switches are read per call."""
    _lib.load().mstg_env_refresh()
    bump_pack_epoch()  # a switch may select another kernel, and with it another packed-filter format


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _req(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"mstg_hip: {name} must live on the GPU (this package has no CPU path)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"mstg_hip: {name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ws(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 16) // 4 + 1, dtype=torch.float32, device=device)


def conv_out_hw(H, W, k, stride, pad, dil, transposed):
    if transposed:
        return (H - 1) * stride - 2 * pad + dil * (k - 1) + 1, (W - 1) * stride - 2 * pad + dil * (k - 1) + 1
    return (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1


def make_desc(N, H, W, Cin, Ho, Wo, Cout, k, stride, pad, dil, transposed=0, x_nchw=0, y_nchw=0, x_ctot=None, x_coff=0,
              y_ctot=None, y_coff=0, act=ACT_NONE, accumulate=0) -> ConvDesc:
    return ConvDesc(N, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad, dil, int(transposed), int(x_nchw), int(y_nchw),
                    Cin if x_ctot is None else x_ctot, x_coff, Cout if y_ctot is None else y_ctot, y_coff, act, int(accumulate))


# ----------------------------------------------------------------------------------------------------------
# Parameter gradients straight into p.grad
# ----------------------------------------------------------------------------------------------------------
# With every p.grad a persistent view of an optimizer's flat gradient buffer (optim.FlatAdam), the reduce kernels can add a
# parameter's gradient to that view themselves; backward() then returns None for the parameter and autograd launches no
# `grad += new` kernel (about 350 of them per CycleGAN step).  Opt-in (EnhancedCycleGAN.train_step): torch.autograd.grad /
# hooks on parameters do not see gradients delivered this way.
_DIRECT_PARAM_GRADS = False


class direct_param_grads:
    def __enter__(self):
        global _DIRECT_PARAM_GRADS
        self.prev, _DIRECT_PARAM_GRADS = _DIRECT_PARAM_GRADS, True

    def __exit__(self, *exc):
        global _DIRECT_PARAM_GRADS
        _DIRECT_PARAM_GRADS = self.prev
        return False


def _grad_slot(p):
    """p.grad if gradients of the parameter `p` can be accumulated in place right now, else None."""
    if not _DIRECT_PARAM_GRADS or not isinstance(p, torch.nn.Parameter) or p.grad is None or not p.grad.is_contiguous():
        return None
    return p.grad


class KernelTimer:
    """Opt-in per-KERNEL timing for bench.py's roofline figure.  The library brackets every kernel it launches with two HIP events on
    the launch stream (mstg_prof_*, csrc/runtime.hip) and reports the symbol rocprofv3 would print; this class adds the
    ALGORITHMIC work of each C-ABI call (multiply-add = 2 FLOP, no recompute, no padding; every tensor read once and
    written once) and books it on the kernel that does the call's arithmetic.  Helper launches of a call (filter packing, slab
    reduction, statistics finalize) and launches outside any `_timed` call appear under their own symbols with zero algorithmic
    work: their time is overhead and is shown as such."""
    enabled = False
    calls = []  # (label, first record, one-past-last record, flops, bytes, detail, split)

    @classmethod
    def start(cls):
        _lib.check(_lib.load().mstg_prof_enable(1), "mstg_prof_enable")
        cls.enabled, cls.calls = True, []

    @classmethod
    def stop(cls):
        cls.enabled = False
        _lib.load().mstg_prof_enable(0)

    @classmethod
    def kernels(cls):
        """[(symbol, milliseconds)] of every launch since start(), in launch order (waits for the events)."""
        lib = _lib.load()
        out, buf, ms = [], C.create_string_buffer(512), C.c_float()
        for i in range(lib.mstg_prof_count()):
            _lib.check(lib.mstg_prof_get(i, buf, 512, C.byref(ms)), "mstg_prof_get")
            out.append((buf.value.decode(), float(ms.value)))
        return out

    @staticmethod
    def _owner(label, names_ms):
        """Index (within a call's launches) of the kernel the call's algorithmic work belongs to: the launch whose symbol starts
        with the label's kernel name, else the longest launch."""
        stem = label.split("<")[0]
        hits = [k for k, (nm, _) in enumerate(names_ms) if nm.split("<")[0] == stem]
        if hits:
            return max(hits, key=lambda k: names_ms[k][1])
        return max(range(len(names_ms)), key=lambda k: names_ms[k][1])

    @classmethod
    def summary(cls, detail=False, by_call=False):
        """{kernel symbol: dict(launches, ms, flops, bytes)} (by_call=True: {call label: dict(calls, ms, flops, bytes, kernels)})."""
        ks = cls.kernels()
        out = {}

        def row(key):
            return out.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        covered = [False] * len(ks)
        for label, c0, c1, fl, by, det, split in cls.calls:
            mine = ks[c0:c1]
            if not mine:
                continue
            for k in range(c0, c1):
                covered[k] = True
            if by_call:
                r = row(f"{label} {det}" if detail else label)
                r["launches"] += 1
                r["ms"] += sum(m for _, m in mine)
                r["flops"] += fl
                r["bytes"] += by
                r.setdefault("kernels", set()).update(nm for nm, _ in mine)
                continue
            shares = {}
            if split:
                for k, (nm, _) in enumerate(mine):
                    for stem, (sfl, sby) in split.items():
                        if nm.split("<")[0] == stem.split("<")[0] and (stem.find("<") < 0 or nm.startswith(stem)):
                            shares[k] = (sfl, sby)
            if not shares:
                shares[cls._owner(label, mine)] = (fl, by)
            for k, (nm, m) in enumerate(mine):
                r = row(f"{nm} {det}" if detail else nm)
                r["launches"] += 1
                r["ms"] += m
                sfl, sby = shares.get(k, (0.0, 0.0))
                r["flops"] += sfl
                r["bytes"] += sby
        for k, (nm, m) in enumerate(ks):
            if not covered[k]:
                r = row(nm)
                r["launches"] += 1
                r["ms"] += m
        return out


def _timed(sym, flops, nbytes, fn, detail="", split=None):
    """Run one C-ABI call; with the timer on, remember which of the library's launch records it produced and its algorithmic work.
    split = {kernel symbol stem: (flops, bytes)} books the work on several kernels of the call (two-kernel passes)."""
    if not KernelTimer.enabled:
        return fn()
    lib = _lib.load()
    c0 = lib.mstg_prof_count()
    fn()
    KernelTimer.calls.append((sym, c0, lib.mstg_prof_count(), float(flops), float(nbytes), detail, split))


def _conv_detail(tag, d: ConvDesc):
    return (f"{tag} N{d.N} {d.H}x{d.W} {d.Cin}->{d.Cout} k{d.KH} s{d.stride} d{d.dil}" + (" T" if d.transposed else "")
            + (f" ycoff{d.y_coff}/{d.y_ctot}" if d.y_ctot != d.Cout else ""))


def _conv_cost(d: ConvDesc):
    """Algorithmic work of one convolution pass: 2*MACs, and read-input-once + write-output-once + weights-once bytes."""
    T = d.KH * d.KW
    macs = d.N * (d.H * d.W if d.transposed else d.Ho * d.Wo) * d.Cin * d.Cout * T
    nbytes = 4 * (d.N * d.H * d.W * d.Cin + d.N * d.Ho * d.Wo * d.Cout + d.Cin * d.Cout * T)
    return 2 * macs, nbytes


def _kernel_name(d: ConvDesc, which: int) -> str:
    return _lib.load().mstg_conv2d_kernel_name(C.byref(d), which).decode() if KernelTimer.enabled else ""


# ----------------------------------------------------------------------------------------------------------
# Filter packs kept between launches
# ----------------------------------------------------------------------------------------------------------
# Every convolution launch of the library starts by re-packing its filter into the workspace (a ~5 us kernel; ~170 of them per
# CycleGAN step, where each generator runs three forwards and three backwards on unchanged weights).  When the layer OWNS its
# filter (the tensors are nn.Parameters) the workspace is kept on the parameter object and the *_cached entry points skip the pack
# while the stamp below is unchanged.  What can change a parameter's values, and how the stamp sees it:
#   * torch in-place ops on the parameter (load_state_dict, p.copy_, ...)         -> p._version
#   * torch in-place ops on the optimizer's flat buffer it is a view of             -> flat._version (FlatAdam sets p._mstg_flat)
#   * raw-pointer writers: the fused Adam step, the DP broadcast of the flat buffer -> PACK_EPOCH, bumped by those call sites
#   * p.data re-homed (module.to / .half / a new FlatAdam)                          -> data_ptr
# The cache dies with the parameter object (no address-reuse hazard), is per stream (the two halves of the train step never share
# a workspace) and is never used for tensors that are not Parameters (the discriminator's W / sigma temporaries).
# MSTG_NO_PACK_CACHE=1 switches it off.
PACK_EPOCH = [0]


def bump_pack_epoch() -> None:
    """Call after writing parameter memory behind torch's back (a kernel handed raw pointers)."""
    PACK_EPOCH[0] += 1


def _pack_stamp(owners):
    return tuple(None if t is None else (id(t), t.data_ptr(), t._version, getattr(getattr(t, "_mstg_flat", None), "_version", -1))
                 for t in owners) + (PACK_EPOCH[0],)


def _cached_ws(owners, kind, dkey, nbytes: int, device):
    """(workspace, 1 if it still holds this call's filter pack else 0)"""
    if (not owners or owners[0] is None or os.environ.get("MSTG_NO_PACK_CACHE", "0") == "1"
            or not all(t is None or isinstance(t, torch.nn.Parameter) for t in owners)):
        return _ws(nbytes, device), 0
    packs = owners[0].__dict__.setdefault("_mstg_packs", {})
    key, stream = (kind, dkey), torch.cuda.current_stream().cuda_stream
    stamp = _pack_stamp(owners)
    ent = packs.get(key)
    if ent is not None and ent[2] == stream and ent[0].numel() * 4 >= nbytes:
        if ent[1] == stamp:
            return ent[0], 1
        packs[key] = (ent[0], stamp, stream)  # same stream: the re-pack is ordered behind every launch that still reads the old one
        return ent[0], 0
    # first use, or a use from ANOTHER stream than the last one (launches of the other stream may still read the old workspace:
    # it is left alone and dropped; in steady state a layer always runs on the same stream)
    ws = _ws(nbytes, device)
    packs[key] = (ws, stamp, stream)
    return ws, 0


def _dkey(d: ConvDesc, pass_: int):
    """What a filter pack depends on.  The persistent kernels' packs (conv_p32.hip: p32_plan) are a function of the filter geometry
    and channel chunking only, so the batch size is left out of the key there -- a generator's batched pass (2N images) and its
    reconstruction pass (N images) share one pack per step; other kernels plan by tile count, their key keeps every field."""
    fields = tuple(getattr(d, n) for n, _ in ConvDesc._fields_)
    name = _lib.load().mstg_conv2d_kernel_name(C.byref(d), pass_)
    return (name,) + (fields[1:] if name.startswith(b"conv_p32") else fields)


def conv_fwd_raw(d: ConvDesc, x, w, b, y, owners=None):
    """owners: the (weight, bias) objects of the calling layer when it owns them (nn.Parameters) -> their pack is kept."""
    fl, by = _conv_cost(d)
    lib = _lib.load()
    ws, packed = _cached_ws(owners, "fwd", _dkey(d, 0), lib.mstg_conv2d_workspace_bytes(C.byref(d)), x.device)
    _timed(_kernel_name(d, 0), fl, by, lambda: _lib.check(
        lib.mstg_conv2d_fwd_cached(C.byref(d), _p(x), _p(w), _p(b), _p(y), _p(ws), ws.numel() * 4, packed, _stream()), "mstg_conv2d_fwd"),
        _conv_detail("fwd", d))


def conv_dgrad_raw(d: ConvDesc, dy, w, dx, owners=None):
    fl, by = _conv_cost(d)
    lib = _lib.load()
    ws, packed = _cached_ws(owners, "dgrad", _dkey(d, 1), lib.mstg_conv2d_workspace_bytes(C.byref(d)), dy.device)
    _timed(_kernel_name(d, 1), fl, by, lambda: _lib.check(
        lib.mstg_conv2d_dgrad_cached(C.byref(d), _p(dy), _p(w), _p(dx), _p(ws), ws.numel() * 4, packed, _stream()), "mstg_conv2d_dgrad"),
        _conv_detail("dgrad", d))


def norm_bsums_enabled() -> bool:
    """MSTG_NORM_BSUMS=0: the norm backward's two reductions come from its own statistics pass again (A/B, tests)."""
    return os.environ.get("MSTG_NORM_BSUMS", "1") != "0"


def conv_dgrad_bsums_supported(d: ConvDesc) -> bool:
    return norm_bsums_enabled() and bool(_lib.load().mstg_conv2d_dgrad_bsums_supported(C.byref(d)))


def conv_dgrad_bsums_raw(d: ConvDesc, dy, w, dx, x_raw, x_stats, owners=None):
    """dx = input gradient of the convolution whose input was ReLU(InstanceNorm(x_raw)); returns sums (N, 1, 2, Cin): what that norm's
    backward needs from a pass over (x_raw, dx), summed in this launch's epilogue instead (mstg_conv2d_dgrad_bsums)."""
    fl, by = _conv_cost(d)
    lib = _lib.load()
    sums = torch.empty((d.N, 1, 2, d.Cin), dtype=torch.float32, device=dy.device)
    ws, packed = _cached_ws(owners, "dgrad_bsums", _dkey(d, 1), lib.mstg_conv2d_dgrad_bsums_workspace_bytes(C.byref(d)), dy.device)
    _timed(_kernel_name(d, 1), fl, by + 4 * x_raw.numel(), lambda: _lib.check(
        lib.mstg_conv2d_dgrad_bsums(C.byref(d), _p(dy), _p(w), _p(dx), _p(x_raw), _p(x_stats), _p(sums), _p(ws), ws.numel() * 4, packed,
                                    _stream()), "mstg_conv2d_dgrad_bsums"), _conv_detail("dgrad+bsums", d))
    return sums


class _NormToken:
    """Identity of one InstanceNorm + ReLU application: travels forward on the norm's output tensor (``_mstg_norm``) to the
    convolution that consumes it, and back on that convolution's input gradient (``_mstg_bsums``) to the norm's backward."""
    __slots__ = ()


def _tag_norm_output(y, x_raw, stats, token):
    if token is not None:
        y._mstg_norm = (x_raw, stats, token)
    return y


def conv_wgrad_raw(d: ConvDesc, x, dy, dw, db=None):
    """dw (and, for Conv2d, the bias gradient db in the same pass over dy)."""
    lib = _lib.load()
    nbytes = lib.mstg_conv2d_wgrad_workspace_bytes(C.byref(d))
    ws = _ws(nbytes, x.device)
    fl, by = _conv_cost(d)
    sym = _kernel_name(d, 2)
    _timed(sym, fl, by, lambda: _lib.check(
        lib.mstg_conv2d_wgrad(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), _p(ws), ws.numel() * 4, _stream()), "mstg_conv2d_wgrad"),
        _conv_detail("wgrad", d))


def channel_sum(x: Tensor, P: int, ctot: int, coff: int, Cn: int, scale: float = 1.0) -> Tensor:
    """out[c] = scale * sum_p x[p, coff + c] over an NHWC tensor viewed as (P, ctot)."""
    lib = _lib.load()
    out = torch.empty(Cn, dtype=torch.float32, device=x.device)
    ws = _ws(lib.mstg_channel_sum_workspace_bytes(P, Cn), x.device)
    _lib.check(lib.mstg_channel_sum(_p(x), P, ctot, coff, Cn, scale, _p(out), _p(ws), ws.numel() * 4, _stream()),
               "mstg_channel_sum")
    return out


def plane_sum_nchw(x: Tensor) -> Tensor:
    """out[c] = sum_{n,h,w} x[n,c,h,w] for the 3-channel NCHW image tensors."""
    lib = _lib.load()
    N, Cn, H, W = x.shape
    out = torch.empty(Cn, dtype=torch.float32, device=x.device)
    ws = _ws(lib.mstg_plane_sum_workspace_bytes(N, Cn, H * W), x.device)
    _lib.check(lib.mstg_plane_sum(_p(x), N, Cn, H * W, 1.0, _p(out), _p(ws), ws.numel() * 4, _stream()), "mstg_plane_sum")
    return out


# ----------------------------------------------------------------------------------------------------------
# convolution (Conv2d / ConvTranspose2d k4 s2 p1)
# ----------------------------------------------------------------------------------------------------------
class ConvFn(torch.autograd.Function):
    """y = conv(x, w) + b [tanh].  cfg = (k, stride, pad, dil, transposed, x_nchw, y_nchw, act)."""

    @staticmethod
    def forward(ctx, x, w, b, cfg):
        k, stride, pad, dil, transposed, x_nchw, y_nchw, act = cfg
        x, w = _req(x, "conv input"), _req(w, "conv weight")
        b = None if b is None else _req(b, "conv bias")
        if x_nchw:
            N, Cin, H, W = x.shape
        else:
            N, H, W, Cin = x.shape
        Cout = w.shape[1] if transposed else w.shape[0]
        if (w.shape[0] if transposed else w.shape[1]) != Cin:
            raise RuntimeError(f"mstg_hip conv: weight {tuple(w.shape)} does not match {Cin} input channels")
        Ho, Wo = conv_out_hw(H, W, k, stride, pad, dil, transposed)
        if Ho <= 0 or Wo <= 0:
            raise RuntimeError(f"mstg_hip conv: input {H}x{W} too small for kernel {k} (stride {stride}, pad {pad}, dil {dil})")
        y = torch.empty((N, Cout, Ho, Wo) if y_nchw else (N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
        d = make_desc(N, H, W, Cin, Ho, Wo, Cout, k, stride, pad, dil, transposed, x_nchw, y_nchw, act=act)
        conv_fwd_raw(d, x, w, b, y, owners=(w, b))
        ctx.cfg, ctx.dims, ctx.has_bias = cfg, (N, H, W, Cin, Ho, Wo, Cout), b is not None
        ctx.norm_src = getattr(x, "_mstg_norm", None) if not x_nchw else None
        # the objects handed to apply() are nn.Parameters when the layer owns them; only their .grad slots are looked up through
        # these references in backward (see _grad_slot) -- the VALUES used there come from saved_tensors (version-checked)
        ctx.prefs = (w, b)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        k, stride, pad, dil, transposed, x_nchw, y_nchw, act = ctx.cfg
        N, H, W, Cin, Ho, Wo, Cout = ctx.dims
        x, w, y = ctx.saved_tensors
        dy = _req(dy, "conv grad_output")
        if act != ACT_NONE:  # tanh'(pre) = 1 - y^2 ; (leaky) relu'(pre) has the sign of y
            dy = act_bwd_raw(y, dy, act)
        d = make_desc(N, H, W, Cin, Ho, Wo, Cout, k, stride, pad, dil, transposed, x_nchw, y_nchw)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            src = getattr(ctx, "norm_src", None)
            if src is not None and conv_dgrad_bsums_supported(d):
                # x was ReLU(InstanceNorm(src[0])): the reductions of that norm's backward come out of this launch's epilogue
                dx._mstg_bsums = (conv_dgrad_bsums_raw(d, dy, w, dx, src[0], src[1], owners=(ctx.prefs[0],)), src[2])
            else:
                conv_dgrad_raw(d, dy, w, dx, owners=(ctx.prefs[0],))
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        fuse_db = want_db and ctx.needs_input_grad[1] and not transposed
        if ctx.needs_input_grad[1]:
            gw = _grad_slot(ctx.prefs[0])
            gb = _grad_slot(ctx.prefs[1]) if fuse_db else None
            if gw is not None and (gb is not None or not fuse_db):  # accumulate into p.grad, hand autograd nothing
                d.accumulate = 1
                conv_wgrad_raw(d, x, dy, gw, gb)
                d.accumulate = 0
                if fuse_db:
                    want_db = False
            else:
                dw = torch.empty_like(w)
                if fuse_db:
                    db = torch.empty(Cout, dtype=torch.float32, device=dy.device)
                conv_wgrad_raw(d, x, dy, dw, db if fuse_db else None)
        if want_db and not fuse_db:
            db = plane_sum_nchw(dy) if y_nchw else channel_sum(dy, N * Ho * Wo, Cout, 0, Cout)
        return dx, dw, db, None


def conv_norm_supported(N, H, W, Cin, Cout, k, stride, pad, dil, transposed) -> bool:
    """True where mstg_conv2d_fwd_norm runs (the persistent kernel's layers): the InstanceNorm that follows the convolution gets its
    statistics from the convolution's epilogue, and one in front of it is applied while the convolution stages its input."""
    Ho, Wo = conv_out_hw(H, W, k, stride, pad, dil, transposed)
    if Ho * Wo < 256:  # tiny maps: E[x^2] - E[x]^2 from fp32 sums is too coarse against eps when the variance is ~0; use the pass
        return False
    d = make_desc(N, H, W, Cin, Ho, Wo, Cout, k, stride, pad, dil, transposed)
    return bool(_lib.load().mstg_conv2d_fwd_norm_supported(C.byref(d)))


def conv_stats_pays(N, H, W, Cin, Cout, k, stride, pad, dil, transposed) -> bool:
    """conv_norm_supported and the epilogue statistics are cheaper than a statistics pass over the output (not so for the 32 <-> 64
    channel 4x4 layers, whose kernel variant has no registers to spare for the sums)."""
    if not conv_norm_supported(N, H, W, Cin, Cout, k, stride, pad, dil, transposed):
        return False
    Ho, Wo = conv_out_hw(H, W, k, stride, pad, dil, transposed)
    d = make_desc(N, H, W, Cin, Ho, Wo, Cout, k, stride, pad, dil, transposed)
    return bool(_lib.load().mstg_conv2d_fwd_stats_pays(C.byref(d)))


class ConvStatsFn(torch.autograd.Function):
    """(y, stats) = conv(x, w) + b with stats[n][c] = (mean, rstd) of y taken in the convolution's epilogue (NHWC, no activation);
    optional in_stats: x is the raw tensor in front of InstanceNorm + ReLU and is normalised while staged (the caller owns that
    norm's backward).  stats is a non-differentiable output; the backward is ConvFn's, on the normalised input when in_stats is
    given (recomputed by the consumers that need it: see MSFusionFn)."""

    @staticmethod
    def forward(ctx, x, w, b, cfg):
        lib = _lib.load()
        k, stride, pad, dil, transposed = cfg
        x, w = _req(x, "conv input"), _req(w, "conv weight")
        b = None if b is None else _req(b, "conv bias")
        N, H, W, Cin = x.shape
        Cout = w.shape[1] if transposed else w.shape[0]
        Ho, Wo = conv_out_hw(H, W, k, stride, pad, dil, transposed)
        y = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
        stats = torch.empty((N, Cout, 2), dtype=torch.float32, device=x.device)
        d = make_desc(N, H, W, Cin, Ho, Wo, Cout, k, stride, pad, dil, transposed)
        ws, packed = _cached_ws((w, b), "fwd_norm", _dkey(d, 0), lib.mstg_conv2d_fwd_norm_workspace_bytes(C.byref(d)), x.device)
        fl, by = _conv_cost(d)
        _timed(_kernel_name(d, 0).replace(", false>", ", true>"), fl, by, lambda: _lib.check(  # the STATS instantiation
            lib.mstg_conv2d_fwd_norm_cached(C.byref(d), _p(x), None, _p(w), _p(b), _p(y), _p(stats), _p(ws), ws.numel() * 4, packed,
                                            _stream()), "mstg_conv2d_fwd_norm"), _conv_detail("fwd", d))
        ctx.cfg, ctx.dims, ctx.has_bias = (k, stride, pad, dil, transposed, 0, 0, ACT_NONE), (N, H, W, Cin, Ho, Wo, Cout), b is not None
        ctx.prefs = (w, b)
        ctx.norm_src = getattr(x, "_mstg_norm", None)
        ctx.save_for_backward(x, w, None)
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        return ConvFn.backward(ctx, dy)


def ms_fusion_supported(N, H, W, Cn) -> bool:
    """True where the MultiScaleBlock's concat norm can be folded into the 1x1 fusion convolution (forward: normalise-on-load in
    the persistent conv kernel; backward: normalise-on-load in the 1x1 weight-gradient kernel)."""
    d = make_desc(N, H, W, Cn, H, W, Cn, 1, 1, 0, 1)
    lib = _lib.load()
    return bool(lib.mstg_conv2d_fwd_norm_supported(C.byref(d))) and bool(lib.mstg_conv2d_wgrad_norm_supported(C.byref(d)))


class MSFusionFn(torch.autograd.Function):
    """f = Conv1x1(ReLU(InstanceNorm2d(cat))) of the MultiScaleBlock (enhanced_generator.py:72-75, 83) without the normalised concat
    ever being written: one statistics pass over the raw concat, then the fusion convolution normalises while it stages its tiles;
    in the backward the 1x1 weight-gradient kernel does the same, and the norm's own backward works from the raw tensor as always.
    Two tensor passes fewer per block than norm kernel + convolution, with the same arithmetic on every element.
    cfg = (k, stride, pad, dil) (default 1x1): the same fold for the stem's InstanceNorm + ReLU in front of down1's 4x4 stride-2
    convolution (enhanced_generator.py:93-95 -> :97), where the persistent convolution / weight-gradient kernels normalise on load."""

    @staticmethod
    def forward(ctx, cat, w, b, cfg=(1, 1, 0, 1)):
        lib = _lib.load()
        cat, w = _req(cat, "fusion input"), _req(w, "fusion weight")
        b = None if b is None else _req(b, "fusion bias")
        N, H, W, Cn = cat.shape
        Cout = w.shape[0]
        k, stride, pad, dil = cfg
        Ho, Wo = conv_out_hw(H, W, k, stride, pad, dil, False)
        stats = torch.empty((N, Cn, 2), dtype=torch.float32, device=cat.device)
        ws = _ws(lib.mstg_norm_workspace_bytes(N, H * W, Cn), cat.device)
        _timed("norm_partial_kernel<false>", 0, 4 * cat.numel(), lambda: _lib.check(
            lib.mstg_norm_stats(_p(cat), _p(stats), N, H * W, Cn, _p(ws), ws.numel() * 4, _stream()), "mstg_norm_stats"))
        y = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=cat.device)
        ystats = torch.empty((N, Cout, 2), dtype=torch.float32, device=cat.device)  # (mean, rstd) of y from the epilogue
        d = make_desc(N, H, W, Cn, Ho, Wo, Cout, k, stride, pad, dil)
        ws2, packed = _cached_ws((w, b), "fwd_norm", _dkey(d, 0), lib.mstg_conv2d_fwd_norm_workspace_bytes(C.byref(d)), cat.device)
        fl, by = _conv_cost(d)
        _timed(_kernel_name(d, 0).replace(", false>", ", true>"), fl, by, lambda: _lib.check(
            lib.mstg_conv2d_fwd_norm_cached(C.byref(d), _p(cat), _p(stats), _p(w), _p(b), _p(y), _p(ystats), _p(ws2), ws2.numel() * 4,
                                            packed, _stream()), "mstg_conv2d_fwd_norm"), _conv_detail("fwd", d))
        ctx.dims, ctx.has_bias, ctx.prefs, ctx.cfg = (N, H, W, Cn, Cout), b is not None, (w, b), tuple(cfg)
        ctx.save_for_backward(cat, stats, w)
        ctx.mark_non_differentiable(ystats)
        return y, ystats

    @staticmethod
    def backward(ctx, dy, _dystats):
        lib = _lib.load()
        N, H, W, Cn, Cout = ctx.dims
        cat, stats, w = ctx.saved_tensors
        dy = _req(dy, "fusion grad_output")
        k, stride, pad, dil = ctx.cfg
        Ho, Wo = conv_out_hw(H, W, k, stride, pad, dil, False)
        d = make_desc(N, H, W, Cn, Ho, Wo, Cout, k, stride, pad, dil)
        dcat = dw = db = None
        if ctx.needs_input_grad[0]:
            dz = torch.empty_like(cat)
            dcat = torch.empty_like(cat)
            if conv_dgrad_bsums_supported(d):  # gradient w.r.t. the normalised concat + the norm backward's reductions, one launch
                sums = conv_dgrad_bsums_raw(d, dy, w, dz, cat, stats, owners=(ctx.prefs[0],))
                _timed("norm_apply_kernel<true>", 0, 4 * cat.numel() * 3, lambda: _lib.check(
                    lib.mstg_norm_bwd_apply(_p(cat), _p(stats), _p(dz), _p(sums), 1, _p(dcat), N, H * W, Cn, ACT_RELU, _stream()),
                    "mstg_norm_bwd_apply"))
            else:
                conv_dgrad_raw(d, dy, w, dz, owners=(ctx.prefs[0],))
                ws = _ws(lib.mstg_norm_workspace_bytes(N, H * W, Cn), cat.device)
                _timed("norm_apply_kernel<true>", 0, 4 * cat.numel() * 3, lambda: _lib.check(
                    lib.mstg_norm_act_bwd(_p(cat), _p(stats), _p(dz), _p(dcat), N, H * W, Cn, ACT_RELU, 0, None, None, None, None, _p(ws),
                                          ws.numel() * 4, _stream()), "mstg_norm_act_bwd"))
        if ctx.needs_input_grad[1]:
            want_db = ctx.has_bias and ctx.needs_input_grad[2]
            gw = _grad_slot(ctx.prefs[0])
            gb = _grad_slot(ctx.prefs[1]) if want_db else None
            direct = gw is not None and (gb is not None or not want_db)
            if not direct:
                dw = torch.empty_like(w)
                db = torch.empty(Cout, dtype=torch.float32, device=dy.device) if want_db else None
            d.accumulate = 1 if direct else 0
            out_w, out_b = (gw, gb) if direct else (dw, db)
            wsw = _ws(lib.mstg_conv2d_wgrad_workspace_bytes(C.byref(d)), cat.device)
            fl, by = _conv_cost(d)
            _timed(_kernel_name(d, 2), fl, by, lambda: _lib.check(
                lib.mstg_conv2d_wgrad_norm(C.byref(d), _p(cat), _p(stats), _p(dy), _p(out_w), _p(out_b), _p(wsw), wsw.numel() * 4, _stream()),
                "mstg_conv2d_wgrad_norm"), _conv_detail("wgrad", d))
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db = channel_sum(dy, N * Ho * Wo, Cout, 0, Cout)
        return dcat, dw, db, None


def norm_conv_supported(N, H, W, Cin, Cout, k, stride, pad, dil) -> bool:
    """MSFusionFn's fold for a k x k convolution: forward with in- and out-statistics and the weight gradient with in-statistics."""
    if not conv_stats_pays(N, H, W, Cin, Cout, k, stride, pad, dil, False):
        return False
    Ho, Wo = conv_out_hw(H, W, k, stride, pad, dil, False)
    d = make_desc(N, H, W, Cin, Ho, Wo, Cout, k, stride, pad, dil)
    return bool(_lib.load().mstg_conv2d_wgrad_norm_supported(C.byref(d)))


def conv2d_stats(x, w, b, k, stride=1, pad=0, dil=1, transposed=False):
    return ConvStatsFn.apply(x, w, b, (k, stride, pad, dil, int(transposed)))


def conv2d(x, w, b, k, stride=1, pad=0, dil=1, transposed=False, x_nchw=False, y_nchw=False, act=ACT_NONE):
    return ConvFn.apply(x, w, b, (k, stride, pad, dil, int(transposed), int(x_nchw), int(y_nchw), act))


class MSBranchesFn(torch.autograd.Function):
    """The four parallel branch convolutions of MultiScaleBlock (enhanced_generator.py:52-71,79-83) writing straight
    into the channel-concatenated buffer: 1x1, 3x3 d1, 3x3 d2, 3x3 d4, each ch -> ch/4.  No torch.cat copy."""

    GEOM = ((1, 0, 1), (3, 1, 1), (3, 2, 2), (3, 4, 4))  # (k, pad, dil)

    @staticmethod
    def forward(ctx, x, *wb):
        x = _req(x, "multi-scale input")
        ws = [_req(t, "branch weight") for t in wb[0::2]]
        bs = [_req(t, "branch bias") for t in wb[1::2]]
        N, H, W, ch = x.shape
        c4 = ws[0].shape[0]
        y = torch.empty((N, H, W, 4 * c4), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        if os.environ.get("MSTG_MS_UNFUSED", "0") != "1" and bool(lib.mstg_msblock_fused_supported(ch)) and 4 * c4 == ch:
            # (H >= 16: smaller maps go to another kernel, which packs the filter differently)
            wsb, packed = _cached_ws(tuple(wb), "ms_fwd", (ch, H >= 16), lib.mstg_msblock_fwd_workspace_bytes(ch), x.device)
            wb_ptrs = []
            for j in range(4):
                wb_ptrs += [_p(ws[j]), _p(bs[j])]
            fwd4 = ch == 16 and H >= 16 and os.environ.get("MSTG_MS_FWD4", "1") != "0"
            _timed("ms_fwd4_kernel" if fwd4 else f"ms_fwd_kernel<{ch}>", 2.0 * N * H * W * ch * c4 * 28, 4.0 * (2 * N * H * W * ch),
                   lambda: _lib.check(lib.mstg_msblock_fwd_cached(_p(x), *wb_ptrs, _p(y), N, H, W, ch, _p(wsb), wsb.numel() * 4, packed,
                                                                  _stream()), "mstg_msblock_fwd"), f"ms-fwd N{N} {H}x{W} ch{ch}")
        else:
            for j, (k, pad, dil) in enumerate(MSBranchesFn.GEOM):
                d = make_desc(N, H, W, ch, H, W, c4, k, 1, pad, dil, y_ctot=4 * c4, y_coff=j * c4)
                conv_fwd_raw(d, x, ws[j], bs[j], y)
        ctx.dims = (N, H, W, ch, c4)
        ctx.prefs = tuple(wb)
        ctx.save_for_backward(x, *ws)
        # second output: x itself, for the block's residual connection.  Routing the residual through this Function brings its
        # gradient into backward() below, where the fused dgrad kernel adds it in its epilogue instead of autograd running a
        # separate full-tensor add.
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres=None):
        N, H, W, ch, c4 = ctx.dims
        x, *ws = ctx.saved_tensors
        dy = _req(dy, "multi-scale grad_output")
        dres = None if dres is None else _req(dres, "multi-scale residual grad")
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        lib = _lib.load()
        fused = os.environ.get("MSTG_MS_UNFUSED", "0") != "1" and bool(lib.mstg_msblock_fused_supported(ch)) and 4 * c4 == ch
        grads = []
        if fused and dx is not None:  # dx of all four branches in one pass over dy, written once
            wsd, packed = _cached_ws(tuple(ctx.prefs[0::2]), "ms_dgrad", (ch,), lib.mstg_msblock_dgrad_workspace_bytes(ch), x.device)
            _timed(f"ms_dgrad_kernel<{ch}>", 2.0 * N * H * W * ch * c4 * 28, 4.0 * (2 * N * H * W * ch),
                   lambda: _lib.check(lib.mstg_msblock_dgrad_cached(_p(dy), *[_p(t) for t in ws], _p(dres), _p(dx), N, H, W, ch, _p(wsd),
                                                                    wsd.numel() * 4, packed, _stream()), "mstg_msblock_dgrad"),
                   f"ms-dgrad N{N} {H}x{W} ch{ch}")
        for j, (k, pad, dil) in enumerate(MSBranchesFn.GEOM):
            d = make_desc(N, H, W, ch, H, W, c4, k, 1, pad, dil, y_ctot=4 * c4, y_coff=j * c4, accumulate=int(j > 0))
            if dx is not None and not fused:
                conv_dgrad_raw(d, dy, ws[j], dx)
            dw = torch.empty_like(ws[j])
            db = torch.empty(c4, dtype=torch.float32, device=dy.device)
            if not fused:
                d.accumulate = 0
                conv_wgrad_raw(d, x, dy, dw, db)
            grads += [dw, db]
        if dx is not None and dres is not None and not fused:
            dx.add_(dres)
        if dx is None and dres is not None:
            dx = dres
        if fused:  # all eight parameter gradients in one pass over x and dy
            wsb = _ws(lib.mstg_msblock_wgrad_workspace_bytes(N, H, W, ch), x.device)
            slots = [_grad_slot(p) for p in ctx.prefs]
            direct = all(t is not None for t in slots)
            outs = slots if direct else grads
            sym = "wgrad_msp_kernel" if ch <= 32 and os.environ.get("MSTG_MS_WGRAD_PACKED", "1") != "0" else "wgrad_ms_kernel"
            _timed(f"{sym}<{ch}>", 2.0 * N * H * W * ch * c4 * 28, 4.0 * (2 * N * H * W * ch),
                   lambda: _lib.check(lib.mstg_msblock_wgrad(_p(x), _p(dy), *[_p(t) for t in outs], int(direct), N, H, W, ch, _p(wsb),
                                                             wsb.numel() * 4, _stream()), "mstg_msblock_wgrad"),
                   f"ms-wgrad N{N} {H}x{W} ch{ch}")
            if direct:
                grads = [None] * 8
        return (dx, *grads)


# ----------------------------------------------------------------------------------------------------------
# InstanceNorm / BatchNorm + activation (+ residual)
# ----------------------------------------------------------------------------------------------------------
class InstNormActFn(torch.autograd.Function):
    """y = act(InstanceNorm2d(x)) [+ residual], NHWC."""

    @staticmethod
    def forward(ctx, x, residual, act, token=None):
        lib = _lib.load()
        ctx.token = token
        x = _req(x, "norm input")
        residual = None if residual is None else _req(residual, "norm residual")
        N, H, W, Cn = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((N, Cn, 2), dtype=torch.float32, device=x.device)
        ws = _ws(lib.mstg_norm_workspace_bytes(N, H * W, Cn), x.device)
        _timed("norm_apply_kernel<false>", 0, 4 * x.numel() * (2 if residual is None else 3), lambda: _lib.check(
            lib.mstg_norm_act_fwd(_p(x), _p(residual), _p(y), _p(stats), N, H * W, Cn, act, 0, None, None, None, None,
                                  _p(ws), ws.numel() * 4, _stream()), "mstg_norm_act_fwd"))
        ctx.act, ctx.has_res = act, residual is not None
        ctx.save_for_backward(x, stats)
        InstNormActFn.last_stats = stats if token is not None else None  # for instnorm_act() to tag y with (see _NormToken)
        return y

    last_stats = None

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, stats = ctx.saved_tensors
        dy = _req(dy, "norm grad_output")
        N, H, W, Cn = x.shape
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            got = getattr(dy, "_mstg_bsums", None)
            if got is not None and getattr(ctx, "token", None) is not None and got[1] is ctx.token and ctx.act == ACT_RELU:
                # the consumer convolution's input-gradient launch already summed what this backward reduces (ConvFn.backward)
                _timed("norm_apply_kernel<true>", 0, 4 * x.numel() * 3, lambda: _lib.check(
                    lib.mstg_norm_bwd_apply(_p(x), _p(stats), _p(dy), _p(got[0]), 1, _p(dx), N, H * W, Cn, ACT_RELU, _stream()),
                    "mstg_norm_bwd_apply"))
            else:
                ws = _ws(lib.mstg_norm_workspace_bytes(N, H * W, Cn), x.device)
                _timed("norm_apply_kernel<true>", 0, 4 * x.numel() * 3, lambda: _lib.check(
                    lib.mstg_norm_act_bwd(_p(x), _p(stats), _p(dy), _p(dx), N, H * W, Cn, ctx.act, 0, None, None, None, None,
                                          _p(ws), ws.numel() * 4, _stream()), "mstg_norm_act_bwd"))
        return dx, (dy if ctx.has_res and ctx.needs_input_grad[1] else None), None, None


def _norm_token(act):
    return _NormToken() if act == ACT_RELU and norm_bsums_enabled() and torch.is_grad_enabled() else None


def instnorm_act(x, act=ACT_RELU, residual=None):
    token = _norm_token(act)
    y = InstNormActFn.apply(x, residual, act, token)
    if token is not None:
        _tag_norm_output(y, x, InstNormActFn.last_stats, token)
        InstNormActFn.last_stats = None
    return y


class InstNormApplyFn(torch.autograd.Function):
    """y = act(InstanceNorm2d(x)) [+ residual] with the statistics given (a producer's epilogue summed them): the forward is the apply
    pass alone; the backward is InstNormActFn's."""

    @staticmethod
    def forward(ctx, x, stats, residual, act, token=None):
        lib = _lib.load()
        ctx.token = token
        x, stats = _req(x, "norm input"), _req(stats, "norm statistics")
        residual = None if residual is None else _req(residual, "norm residual")
        N, H, W, Cn = x.shape
        y = torch.empty_like(x)
        _timed("norm_apply_kernel<false>", 0, 4 * x.numel() * (2 if residual is None else 3), lambda: _lib.check(
            lib.mstg_norm_apply_fwd(_p(x), _p(stats), _p(residual), _p(y), N, H * W, Cn, act, _stream()), "mstg_norm_apply_fwd"))
        ctx.act, ctx.has_res = act, residual is not None
        ctx.save_for_backward(x, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = InstNormActFn.backward(ctx, dy)[0]
        return dx, None, (dy if ctx.has_res and ctx.needs_input_grad[2] else None), None, None


def instnorm_apply(x, stats, act=ACT_RELU, residual=None):
    token = _norm_token(act)
    y = InstNormApplyFn.apply(x, stats, residual, act, token)
    return _tag_norm_output(y, x, stats, token)


class BatchNormActFn(torch.autograd.Function):
    """y = act(BatchNorm2d(x)), NHWC; training mode updates running stats in place (momentum 0.1)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, act, training):
        lib = _lib.load()
        x, gamma, beta = _req(x, "norm input"), _req(gamma, "bn weight"), _req(beta, "bn bias")
        N, H, W, Cn = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((1, Cn, 2), dtype=torch.float32, device=x.device)
        ws = _ws(lib.mstg_norm_workspace_bytes(N, H * W, Cn), x.device)
        _lib.check(lib.mstg_norm_act_fwd(_p(x), None, _p(y), _p(stats), N, H * W, Cn, act, 1 if training else 2, _p(gamma),
                                         _p(beta), _p(running_mean), _p(running_var), _p(ws), ws.numel() * 4, _stream()),
                   "mstg_norm_act_fwd(batch)")
        ctx.act, ctx.training = act, training
        ctx.save_for_backward(x, stats, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        if not ctx.training:
            raise RuntimeError("mstg_hip: backward through eval-mode BatchNorm is not implemented")
        x, stats, gamma, beta = ctx.saved_tensors
        dy = _req(dy, "norm grad_output")
        N, H, W, Cn = x.shape
        dx, dg, db = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(beta)
        ws = _ws(lib.mstg_norm_workspace_bytes(N, H * W, Cn), x.device)
        _lib.check(lib.mstg_norm_act_bwd(_p(x), _p(stats), _p(dy), _p(dx), N, H * W, Cn, ctx.act, 1, _p(gamma), _p(beta), _p(dg),
                                         _p(db), _p(ws), ws.numel() * 4, _stream()), "mstg_norm_act_bwd(batch)")
        return dx, dg, db, None, None, None, None


# ----------------------------------------------------------------------------------------------------------
# window attention core
# ----------------------------------------------------------------------------------------------------------
def _attn_core_symbol(direction, Cn):
    """Kernel symbol attention.hip launches for this width (for the timing table only)."""
    if Cn <= 32 or (Cn <= 64 and os.environ.get("MSTG_ATTN_BLK64", "1") == "0"):
        return f"attn_core_{direction}_kernel<{16 if Cn <= 16 else (32 if Cn <= 32 else 64)}>"
    if Cn <= 64 or os.environ.get("MSTG_ATTN_BLK4", "1") == "0":
        return f"attn_core_{direction}_blk_kernel<{64 if Cn <= 64 else (128 if Cn <= 128 else 256)}>"
    return f"attn_core_{direction}_blk4_kernel<{128 if Cn <= 128 else 256}>"


class WindowAttnCoreFn(torch.autograd.Function):
    """o = attn(q^, k^) v per 4x4 window; qkv NHWC (N,H,W,3C) -> o NHWC (N,H,W,C)."""

    @staticmethod
    def forward(ctx, qkv):
        qkv = _req(qkv, "qkv")
        N, H, W, C3 = qkv.shape
        Cn = C3 // 3
        o = torch.empty((N, H, W, Cn), dtype=torch.float32, device=qkv.device)
        _timed(_attn_core_symbol('fwd', Cn), 4 * Cn * Cn * N * H * W, 4 * 4 * Cn * N * H * W, lambda: _lib.check(
            _lib.load().mstg_window_attn_core_fwd(_p(qkv), _p(o), N, H, W, Cn, _stream()), "mstg_window_attn_core_fwd"), detail=f"attn-core N{N} {H}x{W} C{Cn}")
        ctx.save_for_backward(qkv)
        return o

    @staticmethod
    def backward(ctx, do):
        (qkv,) = ctx.saved_tensors
        do = _req(do, "attention grad_output")
        N, H, W, C3 = qkv.shape
        dqkv = torch.empty_like(qkv)
        Cn = C3 // 3
        _timed(_attn_core_symbol('bwd', Cn), 8 * Cn * Cn * N * H * W, 4 * 7 * Cn * N * H * W, lambda: _lib.check(
            _lib.load().mstg_window_attn_core_bwd(_p(qkv), _p(do), _p(dqkv), N, H, W, Cn, _stream()), "mstg_window_attn_core_bwd"), detail=f"attn-core N{N} {H}x{W} C{Cn}")
        return dqkv


class WindowAttnCoreWsFn(torch.autograd.Function):
    """The attention core for window sizes other than 4 (mstg_window_attn_ws_*): qkv NHWC (N,H,W,3C) -> o NHWC (N,H,W,C)."""

    @staticmethod
    def forward(ctx, qkv, ws):
        qkv = _req(qkv, "qkv")
        N, H, W, C3 = qkv.shape
        Cn = C3 // 3
        o = torch.empty((N, H, W, Cn), dtype=torch.float32, device=qkv.device)
        _timed("attn_ws_fwd_kernel", 4 * Cn * Cn * N * H * W, 4 * 4 * Cn * N * H * W, lambda: _lib.check(
            _lib.load().mstg_window_attn_ws_fwd(_p(qkv), _p(o), N, H, W, Cn, int(ws), _stream()), "mstg_window_attn_ws_fwd"))
        ctx.ws = int(ws)
        ctx.save_for_backward(qkv)
        return o

    @staticmethod
    def backward(ctx, do):
        (qkv,) = ctx.saved_tensors
        do = _req(do, "attention grad_output")
        N, H, W, C3 = qkv.shape
        Cn = C3 // 3
        dqkv = torch.empty_like(qkv)
        _timed("attn_ws_bwd_kernel", 8 * Cn * Cn * N * H * W, 4 * 7 * Cn * N * H * W, lambda: _lib.check(
            _lib.load().mstg_window_attn_ws_bwd(_p(qkv), _p(do), _p(dqkv), N, H, W, Cn, ctx.ws, _stream()), "mstg_window_attn_ws_bwd"))
        return dqkv, None


def _attn_grad_dst(prefs, like):
    """Where the fused attention's four parameter gradients go: the parameters' own .grad slots (accumulating; autograd then gets
    None and launches nothing) when ops.direct_param_grads() is on and every slot exists, else four fresh tensors."""
    slots = [_grad_slot(p) for p in prefs]
    if all(t is not None for t in slots):
        return slots, True
    return [torch.empty_like(t) for t in like], False


def _attn_fused_symbol(direction, Cn, norm):
    """Kernel symbol attention_reg.hip launches for this width (for the timing table only)."""
    stem = "attn_big" if Cn > 32 or (Cn == 32 and os.environ.get("MSTG_ATTN_BIG32") == "1") else "attn_reg"
    return f"{stem}_{direction}_kernel<{Cn}, {'true' if norm else 'false'}>"


class LocalAttentionFusedFn(torch.autograd.Function):
    """Whole LocalAttention (qkv 1x1 conv -> window attention -> proj 1x1 conv) in one kernel per direction, C = 16 / 32 / 64.
    x, y: NHWC (N,H,W,C); wqkv (3C,C,1,1), wproj (C,C,1,1) as stored by the reference."""

    @staticmethod
    def forward(ctx, x, wqkv, bqkv, wproj, bproj):
        x = _req(x, "attention input")
        wqkv, bqkv, wproj, bproj = (_req(t, "attention parameter") for t in (wqkv, bqkv, wproj, bproj))
        N, H, W, Cn = x.shape
        y = torch.empty_like(x)
        _timed(_attn_fused_symbol("fwd", Cn, False), 12 * Cn * Cn * N * H * W, 4 * 2 * Cn * N * H * W, lambda: _lib.check(
            _lib.load().mstg_window_attn_fwd(_p(x), _p(wqkv), _p(bqkv), _p(wproj), _p(bproj), _p(y), N, H, W, Cn, _stream()),
            "mstg_window_attn_fwd"))
        ctx.prefs = (wqkv, bqkv, wproj, bproj)
        ctx.save_for_backward(x, wqkv, bqkv, wproj, bproj)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, wqkv, bqkv, wproj, bproj = ctx.saved_tensors
        dy = _req(dy, "attention grad_output")
        N, H, W, Cn = x.shape
        dx = torch.empty_like(x)
        outs, direct = _attn_grad_dst(ctx.prefs, (wqkv, bqkv, wproj, bproj))
        ws = _ws(lib.mstg_window_attn_bwd_workspace_bytes(N, H, W, Cn), x.device)
        _timed(_attn_fused_symbol("bwd", Cn, False), 24 * Cn * Cn * N * H * W, 4 * 3 * Cn * N * H * W, lambda: _lib.check(
            lib.mstg_window_attn_bwd_direct(_p(x), _p(wqkv), _p(bqkv), _p(wproj), _p(bproj), _p(dy), _p(dx), *[_p(t) for t in outs],
                                            int(direct), N, H, W, Cn, _p(ws), ws.numel() * 4, _stream()), "mstg_window_attn_bwd"))
        return (dx, *((None,) * 4 if direct else outs))


class NormLocalAttentionFn(torch.autograd.Function):
    """LocalAttention(ReLU(InstanceNorm2d(x))) with the norm folded into the attention kernels (C = 16 / 32 / 64): statistics pass over
    x, then the attention forward normalises while it stages each window; the backward's attention kernel sums what the norm's
    backward needs in its epilogue, so the norm costs one read in the forward and one read-read-write pass in the backward
    instead of 3 + 5 tensor passes -- and the normalised tensor is never stored."""

    @staticmethod
    def forward(ctx, x, wqkv, bqkv, wproj, bproj, stats=None):
        lib = _lib.load()
        x = _req(x, "norm input")
        wqkv, bqkv, wproj, bproj = (_req(t, "attention parameter") for t in (wqkv, bqkv, wproj, bproj))
        N, H, W, Cn = x.shape
        if stats is None:  # no producer epilogue delivered them: one pass over x
            stats = torch.empty((N, Cn, 2), dtype=torch.float32, device=x.device)
            ws = _ws(lib.mstg_norm_workspace_bytes(N, H * W, Cn), x.device)
            _timed("norm_partial_kernel<false>", 0, 4 * x.numel(), lambda: _lib.check(
                lib.mstg_norm_stats(_p(x), _p(stats), N, H * W, Cn, _p(ws), ws.numel() * 4, _stream()), "mstg_norm_stats"))
        y = torch.empty_like(x)
        _timed(_attn_fused_symbol("fwd", Cn, True), 12 * Cn * Cn * N * H * W, 4 * 2 * Cn * N * H * W, lambda: _lib.check(
            lib.mstg_window_attn_norm_fwd(_p(x), _p(stats), _p(wqkv), _p(bqkv), _p(wproj), _p(bproj), _p(y), N, H, W, Cn, _stream()),
            "mstg_window_attn_norm_fwd"))
        ctx.prefs = (wqkv, bqkv, wproj, bproj)
        ctx.save_for_backward(x, stats, wqkv, bqkv, wproj, bproj)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, stats, wqkv, bqkv, wproj, bproj = ctx.saved_tensors
        dy = _req(dy, "attention grad_output")
        N, H, W, Cn = x.shape
        dz = torch.empty_like(x)
        outs, direct = _attn_grad_dst(ctx.prefs, (wqkv, bqkv, wproj, bproj))
        S = lib.mstg_window_attn_norm_sums_split()
        sums = torch.empty((N, S, 2, Cn), dtype=torch.float32, device=x.device)
        ws = _ws(lib.mstg_window_attn_norm_bwd_workspace_bytes(N, H, W, Cn), x.device)
        _timed(_attn_fused_symbol("bwd", Cn, True), 24 * Cn * Cn * N * H * W, 4 * 3 * Cn * N * H * W, lambda: _lib.check(
            lib.mstg_window_attn_norm_bwd_direct(_p(x), _p(stats), _p(wqkv), _p(bqkv), _p(wproj), _p(bproj), _p(dy), _p(dz),
                                                 *[_p(t) for t in outs], int(direct), _p(sums), N, H, W, Cn, _p(ws), ws.numel() * 4,
                                                 _stream()), "mstg_window_attn_norm_bwd"))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _timed("norm_apply_kernel<true>", 0, 4 * x.numel() * 3, lambda: _lib.check(
                lib.mstg_norm_bwd_apply(_p(x), _p(stats), _p(dz), _p(sums), S, _p(dx), N, H * W, Cn, ACT_RELU, _stream()),
                "mstg_norm_bwd_apply"))
        return (dx, *((None,) * 4 if direct else outs), None)


def fused_attention_supported(Cn: int) -> bool:
    return bool(_lib.load().mstg_window_attn_fused_supported(int(Cn)))


# ----------------------------------------------------------------------------------------------------------
# element-wise, losses, Adam
# ----------------------------------------------------------------------------------------------------------
def act_fwd_raw(x, act):
    y = torch.empty_like(x)
    _lib.check(_lib.load().mstg_act_fwd(_p(x), _p(y), x.numel(), act, _stream()), "mstg_act_fwd")
    return y


def act_bwd_raw(x_or_y, dy, act):
    dx = torch.empty_like(dy)
    _lib.check(_lib.load().mstg_act_bwd(_p(x_or_y), _p(dy), _p(dx), dy.numel(), act, _stream()), "mstg_act_bwd")
    return dx


class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        x = _req(x, "activation input")
        y = act_fwd_raw(x, act)
        ctx.act = act
        ctx.save_for_backward(y if act == ACT_TANH else x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (t,) = ctx.saved_tensors
        return act_bwd_raw(t, _req(dy, "activation grad_output"), ctx.act), None


def activation(x, act):
    return ActFn.apply(x, act)


class MeanLossFn(torch.autograd.Function):
    """mean(|a-b|) (kind LOSS_L1) or mean((a-b)^2) (kind LOSS_MSE); b may be None => constant bconst."""

    @staticmethod
    def forward(ctx, a, b, bconst, kind):
        lib = _lib.load()
        a = _req(a, "loss input")
        b = None if b is None else _req(b, "loss target")
        if b is not None and b.shape != a.shape:
            raise RuntimeError(f"mstg_hip loss: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
        out = torch.empty((), dtype=torch.float32, device=a.device)
        ws = _ws(lib.mstg_loss_workspace_bytes(a.numel()), a.device)
        _lib.check(lib.mstg_loss_mean_fwd(_p(a), _p(b), float(bconst), a.numel(), kind, _p(out), _p(ws), ws.numel() * 4, _stream()),
                   "mstg_loss_mean_fwd")
        ctx.kind, ctx.bconst = kind, float(bconst)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _req(g, "loss grad_output").reshape(1)
        need_b = b is not None and ctx.needs_input_grad[1]
        da = torch.empty_like(a)
        db = torch.empty_like(b) if need_b else None
        _lib.check(_lib.load().mstg_loss_mean_bwd(_p(a), _p(b), ctx.bconst, a.numel(), ctx.kind, _p(g), 1.0, _p(da), _p(db), _stream()),
                   "mstg_loss_mean_bwd")
        return (da if ctx.needs_input_grad[0] else None), db, None, None


def l1_loss(a, b):
    return MeanLossFn.apply(a, b, 0.0, LOSS_L1)


def mse_loss(a, b):
    return MeanLossFn.apply(a, b, 0.0, LOSS_MSE)


class MaskedL1Fn(torch.autograd.Function):
    """mean(|gen * (1 - mask) - real * (1 - mask)|): the masked-image pre-training loss (pretrain.py:160-162); gradient to gen only."""

    @staticmethod
    def forward(ctx, gen, real, mask):
        lib = _lib.load()
        gen, real, mask = _req(gen, "generated image"), _req(real, "real image"), _req(mask, "mask")
        if gen.shape != real.shape or gen.shape != mask.shape:
            raise RuntimeError("mstg_hip masked_l1: generated image, real image and mask must have one shape")
        out = torch.empty((), dtype=torch.float32, device=gen.device)
        ws = _ws(lib.mstg_loss_workspace_bytes(gen.numel()), gen.device)
        _lib.check(lib.mstg_masked_l1_mean_fwd(_p(gen), _p(real), _p(mask), gen.numel(), _p(out), _p(ws), ws.numel() * 4, _stream()),
                   "mstg_masked_l1_mean_fwd")
        ctx.save_for_backward(gen, real, mask)
        return out

    @staticmethod
    def backward(ctx, g):
        gen, real, mask = ctx.saved_tensors
        g = _req(g, "loss grad_output").reshape(1)
        da = torch.empty_like(gen)
        _lib.check(_lib.load().mstg_masked_l1_mean_bwd(_p(gen), _p(real), _p(mask), gen.numel(), _p(g), _p(da), _stream()),
                   "mstg_masked_l1_mean_bwd")
        return da, None, None


def masked_l1_loss(gen, real, mask):
    return MaskedL1Fn.apply(gen, real, mask)


def clip_grad_norm_flat_(flat_grad: Tensor, max_norm: float) -> Tensor:
    """clip_grad_norm_ over an optimizer's flat gradient buffer (FlatAdam.grad); returns the pre-clip norm as a 0-dim device tensor."""
    lib = _lib.load()
    norm = torch.empty((), dtype=torch.float32, device=flat_grad.device)
    ws = _ws(lib.mstg_loss_workspace_bytes(flat_grad.numel()), flat_grad.device)
    _lib.check(lib.mstg_clip_grad_norm(_p(flat_grad), flat_grad.numel(), float(max_norm), _p(norm), _p(ws), ws.numel() * 4, _stream()),
               "mstg_clip_grad_norm")
    return norm


class WeightedSumFn(torch.autograd.Function):
    """out[j] = sum_i W[j][i] * term_i over 0-dim loss tensors in one launch; row 0 is the differentiable total (one launch for its
    backward), further rows are reported components (no gradient): enhanced_train.py:72-81, 95-131."""

    @staticmethod
    def forward(ctx, weights, *terms):
        terms = [_req(t, "loss term").reshape(()) for t in terms]
        n, nout = len(terms), len(weights)
        out = torch.empty(nout, dtype=torch.float32, device=terms[0].device)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in terms])
        wts = (C.c_float * (n * nout))(*[float(w) for row in weights for w in row])
        _lib.check(_lib.load().mstg_weighted_sum_fwd(ptrs, wts, n, nout, _p(out), _stream()), "mstg_weighted_sum_fwd")
        ctx.weights, ctx.keep = [float(w) for w in weights[0]], terms  # keep the terms alive until the launch has read them
        total, rest = out[0], out[1:]
        ctx.mark_non_differentiable(rest)
        return total, rest

    @staticmethod
    def backward(ctx, g, _grest):
        n = len(ctx.weights)
        g = _req(g, "loss grad_output").reshape(1)
        out = torch.empty(n, dtype=torch.float32, device=g.device)
        wts = (C.c_float * n)(*ctx.weights)
        _lib.check(_lib.load().mstg_weighted_sum_bwd(_p(g), wts, n, _p(out), _stream()), "mstg_weighted_sum_bwd")
        return (None, *[out[i].reshape(()) for i in range(n)])


def weighted_sum(terms, weights, report=()):
    """total = sum_i weights[i] * terms[i] over 0-dim device tensors; ``report``: further weight rows whose sums are returned as a
    (len(report),) tensor without gradient.  Returns total, or (total, reported) when ``report`` is given."""
    total, rest = WeightedSumFn.apply((tuple(weights),) + tuple(tuple(r) for r in report), *terms)
    return (total, rest) if report else total


def mse_to_const(a, value: float):
    """nn.MSELoss()(a, full_like(a, value)) -- the LSGAN targets of enhanced_train.py:72-79,100-101."""
    return MeanLossFn.apply(a, None, float(value), LOSS_MSE)


class SpatialMeanFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d(1) on NHWC: (N,H,W,C) -> (N,C)  (enhanced_generator.py:143,257)."""

    @staticmethod
    def forward(ctx, x):
        x = _req(x, "pool input")
        N, H, W, Cn = x.shape
        ctx.shape = (N, H, W, Cn)
        out = torch.empty((N, Cn), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().mstg_segment_mean_fwd(_p(x), N, H * W, Cn, _p(out), _stream()), "mstg_segment_mean_fwd")
        return out

    @staticmethod
    def backward(ctx, dy):
        N, H, W, Cn = ctx.shape
        dy = _req(dy, "pool grad_output")
        dx = torch.empty((N, H, W, Cn), dtype=torch.float32, device=dy.device)
        _lib.check(_lib.load().mstg_segment_mean_bwd(_p(dy), N, H * W, Cn, _p(dx), _stream()), "mstg_segment_mean_bwd")
        return dx


def spatial_mean(x):
    return SpatialMeanFn.apply(x)


class MaxPool2x2Fn(torch.autograd.Function):
    """F.max_pool2d(x, 2) on NHWC; the arg-max slot of every element is kept as a byte for the backward."""

    @staticmethod
    def forward(ctx, x):
        x = _req(x, "pool input")
        N, H, W, Cn = x.shape
        y = torch.empty((N, H // 2, W // 2, Cn), dtype=torch.float32, device=x.device)
        idx = torch.empty((N, H // 2, W // 2, Cn), dtype=torch.uint8, device=x.device)
        _lib.check(_lib.load().mstg_maxpool2x2_fwd(_p(x), _p(y), _p(idx), N, H, W, Cn, _stream()), "mstg_maxpool2x2_fwd")
        ctx.shape = (N, H, W, Cn)
        ctx.save_for_backward(idx)
        ctx.mark_non_differentiable(idx)
        return y, idx

    @staticmethod
    def backward(ctx, dy, _didx):
        (idx,) = ctx.saved_tensors
        N, H, W, Cn = ctx.shape
        dy = _req(dy, "pool grad_output")
        dx = torch.zeros((N, H, W, Cn), dtype=torch.float32, device=dy.device) if (H & 1 or W & 1) else \
            torch.empty((N, H, W, Cn), dtype=torch.float32, device=dy.device)
        _lib.check(_lib.load().mstg_maxpool2x2_bwd(_p(dy), _p(idx), _p(dx), N, H, W, Cn, _stream()), "mstg_maxpool2x2_bwd")
        return dx


def maxpool2x2(x, return_indices=False):
    y, idx = MaxPool2x2Fn.apply(x)
    return (y, idx) if return_indices else y


class GramFn(torch.autograd.Function):
    """G[n] = F[n]^T F[n] / (C H W) for NHWC features (N,H,W,C) -> (N,C,C)."""

    @staticmethod
    def forward(ctx, f):
        lib = _lib.load()
        f = _req(f, "gram input")
        N, H, W, Cn = f.shape
        scale = 1.0 / (Cn * H * W)
        g = torch.empty((N, Cn, Cn), dtype=torch.float32, device=f.device)
        ws = _ws(lib.mstg_gram_workspace_bytes(N, H * W, Cn), f.device)
        _timed("gram_fwd_kernel", 2.0 * N * H * W * Cn * Cn, 4.0 * (N * H * W * Cn + N * Cn * Cn), lambda: _lib.check(
            lib.mstg_gram_fwd(_p(f), _p(g), N, H * W, Cn, scale, _p(ws), ws.numel() * 4, _stream()), "mstg_gram_fwd"))
        ctx.scale = scale
        ctx.save_for_backward(f)
        return g

    @staticmethod
    def backward(ctx, dg):
        (f,) = ctx.saved_tensors
        dg = _req(dg, "gram grad_output")
        N, H, W, Cn = f.shape
        df = torch.empty_like(f)
        _timed("gram_bwd_kernel", 2.0 * N * H * W * Cn * Cn, 4.0 * 2 * N * H * W * Cn, lambda: _lib.check(
            _lib.load().mstg_gram_bwd(_p(f), _p(dg), _p(df), N, H * W, Cn, ctx.scale, _stream()), "mstg_gram_bwd"))
        return df


def gram_matrix(f):
    return GramFn.apply(f)


class SpectralNormFn(torch.autograd.Function):
    """weight_orig -> weight_orig / sigma with torch.nn.utils.spectral_norm's semantics (one power iteration per
    training-mode call, in place on the module's weight_u / weight_v buffers; u, v constants for autograd)."""

    @staticmethod
    def forward(ctx, w, u, v, eps, training):
        w = _req(w, "spectral_norm weight")
        u, v = _req(u, "spectral_norm u"), _req(v, "spectral_norm v")
        M = w.shape[0]
        K = w.numel() // M
        if u.numel() != M or v.numel() != K:
            raise RuntimeError(f"mstg_hip spectral_norm: u/v sizes {u.numel()}/{v.numel()} do not match weight {tuple(w.shape)}")
        out = torch.empty_like(w)
        sigma = torch.empty(1, dtype=torch.float32, device=w.device)
        lib = _lib.load()
        ws = _ws(lib.mstg_spectral_norm_workspace_bytes(M, K), w.device)
        # the buffers move on at the next forward; this call's backward needs the values it produced (copies written by the
        # kernel itself, and only when a backward can follow)
        need = ctx.needs_input_grad[0]
        us = torch.empty_like(u) if need else None
        vs = torch.empty_like(v) if need else None
        _lib.check(lib.mstg_spectral_norm_fwd(_p(w), _p(u), _p(v), _p(out), _p(sigma), _p(us), _p(vs), M, K, float(eps),
                                              int(bool(training)), _p(ws), ws.numel() * 4, _stream()), "mstg_spectral_norm_fwd")
        ctx.pref = w
        if need:
            ctx.save_for_backward(w, us, vs, sigma)
        ctx.dims = (M, K)
        return out

    @staticmethod
    def backward(ctx, dwn):
        w, u, v, sigma = ctx.saved_tensors
        M, K = ctx.dims
        dwn = _req(dwn, "spectral_norm grad_output")
        slot = _grad_slot(ctx.pref)
        dw = slot if slot is not None else torch.empty_like(w)
        lib = _lib.load()
        ws = _ws(lib.mstg_spectral_norm_workspace_bytes(M, K), w.device)
        _lib.check(lib.mstg_spectral_norm_bwd(_p(dwn), _p(w), _p(u), _p(v), _p(sigma), _p(dw), int(slot is not None), M, K, _p(ws),
                                              ws.numel() * 4, _stream()), "mstg_spectral_norm_bwd")
        return (None if slot is not None else dw), None, None, None, None


def install_fused_spectral_norm(module, name: str = "weight") -> bool:
    """Swap the forward pre-hook that torch.nn.utils.spectral_norm registered on `module` for one that computes the same
    thing in one launch (SpectralNormFn).  Parameters / buffers (`weight_orig`, `weight_u`, `weight_v`) and the state_dict
    hooks stay torch's, so checkpoints are unchanged (enhanced_generator.py:269-271)."""
    from torch.nn.utils.spectral_norm import SpectralNorm
    for key, hook in list(module._forward_pre_hooks.items()):
        if isinstance(hook, SpectralNorm) and hook.name == name:
            if hook.n_power_iterations != 1 or hook.dim != 0:
                return False
            eps = float(hook.eps)
            del module._forward_pre_hooks[key]

            def pre(mod, inputs, _eps=eps, _name=name):
                if mod.__dict__.pop("_mstg_sn_fresh", False):
                    return  # spectral_norm_group() has just normalised this weight (and run its power iteration) for this forward
                setattr(mod, _name, SpectralNormFn.apply(getattr(mod, _name + "_orig"), getattr(mod, _name + "_u"),
                                                         getattr(mod, _name + "_v"), _eps, mod.training))

            module.register_forward_pre_hook(pre)
            module._mstg_sn = (name, eps)
            return True
    return False


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def _int_array(values):
    return (C.c_int * len(values))(*[int(v) for v in values])


class SpectralNormGroupFn(torch.autograd.Function):
    """SpectralNormFn for several weights in one call (mstg_spectral_norm_group_*): apply(eps, training, n, w_0..w_{n-1}, u_0.., v_0..)
    -> n normalised weights.  Three launches for the group instead of up to thirteen; in the backward the weights whose output
    received a gradient go through two launches together."""

    @staticmethod
    def forward(ctx, eps, training, n, *tensors):
        lib = _lib.load()
        ws_, us_, vs_ = tensors[:n], tensors[n:2 * n], tensors[2 * n:3 * n]
        ws_ = [_req(w, "spectral_norm weight") for w in ws_]
        Ms = [w.shape[0] for w in ws_]
        Ks = [w.numel() // w.shape[0] for w in ws_]
        for w, u, v, M, K in zip(ws_, us_, vs_, Ms, Ks):
            _req(u, "spectral_norm u"), _req(v, "spectral_norm v")
            if u.numel() != M or v.numel() != K:
                raise RuntimeError(f"mstg_hip spectral_norm: u/v sizes {u.numel()}/{v.numel()} do not match weight {tuple(w.shape)}")
        dev = ws_[0].device
        outs = [torch.empty_like(w) for w in ws_]
        sigma = torch.empty(n, dtype=torch.float32, device=dev)
        need = any(ctx.needs_input_grad[3:3 + n])
        u_sv = [torch.empty_like(u) for u in us_] if need else None
        v_sv = [torch.empty_like(v) for v in vs_] if need else None
        Ma, Ka = _int_array(Ms), _int_array(Ks)
        ws = _ws(lib.mstg_spectral_norm_group_workspace_bytes(n, Ma, Ka), dev)
        _lib.check(lib.mstg_spectral_norm_group_fwd(n, _ptr_array(ws_), _ptr_array(us_), _ptr_array(vs_), _ptr_array(outs), _p(sigma),
                                                    _ptr_array(u_sv) if need else None, _ptr_array(v_sv) if need else None, Ma, Ka,
                                                    float(eps), int(bool(training)), _p(ws), ws.numel() * 4, _stream()),
                   "mstg_spectral_norm_group_fwd")
        ctx.n, ctx.dims, ctx.prefs = n, (Ms, Ks), tuple(ws_)
        ctx.set_materialize_grads(False)  # an output the loss never used stays None in backward: its weight gets no gradient
        if need:
            ctx.save_for_backward(sigma, *ws_, *u_sv, *v_sv)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dwns):
        lib = _lib.load()
        n = ctx.n
        saved = ctx.saved_tensors
        sigma, ws_, us_, vs_ = saved[0], saved[1:1 + n], saved[1 + n:1 + 2 * n], saved[1 + 2 * n:1 + 3 * n]
        Ms, Ks = ctx.dims
        live = [j for j in range(n) if dwns[j] is not None and ctx.needs_input_grad[3 + j]]
        grads = [None] * n
        if live:
            dws, accs, dwn_l = [], [], []
            for j in live:
                slot = _grad_slot(ctx.prefs[j])
                dws.append(slot if slot is not None else torch.empty_like(ws_[j]))
                accs.append(slot is not None)
                dwn_l.append(_req(dwns[j], "spectral_norm grad_output"))
                if slot is None:
                    grads[j] = dws[-1]
            Ma, Ka = _int_array([Ms[j] for j in live]), _int_array([Ks[j] for j in live])
            sig_ptrs = (C.c_void_p * len(live))(*[sigma.data_ptr() + 4 * j for j in live])
            ws = _ws(lib.mstg_spectral_norm_group_workspace_bytes(len(live), Ma, Ka), sigma.device)
            _lib.check(lib.mstg_spectral_norm_group_bwd(len(live), _ptr_array(dwn_l), _ptr_array([ws_[j] for j in live]),
                                                        _ptr_array([us_[j] for j in live]), _ptr_array([vs_[j] for j in live]), sig_ptrs,
                                                        _ptr_array(dws), _int_array(accs), Ma, Ka, _p(ws), ws.numel() * 4, _stream()),
                       "mstg_spectral_norm_group_bwd")
        return (None, None, None, *grads, *([None] * (2 * n)))


def spectral_norm_group(modules) -> None:
    """Normalise the weights of `modules` (each carrying the hook of install_fused_spectral_norm) for the forward that follows, all
    in one grouped call; the modules' own hooks then find their weight fresh and do nothing.  Same buffers, same power iteration
    per training-mode forward, same gradients as the per-module hook (enhanced_generator.py:269-271)."""
    lib = _lib.load()
    cap = lib.mstg_spectral_norm_group_max()
    mods = list(modules)
    for k in range(0, len(mods), cap):
        part = mods[k:k + cap]
        names = [m._mstg_sn[0] for m in part]
        eps = part[0]._mstg_sn[1]
        if any(m._mstg_sn[1] != eps or m.training != part[0].training for m in part):
            raise RuntimeError("mstg_hip spectral_norm_group: the modules of a group share eps and the training flag")
        outs = SpectralNormGroupFn.apply(eps, part[0].training, len(part), *[getattr(m, nm + "_orig") for m, nm in zip(part, names)],
                                         *[getattr(m, nm + "_u") for m, nm in zip(part, names)],
                                         *[getattr(m, nm + "_v") for m, nm in zip(part, names)])
        for m, nm, w in zip(part, names, outs):
            setattr(m, nm, w)
            m.__dict__["_mstg_sn_fresh"] = True


# ----------------------------------------------------------------------------------------------------------
# build-defined StructuralTransformerBlock pieces (parity unpinned; see structural_transformer.py)
# ----------------------------------------------------------------------------------------------------------
def structure_map(img: Tensor) -> Tensor:
    """(N,3,H,W) image -> (N, H/4, W/4, 4) structure features; treated as a constant (no gradient)."""
    img = _req(img.detach(), "structure-map image")
    N, _, H, W = img.shape
    out = torch.empty((N, H // 4, W // 4, 4), dtype=torch.float32, device=img.device)
    _lib.check(_lib.load().mstg_structure_map(_p(img), _p(out), N, H, W, _stream()), "mstg_structure_map")
    return out


class LayerNormModFn(torch.autograd.Function):
    """y = (LayerNorm(x; gamma, beta) ) * (1 + gmod[n]) + bmod[n]; x (N, L, dim), gmod / bmod (N, dim) or None."""

    @staticmethod
    def forward(ctx, x, gamma, beta, gmod, bmod, eps):
        lib = _lib.load()
        x, gamma, beta = _req(x, "layer-norm input"), _req(gamma, "layer-norm weight"), _req(beta, "layer-norm bias")
        gmod = None if gmod is None else _req(gmod, "modulation scale")
        bmod = None if bmod is None else _req(bmod, "modulation shift")
        N, L, dim = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((N * L, 2), dtype=torch.float32, device=x.device)
        _timed("ln_mod_fwd_kernel", 0, 4.0 * 2 * x.numel(), lambda: _lib.check(
            lib.mstg_ln_mod_fwd(_p(x), _p(gamma), _p(beta), _p(gmod), _p(bmod), _p(y), _p(stats), N, L, dim, float(eps), _stream()),
            "mstg_ln_mod_fwd"))
        ctx.prefs = (gamma, beta)
        ctx.has_mod = gmod is not None
        ctx.save_for_backward(x, stats, gamma, beta, gmod)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, stats, gamma, beta, gmod = ctx.saved_tensors
        dy = _req(dy, "layer-norm grad_output")
        N, L, dim = x.shape
        dx = torch.empty_like(x)
        sg, sb = _grad_slot(ctx.prefs[0]), _grad_slot(ctx.prefs[1])
        direct = sg is not None and sb is not None
        dgamma = sg if direct else torch.empty_like(gamma)
        dbeta = sb if direct else torch.empty_like(beta)
        dgm = torch.empty((N, dim), dtype=torch.float32, device=x.device) if ctx.has_mod else None
        dbm = torch.empty((N, dim), dtype=torch.float32, device=x.device) if ctx.has_mod else None
        ws = _ws(lib.mstg_ln_mod_bwd_workspace_bytes(N, L, dim), x.device)
        _timed("ln_mod_bwd_kernel", 0, 4.0 * 3 * x.numel(), lambda: _lib.check(
            lib.mstg_ln_mod_bwd(_p(x), _p(stats), _p(gamma), _p(beta), _p(gmod), _p(dy), _p(dx), _p(dgamma), _p(dbeta), _p(dgm), _p(dbm),
                                int(direct), N, L, dim, _p(ws), ws.numel() * 4, _stream()), "mstg_ln_mod_bwd"))
        return dx, (None if direct else dgamma), (None if direct else dbeta), dgm, dbm, None


def layer_norm_mod(x, gamma, beta, gmod=None, bmod=None, eps=1e-5):
    return LayerNormModFn.apply(x, gamma, beta, gmod, bmod, eps)


class FlashAttnFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(D)) v over all tokens of an image, per head; qkv (N, L, 3*heads*D) -> (N, L, heads*D)."""

    @staticmethod
    def forward(ctx, qkv, heads):
        qkv = _req(qkv, "attention qkv")
        N, L, C3 = qkv.shape
        dim = C3 // 3
        D = dim // heads
        if dim * 3 != C3 or D * heads != dim:
            raise RuntimeError(f"mstg_hip flash attention: {C3} channels do not split into q|k|v x {heads} heads")
        out = torch.empty((N, L, dim), dtype=torch.float32, device=qkv.device)
        lse = torch.empty((N, heads, L), dtype=torch.float32, device=qkv.device)
        _timed(f"flash_fwd_kernel<{D}>", 4.0 * N * heads * L * L * D, 4.0 * (qkv.numel() + out.numel()), lambda: _lib.check(
            _lib.load().mstg_flash_attn_fwd(_p(qkv), _p(out), _p(lse), N, L, heads, D, _stream()), "mstg_flash_attn_fwd"),
            f"N{N} L{L} heads{heads} D{D}")
        ctx.heads = heads
        ctx.save_for_backward(qkv, out, lse)
        return out

    @staticmethod
    def backward(ctx, d_out):
        qkv, out, lse = ctx.saved_tensors
        d_out = _req(d_out, "attention grad_output")
        N, L, C3 = qkv.shape
        heads = ctx.heads
        D = C3 // 3 // heads
        dqkv = torch.empty_like(qkv)
        delta = torch.empty((N, heads, L), dtype=torch.float32, device=qkv.device)
        _timed(f"flash_bwd_dkv_kernel<{D}>", 8.0 * N * heads * L * L * D, 4.0 * (2 * qkv.numel() + 2 * out.numel()), lambda: _lib.check(
            _lib.load().mstg_flash_attn_bwd(_p(qkv), _p(out), _p(lse), _p(d_out), _p(dqkv), _p(delta), N, L, heads, D, _stream()),
            "mstg_flash_attn_bwd"), f"N{N} L{L} heads{heads} D{D}",
            split={f"flash_bwd_dq_kernel<{D}>": (4.0 * N * heads * L * L * D, 4.0 * (qkv.numel() + out.numel())),
                   f"flash_bwd_dkv_kernel<{D}>": (4.0 * N * heads * L * L * D, 4.0 * (qkv.numel() + out.numel()))})
        return dqkv, None


def flash_attention(qkv, heads):
    return FlashAttnFn.apply(qkv, heads)


class AddFn(torch.autograd.Function):
    """y = a + b on the HIP element-wise kernel (same shapes)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _req(a, "add lhs"), _req(b, "add rhs")
        if a.shape != b.shape:
            raise RuntimeError(f"mstg_hip add: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
        y = torch.empty_like(a)
        _lib.check(_lib.load().mstg_add(_p(a), _p(b), _p(y), a.numel(), _stream()), "mstg_add")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return AddFn.apply(a, b)


def linear_tokens(x, weight, bias, act=ACT_NONE):
    """nn.Linear over the last dimension of (N, L, Cin) tokens (or (N, Cin) vectors) = a 1x1 convolution on an NHWC view: runs on
    the implicit-GEMM kernels with their weight / bias gradients; `weight` (Cout, Cin) as nn.Linear stores it."""
    squeeze = x.dim() == 2
    x4 = x.reshape(x.shape[0], 1, -1 if not squeeze else 1, x.shape[-1]) if not squeeze else x.reshape(x.shape[0], 1, 1, x.shape[-1])
    y = conv2d(x4, weight.view(weight.shape[0], weight.shape[1], 1, 1), bias, 1, act=act)
    return y.reshape(*x.shape[:-1], weight.shape[0])


def adam_step_flat(p, g, m, v, lr, beta1, beta2, eps, step, mask=None):
    bump_pack_epoch()  # parameter memory is about to change behind torch's version counters
    _lib.check(_lib.load().mstg_adam_step_flat(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, int(step), _p(mask),
                                               _stream()), "mstg_adam_step_flat")
