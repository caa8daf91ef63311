"""CPU restatement of the reference hot path -- TEST INFRASTRUCTURE ONLY.

This file is the parity oracle for the MI355X build.  It restates, in plain fp32
``torch`` functional ops over a flat ``{key: tensor}`` state dict, what the
reference computes on its hot path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; the product package
(``multi-style-transfer-gan_amd/``) never does and fails loudly without its HIP
library.

Pinned (see ``oracle/make_golden.py`` and ``tests/golden/``): every function below
was checked against the reference's own modules imported from ``/root/reference``
in the build container (<=1e-5), and the resulting vectors are committed as
fixtures so the check can be repeated without the reference.

Not pinned ("parity unpinned"): ``gram_matrix`` / ``vgg_features`` /
``multi_style_gram_loss`` / ``structural_transformer_block`` / ``structure_map`` -- the reference has no
implementation of them; they restate the build's own definition.

Reference sites (relative to /root/reference):
  local_attention        enhanced_generator.py:13-47
  multi_scale_block      enhanced_generator.py:78-84  (ctor :50-76)
  generator_forward      enhanced_generator.py:210-228 (stages :91-139)
  discriminator_forward  enhanced_generator.py:231-274 (+ torch spectral_norm hook)
  plain_generator_forward pretrain.py:60-97
  cyclegan_losses / train_step   enhanced_train.py:59-131, hyper-params :36-57
  adam_step              torch.optim.Adam as configured at enhanced_train.py:36-43
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

IN_EPS = 1e-5  # nn.InstanceNorm2d / nn.BatchNorm2d default eps
SN_EPS = 1e-12  # torch.nn.utils.spectral_norm default eps


# --------------------------------------------------------------------------------------
# state-dict specifications and deterministic weights
# --------------------------------------------------------------------------------------
def generator_spec(C: int) -> List[Tuple[str, Tuple[int, ...]]]:
    """Key names/shapes of EnhancedGenerator(C, num_transformer_blocks=0).state_dict()."""
    spec: List[Tuple[str, Tuple[int, ...]]] = [("initial.0.weight", (C, 3, 7, 7)), ("initial.0.bias", (C,))]

    def stage(name, cin, ch, transpose):
        w = (cin, ch, 4, 4) if transpose else (ch, cin, 4, 4)
        out = [(f"{name}.0.weight", w), (f"{name}.0.bias", (ch,)),
               (f"{name}.3.qkv.weight", (3 * ch, ch, 1, 1)), (f"{name}.3.qkv.bias", (3 * ch,)),
               (f"{name}.3.proj.weight", (ch, ch, 1, 1)), (f"{name}.3.proj.bias", (ch,)),
               (f"{name}.4.branch1.0.weight", (ch // 4, ch, 1, 1)), (f"{name}.4.branch1.0.bias", (ch // 4,))]
        for b in (2, 3, 4):
            out += [(f"{name}.4.branch{b}.0.weight", (ch // 4, ch, 3, 3)), (f"{name}.4.branch{b}.0.bias", (ch // 4,))]
        out += [(f"{name}.4.fusion.0.weight", (ch, ch, 1, 1)), (f"{name}.4.fusion.0.bias", (ch,))]
        return out

    spec += stage("down1", C, 2 * C, False)
    spec += stage("down2", 2 * C, 4 * C, False)
    spec += stage("up1", 4 * C, 2 * C, True)
    spec += stage("up2", 2 * C, C, True)
    spec += [("output.0.weight", (3, C, 7, 7)), ("output.0.bias", (3,)),
             ("style_encoder.2.weight", (4 * C, 4 * C)), ("style_encoder.2.bias", (4 * C,))]
    return spec


def discriminator_conv_shapes(C: int) -> List[Tuple[str, Tuple[int, int, int, int]]]:
    return [("main.0", (C, 3, 4, 4)), ("main.2", (2 * C, C, 4, 4)), ("main.5", (4 * C, 2 * C, 4, 4)),
            ("main.8", (8 * C, 4 * C, 4, 4)), ("batch_head.0", (1, 8 * C, 4, 4)),
            ("structure_head.0", (8 * C, 8 * C, 3, 3)), ("structure_head.3", (1, 8 * C, 4, 4))]


def discriminator_spec(C: int) -> List[Tuple[str, Tuple[int, ...]]]:
    """Key names/shapes of EnhancedDiscriminator(C).state_dict() (spectral-norm triplets)."""
    spec = []
    for name, w in discriminator_conv_shapes(C):
        spec += [(f"{name}.bias", (w[0],)), (f"{name}.weight_orig", w),
                 (f"{name}.weight_u", (w[0],)), (f"{name}.weight_v", (w[1] * w[2] * w[3],))]
    return spec


def plain_generator_spec(C: int) -> List[Tuple[str, Tuple[int, ...]]]:
    """Key names/shapes of the plain Generator(C).state_dict() (pretrain.py:65-92)."""
    spec = []
    enc = [(0, 3, C), (2, C, 2 * C), (5, 2 * C, 4 * C), (8, 4 * C, 8 * C)]
    for idx, ci, co in enc:
        spec += [(f"encoder.{idx}.weight", (co, ci, 4, 4)), (f"encoder.{idx}.bias", (co,))]
    for idx, ch in [(3, 2 * C), (6, 4 * C), (9, 8 * C)]:
        spec += _bn_spec(f"encoder.{idx}", ch)
    dec = [(0, 8 * C, 4 * C), (3, 4 * C, 2 * C), (6, 2 * C, C), (9, C, 3)]
    for idx, ci, co in dec:
        spec += [(f"decoder.{idx}.weight", (ci, co, 4, 4)), (f"decoder.{idx}.bias", (co,))]
    for idx, ch in [(1, 4 * C), (4, 2 * C), (7, C)]:
        spec += _bn_spec(f"decoder.{idx}", ch)
    return spec


def _bn_spec(prefix, ch):
    return [(f"{prefix}.weight", (ch,)), (f"{prefix}.bias", (ch,)), (f"{prefix}.running_mean", (ch,)),
            (f"{prefix}.running_var", (ch,)), (f"{prefix}.num_batches_tracked", ())]


def make_state_dict(spec, seed: int, gain: float = 1.0) -> SD:
    """Deterministic, platform-independent weights (numpy RandomState), fp32.

    Conv / linear weights ~ N(0, gain^2 * 2 / fan_in) ; biases ~ 0.1 * N(0,1);
    spectral-norm u/v are unit vectors; BatchNorm weight ~ 1 + 0.1 N, running_var in [0.5, 1.5].
    """
    rs = np.random.RandomState(seed)
    sd: SD = {}
    for key, shape in spec:
        if key.endswith("num_batches_tracked"):
            sd[key] = torch.zeros((), dtype=torch.int64)
            continue
        a = rs.standard_normal(size=shape).astype(np.float32) if shape else np.float32(rs.standard_normal())
        if key.endswith(("weight_u", "weight_v")):
            a = a / max(float(np.linalg.norm(a)), 1e-12)
        elif key.endswith("running_var"):
            a = (0.5 + rs.random_sample(size=shape)).astype(np.float32)
        elif key.endswith("running_mean"):
            a = 0.1 * a
        elif len(shape) >= 2:
            a = a * np.float32(gain * math.sqrt(2.0 / int(np.prod(shape[1:]))))
        elif key.endswith("bias"):
            a = 0.1 * a
        elif key.endswith("weight"):  # 1-D weight = BatchNorm gamma
            a = 1.0 + 0.1 * a
        sd[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return sd


def make_input(shape, seed: int) -> Tensor:
    """Uniform [-1, 1) fp32 input, numpy-seeded (platform independent)."""
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.random_sample(size=shape) * 2.0 - 1.0).astype(np.float32))


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def instance_norm(x: Tensor) -> Tensor:
    """nn.InstanceNorm2d defaults: affine=False, biased variance, eps 1e-5, no running stats."""
    mu = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (x - mu) * torch.rsqrt(var + IN_EPS)


def local_attention(x: Tensor, sd: SD, p: str, ws: int = 4) -> Tensor:
    """enhanced_generator.py:13-47 restated on NCHW without the view/permute dance.

    Per ws x ws window: qkv 1x1; L2-normalise q and k per pixel over channels
    (F.normalize eps 1e-12); attn = softmax_c2( sum_pixels qn[c1,p] kn[c2,p] );
    out[c1,p] = sum_c2 attn[c1,c2] v[c2,p]; proj 1x1.  H, W must be multiples of ws
    (the reference's padding branch raises for anything else).
    """
    B, C, H, W = x.shape
    if H % ws or W % ws:
        raise RuntimeError(f"LocalAttention needs H, W multiples of {ws}, got {H}x{W}")
    qkv = F.conv2d(x, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    q, k, v = qkv.chunk(3, dim=1)
    q = q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)
    k = k / k.norm(dim=1, keepdim=True).clamp_min(1e-12)

    def win(t):  # (B,C,H,W) -> (B, nH, nW, C, ws*ws)
        t = t.reshape(B, C, H // ws, ws, W // ws, ws).permute(0, 2, 4, 1, 3, 5)
        return t.reshape(B, H // ws, W // ws, C, ws * ws)

    qw, kw, vw = win(q), win(k), win(v)
    attn = torch.einsum("bhwcp,bhwdp->bhwcd", qw, kw).softmax(dim=-1)
    ow = torch.einsum("bhwcd,bhwdp->bhwcp", attn, vw)
    o = ow.reshape(B, H // ws, W // ws, C, ws, ws).permute(0, 3, 1, 4, 2, 5).reshape(B, C, H, W)
    return F.conv2d(o, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def multi_scale_block(x: Tensor, sd: SD, p: str) -> Tensor:
    """enhanced_generator.py:78-84: 1x1 + three dilated 3x3 branches, IN+ReLU each, cat, fusion, +x."""
    outs = [F.relu(instance_norm(F.conv2d(x, sd[p + ".branch1.0.weight"], sd[p + ".branch1.0.bias"])))]
    for b, d in ((2, 1), (3, 2), (4, 4)):
        y = F.conv2d(x, sd[p + f".branch{b}.0.weight"], sd[p + f".branch{b}.0.bias"], padding=d, dilation=d)
        outs.append(F.relu(instance_norm(y)))
    cat = torch.cat(outs, dim=1)
    f = F.relu(instance_norm(F.conv2d(cat, sd[p + ".fusion.0.weight"], sd[p + ".fusion.0.bias"])))
    return f + x


def _stage(x: Tensor, sd: SD, name: str, transpose: bool) -> Tensor:
    if transpose:
        x = F.conv_transpose2d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], stride=2, padding=1)
    else:
        x = F.conv2d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], stride=2, padding=1)
    x = F.relu(instance_norm(x))
    x = local_attention(x, sd, name + ".3", 4)
    return multi_scale_block(x, sd, name + ".4")


def generator_forward(sd: SD, x: Tensor, taps: Optional[dict] = None, num_blocks: int = 0, num_heads: int = 4) -> Tensor:
    """EnhancedGenerator.forward, enhanced_generator.py:210-228.

    ``taps`` (optional dict) receives initial/down1/down2/up1/up2/pre_tanh/out.
    With ``num_blocks`` = 0 (the parity-pinned configuration) the style encoder (:216) is dead code and is skipped; with blocks the
    token path of :216-225 runs through the BUILD-DEFINED ``structural_transformer_block`` (parity unpinned).
    """
    h = F.relu(instance_norm(F.conv2d(x, sd["initial.0.weight"], sd["initial.0.bias"], padding=3)))
    t = {"initial": h}
    h = _stage(h, sd, "down1", False); t["down1"] = h
    h = _stage(h, sd, "down2", False); t["down2"] = h
    if num_blocks:
        style = style_encoder(sd, h)                                  # :216
        B, Cn, H4, W4 = h.shape
        tokens = h.flatten(2).transpose(1, 2)                         # :218-219
        for i in range(num_blocks):
            tokens = structural_transformer_block(sd, f"transformer_blocks.{i}", tokens, style, x, num_heads)   # :222-223
        h = tokens.transpose(1, 2).reshape(B, Cn, H4, W4)             # :225
        t["tokens"] = h
    h = _stage(h, sd, "up1", True); t["up1"] = h
    h = _stage(h, sd, "up2", True); t["up2"] = h
    pre = F.conv2d(h, sd["output.0.weight"], sd["output.0.bias"], padding=3)
    t["pre_tanh"] = pre
    out = torch.tanh(pre)
    t["out"] = out
    if taps is not None:
        taps.update(t)
    return out


def style_encoder(sd: SD, feat: Tensor) -> Tensor:
    """enhanced_generator.py:142-147: global average pool -> Linear -> ReLU."""
    return F.relu(F.linear(feat.mean(dim=(2, 3)), sd["style_encoder.2.weight"], sd["style_encoder.2.bias"]))


# -- spectral norm (torch.nn.utils.spectral_norm, hook flavour, n_power_iterations=1) ----------
def spectral_weight(sd: SD, name: str, train: bool) -> Tensor:
    """W / sigma; in train mode one power iteration updates u, v IN PLACE first (no grad)."""
    w = sd[name + ".weight_orig"]
    u, v = sd[name + ".weight_u"], sd[name + ".weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if train:
        with torch.no_grad():
            vn = F.normalize(torch.mv(wm.t(), u), dim=0, eps=SN_EPS)
            un = F.normalize(torch.mv(wm, vn), dim=0, eps=SN_EPS)
            v.copy_(vn)
            u.copy_(un)
        u, v = u.clone(), v.clone()
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


def discriminator_forward(sd: SD, x: Tensor, train: bool = True) -> Tuple[Tensor, Tensor]:
    """EnhancedDiscriminator.forward, enhanced_generator.py:273-274 -> (score.squeeze(), structure map)."""
    def conv(h, name, stride, pad):
        return F.conv2d(h, spectral_weight(sd, name, train), sd[name + ".bias"], stride=stride, padding=pad)

    h = F.leaky_relu(conv(x, "main.0", 2, 1), 0.2)
    for name in ("main.2", "main.5", "main.8"):
        h = F.leaky_relu(instance_norm(conv(h, name, 2, 1)), 0.2)
    score = conv(h, "batch_head.0", 1, 1).mean(dim=(2, 3), keepdim=True).squeeze()
    s = F.leaky_relu(instance_norm(conv(h, "structure_head.0", 1, 1)), 0.2)
    return score, conv(s, "structure_head.3", 1, 1)


# -- plain CycleGAN generator (BatchNorm) -----------------------------------------------------
def _batch_norm(x: Tensor, sd: SD, p: str, train: bool, momentum: float = 0.1) -> Tensor:
    if train:
        mu = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        with torch.no_grad():
            n = x.numel() / x.shape[1]
            sd[p + ".running_mean"].mul_(1 - momentum).add_(momentum * mu)
            sd[p + ".running_var"].mul_(1 - momentum).add_(momentum * var * n / max(n - 1, 1))
            sd[p + ".num_batches_tracked"] += 1
    else:
        mu, var = sd[p + ".running_mean"], sd[p + ".running_var"]
    xh = (x - mu[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + IN_EPS)
    return xh * sd[p + ".weight"][None, :, None, None] + sd[p + ".bias"][None, :, None, None]


def plain_generator_forward(sd: SD, x: Tensor, train: bool = True) -> Tensor:
    """Generator.forward, pretrain.py:94-97 (encoder :65-77, decoder :80-92)."""
    h = F.leaky_relu(F.conv2d(x, sd["encoder.0.weight"], sd["encoder.0.bias"], stride=2, padding=1), 0.2)
    for ci, bi in ((2, 3), (5, 6), (8, 9)):
        h = F.conv2d(h, sd[f"encoder.{ci}.weight"], sd[f"encoder.{ci}.bias"], stride=2, padding=1)
        h = F.leaky_relu(_batch_norm(h, sd, f"encoder.{bi}", train), 0.2)
    for ci, bi in ((0, 1), (3, 4), (6, 7)):
        h = F.conv_transpose2d(h, sd[f"decoder.{ci}.weight"], sd[f"decoder.{ci}.bias"], stride=2, padding=1)
        h = F.relu(_batch_norm(h, sd, f"decoder.{bi}", train))
    h = F.conv_transpose2d(h, sd["decoder.9.weight"], sd["decoder.9.bias"], stride=2, padding=1)
    return torch.tanh(h)


# --------------------------------------------------------------------------------------
# training step (enhanced_train.py:59-131), restated functionally
# --------------------------------------------------------------------------------------
LAMBDA_CYCLE, LAMBDA_IDENTITY, LAMBDA_STRUCTURE = 10.0, 2.0, 0.5  # enhanced_train.py:55-57
G_LR, D_LR, BETAS, ADAM_EPS = 5e-5, 2e-4, (0.5, 0.999), 1e-8  # enhanced_train.py:36-43


def mse(a: Tensor, target: float) -> Tensor:
    return ((a - target) ** 2).mean()


def l1(a: Tensor, b: Tensor) -> Tensor:
    return (a - b).abs().mean()


class AdamState:
    """torch.optim.Adam (no amsgrad, no weight decay) over a list of tensors; skips grads that are None."""

    def __init__(self, params: List[Tensor], lr: float):
        self.params, self.lr, self.t = params, lr, [0] * len(params)
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    @torch.no_grad()
    def step(self, grads: List[Optional[Tensor]]):
        b1, b2 = BETAS
        for i, (p, g) in enumerate(zip(self.params, grads)):
            if g is None:
                continue
            self.t[i] += 1
            self.m[i].mul_(b1).add_(g, alpha=1 - b1)
            self.v[i].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1, bc2 = 1 - b1 ** self.t[i], 1 - b2 ** self.t[i]
            denom = (self.v[i].sqrt() / math.sqrt(bc2)).add_(ADAM_EPS)
            p.addcdiv_(self.m[i], denom, value=-self.lr / bc1)


class CycleGANOracle:
    """Functional mirror of EnhancedCycleGAN (enhanced_train.py:13-131) for blocks=0, fp32, CPU.

    Holds four state dicts; trainable tensors have requires_grad=True.  ``train_step`` follows the
    reference order exactly: 2 G forwards, D update (4 D forwards), G update (4 more G forwards,
    6 D forwards), Adam steps, returns the five python floats.
    """

    def __init__(self, g_ab: SD, g_ba: SD, d_a: SD, d_b: SD):
        self.G_AB, self.G_BA, self.D_A, self.D_B = g_ab, g_ba, d_a, d_b
        self.g_keys = [(sd, k) for sd in (g_ab, g_ba) for k in sd]
        self.d_keys = [(sd, k) for sd in (d_a, d_b) for k in sd if not k.endswith(("weight_u", "weight_v"))]
        for sd, k in self.g_keys + self.d_keys:
            sd[k].requires_grad_(True)
        self.g_opt = AdamState([sd[k] for sd, k in self.g_keys], G_LR)
        self.d_opt = AdamState([sd[k] for sd, k in self.d_keys], D_LR)

    def train_step(self, real_A: Tensor, real_B: Tensor) -> Dict[str, float]:
        G_AB, G_BA, D_A, D_B = self.G_AB, self.G_BA, self.D_A, self.D_B
        fake_B = generator_forward(G_AB, real_A)
        fake_A = generator_forward(G_BA, real_B)
        # --- discriminator update (:67-85)
        ra, _ = discriminator_forward(D_A, real_A)
        rb, _ = discriminator_forward(D_B, real_B)
        d_real = (mse(ra, 1.0) + mse(rb, 1.0)) * 0.5
        fa, _ = discriminator_forward(D_A, fake_A.detach())
        fb, _ = discriminator_forward(D_B, fake_B.detach())
        d_fake = (mse(fa, 0.0) + mse(fb, 0.0)) * 0.5
        d_loss = d_real + d_fake
        d_params = [sd[k] for sd, k in self.d_keys]
        self.d_opt.step(list(torch.autograd.grad(d_loss, d_params, allow_unused=True)))
        # --- generator update (:88-123)
        idt_A = generator_forward(G_BA, real_A)
        idt_B = generator_forward(G_AB, real_B)
        identity = (l1(idt_A, real_A) + l1(idt_B, real_B)) * LAMBDA_IDENTITY
        fa, _ = discriminator_forward(D_A, fake_A)
        fb, _ = discriminator_forward(D_B, fake_B)
        g_loss = mse(fa, 1.0) + mse(fb, 1.0)
        recon_A = generator_forward(G_BA, fake_B)
        recon_B = generator_forward(G_AB, fake_A)
        cycle = (l1(recon_A, real_A) + l1(recon_B, real_B)) * LAMBDA_CYCLE
        _, ras = discriminator_forward(D_A, real_A)
        _, fas = discriminator_forward(D_A, fake_A)
        _, rbs = discriminator_forward(D_B, real_B)
        _, fbs = discriminator_forward(D_B, fake_B)
        structure = (l1(ras, fas) + l1(rbs, fbs)) * LAMBDA_STRUCTURE
        total = g_loss + cycle + identity + structure
        g_params = [sd[k] for sd, k in self.g_keys]
        self.g_opt.step(list(torch.autograd.grad(total, g_params, allow_unused=True)))
        return {"d_loss": d_loss.item(), "g_loss": g_loss.item(), "cycle_loss": cycle.item(),
                "identity_loss": identity.item(), "structure_loss": structure.item()}


# --------------------------------------------------------------------------------------
# build-defined extensions -- PARITY UNPINNED (no reference implementation)
# --------------------------------------------------------------------------------------
def transformer_block_spec(dim: int, prefix: str = "transformer_blocks.0", mlp_ratio: int = 2):
    """state_dict keys/shapes of the build-defined StructuralTransformerBlock(dim) (structural_transformer.py)."""
    return [(f"{prefix}.struct_proj.weight", (dim, 4)), (f"{prefix}.struct_proj.bias", (dim,)),
            (f"{prefix}.style_mod.weight", (2 * dim, dim)), (f"{prefix}.style_mod.bias", (2 * dim,)),
            (f"{prefix}.norm1.weight", (dim,)), (f"{prefix}.norm1.bias", (dim,)),
            (f"{prefix}.qkv.weight", (3 * dim, dim)), (f"{prefix}.qkv.bias", (3 * dim,)),
            (f"{prefix}.proj.weight", (dim, dim)), (f"{prefix}.proj.bias", (dim,)),
            (f"{prefix}.norm2.weight", (dim,)), (f"{prefix}.norm2.bias", (dim,)),
            (f"{prefix}.fc1.weight", (mlp_ratio * dim, dim)), (f"{prefix}.fc1.bias", (mlp_ratio * dim,)),
            (f"{prefix}.fc2.weight", (dim, mlp_ratio * dim)), (f"{prefix}.fc2.bias", (dim,))]


def generator_spec_with_blocks(C: int, num_blocks: int = 1):
    """EnhancedGenerator(C, num_transformer_blocks=num_blocks).state_dict() key order (blocks sit after down2 in module order)."""
    spec = generator_spec(C)
    idx = next(i for i, (k, _) in enumerate(spec) if k.startswith("up1."))
    blocks = []
    for i in range(num_blocks):
        blocks += transformer_block_spec(4 * C, f"transformer_blocks.{i}")
    return spec[:idx] + blocks + spec[idx:]


def structure_map(img: Tensor) -> Tensor:
    """(N,3,H,W) -> (N, L, 4): per 4x4 cell mean R, G, B and mean of |dx| + |dy| of the luminance (forward differences, zero
    across the image border)."""
    lum = 0.299 * img[:, 0] + 0.587 * img[:, 1] + 0.114 * img[:, 2]
    gx = torch.zeros_like(lum)
    gy = torch.zeros_like(lum)
    gx[:, :, :-1] = lum[:, :, 1:] - lum[:, :, :-1]
    gy[:, :-1, :] = lum[:, 1:, :] - lum[:, :-1, :]
    feats = torch.cat([img, (gx.abs() + gy.abs()).unsqueeze(1)], dim=1)
    return F.avg_pool2d(feats, 4).flatten(2).transpose(1, 2)


def structural_transformer_block(sd: SD, p: str, x: Tensor, style: Tensor, orig: Tensor, num_heads: int = 4) -> Tensor:
    """The build's definition of StructuralTransformerBlock.forward (structural_transformer.py docstring), plain torch ops."""
    N, L, dim = x.shape
    d = dim // num_heads
    with torch.no_grad():
        s = structure_map(orig)
    h = x + F.linear(s.to(x.dtype), sd[p + ".struct_proj.weight"], sd[p + ".struct_proj.bias"])
    g, b = F.linear(style, sd[p + ".style_mod.weight"], sd[p + ".style_mod.bias"]).chunk(2, dim=-1)
    u = F.layer_norm(h, (dim,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5) * (1 + g[:, None, :]) + b[:, None, :]
    qkv = F.linear(u, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    q, k, v = (t.reshape(N, L, num_heads, d).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1) @ v
    h = h + F.linear(a.transpose(1, 2).reshape(N, L, dim), sd[p + ".proj.weight"], sd[p + ".proj.bias"])
    m = F.gelu(F.linear(F.layer_norm(h, (dim,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5), sd[p + ".fc1.weight"], sd[p + ".fc1.bias"]))
    return h + F.linear(m, sd[p + ".fc2.weight"], sd[p + ".fc2.bias"])


VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512]  # VGG16 up to relu4_3
VGG_TAPS = (1, 3, 6, 9)  # conv indices whose ReLU output is tapped: relu1_2, relu2_2, relu3_3, relu4_3


def vgg_spec(width_div: int = 1):
    spec, cin, i = [], 3, 0
    for c in VGG_CFG:
        if c == "M":
            continue
        co = c // width_div
        spec += [(f"conv{i}.weight", (co, cin, 3, 3)), (f"conv{i}.bias", (co,))]
        cin, i = co, i + 1
    return spec


def vgg_features(sd: SD, x: Tensor) -> List[Tensor]:
    """Frozen VGG16-topology stack (3x3 conv + ReLU, 2x2 max-pool), taps after relu1_2/2_2/3_3/4_3."""
    feats, i, h = [], 0, x
    for c in VGG_CFG:
        if c == "M":
            h = F.max_pool2d(h, 2)
            continue
        h = F.relu(F.conv2d(h, sd[f"conv{i}.weight"], sd[f"conv{i}.bias"], padding=1))
        if i in VGG_TAPS:
            feats.append(h)
        i += 1
    return feats


def gram_matrix(f: Tensor) -> Tensor:
    """G = F F^T / (C H W), F = features reshaped (N, C, H*W)."""
    n, c, h, w = f.shape
    m = f.reshape(n, c, h * w)
    return torch.bmm(m, m.transpose(1, 2)) / (c * h * w)


def multi_style_gram_loss(sd: SD, y: Tensor, styles: List[Tensor], weights: List[float]) -> Tensor:
    """sum_l MSE(G_l(y), sum_k w_k G_l(s_k)); style Grams are averaged over the style batch dim."""
    fy = vgg_features(sd, y)
    fs = [vgg_features(sd, s) for s in styles]
    loss = y.new_zeros(())
    for l, f in enumerate(fy):
        target = sum(w * gram_matrix(fk[l]).mean(dim=0, keepdim=True) for w, fk in zip(weights, fs))
        loss = loss + ((gram_matrix(f) - target) ** 2).mean()
    return loss
