"""Reader of tests/golden/train_step_fp64_c8.npz (oracle/make_golden.py::gen_train_step_fp64): the reference's unmodified
train_step run in fp64 for three steps -- losses, every gradient both optimizers saw, the spectral-norm vectors each step
started from -- plus, per step and tensor, the distance of the reference's own fp32 evaluation of the same step.

The fixture stores gradients, not parameters: the state every step starts from (parameters, Adam moments) is re-derived here
in fp64 with torch.optim.Adam's recurrences (enhanced_train.py:36-43: betas (0.5, 0.999), eps 1e-8, no weight decay), which is
what the reference's fp64 run did.  A parameter whose gradient is None (style_encoder with no transformer blocks) never moves."""
import math
import os

import numpy as np
import torch

LOSS_KEYS = ("d_loss", "g_loss", "cycle_loss", "identity_loss", "structure_loss")
BETAS, EPS = (0.5, 0.999), 1e-8
LR = {"g": 5e-5, "d": 2e-4}


def dead_bias(name: str) -> bool:
    """Bias of a convolution that feeds an InstanceNorm: its gradient is exactly zero in exact arithmetic."""
    if not name.endswith(".bias"):
        return False
    stem = name[:-5]
    return (stem == "initial.0" or stem.endswith((".branch1.0", ".branch2.0", ".branch3.0", ".branch4.0", ".fusion.0"))
            or stem in ("down1.0", "down2.0", "up1.0", "up2.0", "main.2", "main.5", "main.8", "structure_head.0"))


class Fp64TrainStepFixture:
    def __init__(self, gold_dir):
        self.g = np.load(os.path.join(gold_dir, "train_step_fp64_c8.npz"))
        g = self.g
        self.C, self.shape = int(g["C"]), tuple(int(v) for v in g["shape"])
        self.seeds, self.in_seed, self.steps = [int(s) for s in g["seeds"]], int(g["in_seed"]), int(g["steps"])
        self.names = {"g": [str(n) for n in g["g_names"]], "d": [str(n) for n in g["d_names"]]}

    def losses64(self, k):
        return dict(zip(LOSS_KEYS, (float(v) for v in self.g[f"losses64_{k}"])))

    def losses32(self, k):
        return dict(zip(LOSS_KEYS, (float(v) for v in self.g[f"losses32_{k}"])))

    def grads(self, which, k):
        """fp64 gradients (stored rounded to fp32) the `which` optimizer saw at step k; None where the reference has no gradient."""
        out = []
        for i in range(len(self.names[which])):
            key = f"{which}64_{k}_{i}"
            out.append(torch.from_numpy(self.g[key]).double() if key in self.g.files else None)
        return out

    def ref32_dist(self, which, k):
        """(per-tensor distances of the reference's own teacher-forced fp32 run from fp64 [-1 = no gradient], aggregate over live tensors)"""
        return [float(v) for v in self.g[f"{which}32_dist_{k}"]], float(self.g[f"{which}32_agg_{k}"])

    def scan(self, which="g"):
        """the reference's own teacher-forced fp32-vs-fp64 aggregate gradient distances over many draws of this shape (sorted)"""
        return np.sort(np.asarray(self.g[f"scan_{which}32_agg"], dtype=np.float64))

    def uv(self, k):
        """{'D_A.main.0.weight_u': tensor, ...}: spectral-norm vectors step k starts from"""
        pre = f"uv_{k}_"
        return {key[len(pre):]: torch.from_numpy(self.g[key]) for key in self.g.files if key.startswith(pre)}

    def state_at(self, which, k, p0):
        """(params, exp_avg, exp_avg_sq) in fp64 at the START of step k, from the initial parameters p0 (list, optimizer order)."""
        b1, b2 = BETAS
        p = [t.double().clone() for t in p0]
        m = [torch.zeros_like(t) for t in p]
        v = [torch.zeros_like(t) for t in p]
        for t in range(1, k + 1):
            for i, gr in enumerate(self.grads(which, t - 1)):
                if gr is None:
                    continue
                gr = gr.view_as(p[i])
                m[i].mul_(b1).add_(gr, alpha=1 - b1)
                v[i].mul_(b2).addcmul_(gr, gr, value=1 - b2)
                denom = (v[i].sqrt() / math.sqrt(1 - b2 ** t)).add_(EPS)
                p[i].addcdiv_(m[i], denom, value=-LR[which] / (1 - b1 ** t))
        return p, m, v
