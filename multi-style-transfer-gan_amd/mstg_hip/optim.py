"""Flat-buffer Adam: one fused HIP launch per optimizer step.

All parameters of an optimizer are re-homed into ONE contiguous fp32 buffer (``flat``) and their gradients into
another (``grad``): ``p.data`` / ``p.grad`` become views.  The step is a single ``mstg_adam_step_flat`` launch over
the buffer, and the same gradient buffer is what data-parallel training all-reduces (one RCCL collective per
optimizer, see ``dp.py``).  Numerics follow torch.optim.Adam as configured at enhanced_train.py:36-43
(no weight decay, no amsgrad).  A parameter whose gradient stays zero is left untouched, which is what
torch.optim.Adam does for ``grad is None`` parameters (style_encoder when there are no transformer blocks).
"""
from __future__ import annotations

from typing import Iterable, List

import torch

from . import ops


class FlatAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        if not self.params:
            raise ValueError("FlatAdam: empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam: parameters must live on the GPU (move the model before building the optimizer)")
        self.param_groups = [{"params": self.params, "lr": lr, "betas": tuple(betas), "eps": eps}]
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4  # keep every view 16-byte aligned
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self.step_count = 0
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p._mstg_flat = self.flat  # ops' filter-pack cache watches this buffer's version counter too
        self._attach_grads()

    def _attach_grads(self):
        for p, o in zip(self.params, self.offsets):
            g = self.grad[o:o + p.numel()].view_as(p)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def zero_grad(self, set_to_none: bool = True):  # signature of torch.optim.Optimizer.zero_grad
        self.grad.zero_()
        self._attach_grads()

    def check_views(self):
        """Every parameter and its gradient must still be the views into the flat buffers this optimizer created: a later
        ``module.half()`` / ``.to(device)`` / ``module.zero_grad(set_to_none=True)`` silently re-homes them, after which the fused
        step would update stale copies."""
        base_p, base_g = self.flat.data_ptr(), self.grad.data_ptr()
        for p, o in zip(self.params, self.offsets):
            if p.data_ptr() != base_p + 4 * o:
                raise RuntimeError("FlatAdam: a parameter no longer lives in the optimizer's flat buffer (was the module moved or "
                                   "cast after the optimizer was built?)")
            if p.grad is None or p.grad.data_ptr() != base_g + 4 * o:
                raise RuntimeError("FlatAdam: a parameter's .grad is not the optimizer's flat-gradient view; use "
                                   "optimizer.zero_grad(), not module.zero_grad(set_to_none=True)")

    @torch.no_grad()
    def step(self):
        self.check_views()
        g = self.param_groups[0]
        self.step_count += 1
        ops.adam_step_flat(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                           self.step_count)

    def state_dict(self):
        """torch.optim.Adam's layout (what pretrain.py:210-216 saves and pretrain_resume.py:146-147 loads into optim.Adam): per-parameter
        ``state[i] = {step, exp_avg, exp_avg_sq}`` sliced out of the flat buffers, ``param_groups`` with parameter indices.  As
        with torch, a parameter that never received a gradient (both moments identically zero) has no state entry."""
        g = self.param_groups[0]
        state = {}
        if self.step_count > 0:
            m_all, v_all = self.exp_avg.detach(), self.exp_avg_sq.detach()
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                m, v = m_all[o:o + p.numel()], v_all[o:o + p.numel()]
                if bool((m != 0).any()) or bool((v != 0).any()):
                    state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m.view_as(p).clone(),
                                "exp_avg_sq": v.view_as(p).clone()}
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    @torch.no_grad()
    def load_state_dict(self, sd):
        """Accepts torch.optim.Adam's state_dict (a checkpoint the reference wrote, or this class's own) and the flat layout this
        class wrote in round 2 ({step, exp_avg, exp_avg_sq, param_groups})."""
        if "state" not in sd:  # round-2 flat layout
            self.step_count = int(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            self.param_groups[0].update({k: v for k, v in sd["param_groups"][0].items() if k != "params"})
            return
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError("FlatAdam.load_state_dict: expected one parameter group with "
                             f"{len(self.params)} parameters, got {[len(g_['params']) for g_ in groups]}")
        if groups[0].get("weight_decay", 0) or groups[0].get("amsgrad", False) or groups[0].get("maximize", False):
            raise ValueError("FlatAdam.load_state_dict: weight_decay / amsgrad / maximize are not implemented (the reference uses none)")
        index_of = {pid: i for i, pid in enumerate(groups[0]["params"])}
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = set()
        for pid, st in sd["state"].items():
            i = index_of[pid]
            p, o = self.params[i], self.offsets[i]
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"FlatAdam.load_state_dict: state {pid} has shape {tuple(st['exp_avg'].shape)}, parameter {tuple(p.shape)}")
            self.exp_avg[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"FlatAdam.load_state_dict: parameters at different step counts {sorted(steps)} (one fused step serves all)")
        self.step_count = steps.pop() if steps else 0
        self.param_groups[0].update({k: groups[0][k] for k in ("lr", "eps") if k in groups[0]})
        if "betas" in groups[0]:
            self.param_groups[0]["betas"] = tuple(groups[0]["betas"])
