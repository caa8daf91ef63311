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
        return {"step": self.step_count, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups[0].update(sd["param_groups"][0])
