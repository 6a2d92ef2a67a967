"""Fused Adam / AdamW over the engine's flat buffers, behind the torch.optim.Optimizer interface.

The reference builds ``torch.optim.Adam(params, lr)`` / ``AdamW`` with two parameter groups
``[{"params": backbone, "lr": 0, "name": "backbone"}, {"params": head, "name": "head"}]``
(lightning_pose/models/base.py:458-479, models/heatmap_tracker.py:193-205); ``UnfreezeBackbone`` then rewrites
``param_groups[0]["lr"]`` every batch (callbacks.py:126-148) and ``MultiStepLR`` scales both groups per epoch.
This class keeps exactly that surface (``param_groups``, ``step``, ``zero_grad``, ``state_dict``) but a step is one
``lp_adam_step`` launch per group over a contiguous fp32 range, which also emits the bf16 operand copy; the
transposed data-gradient copies are refreshed only for groups whose lr is non-zero.
"""

from __future__ import annotations

import torch

from . import _lib
from ._lib import check
from . import ops
from .ops import _p


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, engine, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 decoupled_weight_decay: bool = False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=decoupled_weight_decay)
        super().__init__(params, defaults)
        self.engine = engine
        n = engine.plan.n_total
        self.exp_avg = torch.zeros(n, device=engine.device, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=engine.device, dtype=torch.float32)
        # group -> flat range, by the "name" key the reference's callbacks rely on
        self._ranges = {"backbone": (0, engine.plan.n_backbone), "head": (engine.plan.n_backbone, n)}
        for g in self.param_groups:
            if g.get("name") not in self._ranges:
                raise ValueError('FusedAdam expects the reference\'s parameter groups named "backbone" and "head"')
            g.setdefault("step", 0)
        self.grad_scale = 1.0  # e.g. 1 / world_size after a SUM all-reduce

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        e = self.engine
        lib = _lib.lib()
        for g in self.param_groups:
            lo, hi = self._ranges[g["name"]]
            b1, b2 = g["betas"]
            g["step"] += 1
            check(lib.lp_adam_step(_p(e.P[lo:hi]), _p(e.G[lo:hi]), _p(self.exp_avg[lo:hi]), _p(self.exp_avg_sq[lo:hi]), hi - lo,
                                   float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]),
                                   int(bool(g["decoupled"])), int(g["step"]), float(self.grad_scale), _p(e.Wb[lo:hi]), ops._stream()),
                  "lp_adam_step")
            if float(g["lr"]) != 0.0 or (float(g["weight_decay"]) != 0.0 and g["decoupled"]):
                e.refresh_dgrad_copies(lo, hi)
        if hasattr(e, "invalidate_inference_copies"):
            e.invalidate_inference_copies()
        return loss

    def zero_grad(self, set_to_none: bool = False):  # gradients are views of one flat buffer: zero it in one memset
        self.engine.zero_grad()

    def state_dict(self):
        sd = super().state_dict()
        sd["lp_flat_state"] = {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}
        return sd

    def load_state_dict(self, state_dict):
        flat = state_dict.pop("lp_flat_state", None)
        super().load_state_dict(state_dict)
        if flat is not None:
            self.exp_avg.copy_(flat["exp_avg"])
            self.exp_avg_sq.copy_(flat["exp_avg_sq"])
