"""FusedSGD: torch.optim.SGD(momentum, weight_decay) semantics (main.py:749) over flat buffers.

One launch per parameter group (an encoder or the head).  A group whose gradient is "None" is
skipped, exactly like torch >= 2 after `zero_grad()` (set_to_none); `legacy_zero_grad=True`
reproduces pinned torch 1.8.1, where zeroed gradients still receive weight decay + momentum
(SURVEY.md Q6).
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import ops


class FusedSGD:
    def __init__(self, groups: Dict[str, object], lr: float = 1e-3, momentum: float = 0.9, weight_decay: float = 1e-4,
                 legacy_zero_grad: bool = False):
        """groups: name -> object with `.flat` and `.grad` (ResNet18Encoder / SharedHead)."""
        self.groups = groups
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.legacy_zero_grad = legacy_zero_grad
        self.buf = {k: torch.zeros_like(g.flat) for k, g in groups.items()}
        self.initialized = {k: False for k in groups}
        # gradient state per group: "none" | "zero" | "ready"
        self.grad_state = {k: "none" for k in groups}

    def mark_ready(self, name: str) -> None:
        self.grad_state[name] = "ready"

    def step(self) -> None:
        for k, g in self.groups.items():
            state = self.grad_state[k]
            if state == "none":
                continue                                   # p.grad is None -> skipped by torch.optim.SGD
            grad = g.grad if state == "ready" else None     # "zero": zeroed grads (1.8.1): wd + momentum still apply
            ops.sgd_step(g.flat, grad, self.buf[k], self.lr, self.momentum, self.weight_decay,
                         first=not self.initialized[k])
            self.initialized[k] = True

    def step_group(self, name: str) -> None:
        g = self.groups[name]
        state = self.grad_state[name]
        if state == "none":
            return
        ops.sgd_step(g.flat, g.grad if state == "ready" else None, self.buf[name], self.lr, self.momentum,
                     self.weight_decay, first=not self.initialized[name])
        self.initialized[name] = True

    def zero_grad(self) -> None:
        for k in self.groups:
            if self.grad_state[k] != "none":
                self.grad_state[k] = "zero" if self.legacy_zero_grad else "none"

    def drop_grads(self) -> None:
        """main.py:468-470: `del p.grad` for every parameter."""
        for k in self.groups:
            self.grad_state[k] = "none"

    def set_lr(self, lr: float) -> None:
        self.lr = lr

    def state_dict(self) -> dict:
        return {"lr": self.lr, "momentum": self.momentum, "weight_decay": self.weight_decay,
                "momentum_buffer": {k: v.clone() for k, v in self.buf.items()},
                "initialized": dict(self.initialized)}

    def load_state_dict(self, sd: dict) -> None:
        self.lr, self.momentum, self.weight_decay = sd["lr"], sd["momentum"], sd["weight_decay"]
        for k, v in sd["momentum_buffer"].items():
            self.buf[k].copy_(v)
        self.initialized = dict(sd["initialized"])
