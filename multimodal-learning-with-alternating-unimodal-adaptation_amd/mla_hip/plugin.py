"""GSPlugin: the head-gradient modification that guards the shared head (utils/utils.py:12-41).

Same surface as the reference: attributes `Pl` (D,D) and `exp_count`, method
`before_update(model, before_batch_input, batch_index, len_dataloader, train_exp_counter)` which
mutates the head's weight gradient and `Pl` in place.

Two parity modes:
  * "as_published": the reference compares parameter names with "module.weight" while it is
    handed the bare nn.Linear (names "weight"/"bias"), so the body never runs -> no-op;
  * "as_intended" (default): executes utils/utils.py:34-41 literally -- element-wise DxD
    denominator `alpha + k r`, Frobenius renormalisation (Q2) -- on the HIP kernels.
D is taken from the head (512 ResNet, 768 M3AE/CAV-MAE) instead of the hard-wired eye(512) (Q3);
the update is graph-free (Q4).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import ops


class GSPlugin:
    def __init__(self, gs_flag: bool = True, dim: int = 512, device="cuda", mode: Optional[str] = None):
        """`GSPlugin()` as at main.py:819.  mode: "as_intended" (default; $MLA_GS_MODE overrides) or "as_published"."""
        mode = mode or os.environ.get("MLA_GS_MODE", "as_intended")
        if mode not in ("as_intended", "as_published"):
            raise ValueError("mode must be 'as_intended' or 'as_published'")
        self.mode = mode
        self.device = torch.device(device)
        self.Pl = torch.eye(dim, device=self.device, dtype=torch.float32)       # utils/utils.py:20
        self.exp_count = 0                                                      # utils/utils.py:21
        self._r = torch.empty(dim, device=self.device, dtype=torch.float32)
        self._ws: Optional[torch.Tensor] = None
        self._fired = False

    @staticmethod
    def alpha(batch_index: int, len_dataloader: int) -> float:
        lamda = batch_index / len_dataloader + 1                                # utils/utils.py:26
        return 1.0 * 0.1 ** lamda                                               # utils/utils.py:27

    def _resize(self, D: int) -> None:
        """The reference hard-wires eye(512) (the 768 variant is commented out, utils/utils.py:19-20): take D from
        the head it is handed, as long as the projector is still the untouched identity."""
        if self._fired:
            raise ValueError(f"GSPlugin dim {self.Pl.shape[0]} does not match head in_features {D}")
        self.Pl = torch.eye(D, device=self.device, dtype=torch.float32)
        self._r = torch.empty(D, device=self.device, dtype=torch.float32)

    def before_update(self, model, before_batch_input: Optional[torch.Tensor], batch_index: int, len_dataloader: int,
                      train_exp_counter: int, r_mean: Optional[torch.Tensor] = None, grad: Optional[torch.Tensor] = None) -> None:
        """`model` is the shared head (fc_out).  Protocol path (main.py:437-438): the gradient is read from -- and written
        back through -- `w.grad` of the parameter named 'weight' in `model.named_parameters()` (the reading of
        utils/utils.py:30-32 that makes the body run), `before_batch_input` is the attached feature tensor.
        Trainer path: `grad` = the head's flat weight-gradient view, `r_mean` (optional) = a pre-reduced global-batch
        feature mean (data parallel)."""
        if self.mode == "as_published" or train_exp_counter == 0:               # utils/utils.py:29-32 (Q1, Q5)
            return
        write_back = None
        if grad is None:
            w = dict(model.named_parameters()).get("weight")
            if w is None or w.grad is None:
                raise AttributeError("GSPlugin.before_update: the head's weight has no .grad (call loss.backward() first)")
            grad = w.grad.detach()
            if not (grad.is_contiguous() and grad.dtype == torch.float32):
                write_back = w
                grad = grad.contiguous().float()
        G = grad
        C, D = G.shape
        if D != self.Pl.shape[0]:
            self._resize(D)
        if self._ws is None or self._ws.numel() < ops.gs_ws_elems(D, C):
            self._ws = torch.empty(ops.gs_ws_elems(D, C), device=self.device, dtype=torch.float32)
        if r_mean is None:
            X = before_batch_input.detach()                                     # graph-free (Q4)
            comm = getattr(model, "comm", None)
            if comm is not None and comm.active:                                # global-batch mean under data parallel
                ops.colsum(X.contiguous(), self._r, 1.0 / (X.shape[0] * comm.world))
                comm.allreduce_small(self._r)
            else:
                ops.colsum(X.contiguous(), self._r, 1.0 / X.shape[0])           # r = mean(X, 0)      :34
            r_mean = self._r
        ops.gs_project(self.Pl, r_mean, G, self.alpha(batch_index, len_dataloader), self._ws)   # :35-41
        self._fired = True
        if write_back is not None:
            write_back.grad.data.copy_(G)                                       # w.grad.data = ...   :41
