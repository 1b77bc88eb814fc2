"""The MLA alternating-unimodal training step (main.py:419-476), fused driver.

    joint forward of both encoders                                   main.py:431
    for modality in (audio, visual):                                 main.py:432-454
        head forward + CE + head/feature gradients                   :432-435 / :444-447
        encoder backward
        GSPlugin.before_update on the head gradient                  :437 / :449
        optimizer.step(); optimizer.zero_grad(); exp_count += 1      :439-442 / :451-454
    drop gradients, accumulate the reported losses                   :468-476

Orchestration only: every arithmetic step is a libmla_hip.so kernel.  Quirks reproduced:
Q5 (exp_count / alpha), Q6 (`legacy_zero_grad`), Q7 (visual logits use the head already updated by the
audio step), Q8 (av_alpha fixed at 0.55 in the reported loss).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .dist import Comm
from .model import AVClassifier
from .optim import FusedSGD
from .plugin import GSPlugin


class MLATrainer:
    def __init__(self, model: AVClassifier, lr: float = 1e-3, momentum: float = 0.9, weight_decay: float = 1e-4,
                 gs_mode: str = "as_intended", legacy_zero_grad: bool = False, av_alpha: float = 0.55,
                 comm: Optional[Comm] = None):
        self.model = model
        self.head = model.fusion_module.fc_out
        self.gs_plugin = GSPlugin(dim=self.head.in_features, device=model.device, mode=gs_mode)
        self.optimizer = FusedSGD({"audio": model.audio_net, "visual": model.visual_net, "head": self.head},
                                  lr, momentum, weight_decay, legacy_zero_grad)
        self.av_alpha = av_alpha
        self.comm = comm if comm is not None else Comm()
        dev = model.device
        self._colsum = torch.empty(self.head.in_features, device=dev, dtype=torch.float32)
        self._msg = torch.empty(self.head.numel + self.head.in_features + 1, device=dev, dtype=torch.float32)
        self.losses = {k: torch.zeros(1, device=dev, dtype=torch.float32) for k in ("loss", "loss_a", "loss_v")}
        self.last = {}

    def _phase(self, name: str, enc, feat: torch.Tensor, pooled_px: int, label: torch.Tensor, inv_batch: float,
               batch_step: int, len_dataloader: int, pending: list):
        logits, loss, dX = self.head.forward_backward(feat, label, inv_batch, slot=name)               # :432-435
        self.last["out_" + name] = logits
        self.losses["loss_" + name].copy_(loss)
        enc.backward_from_pooled(dX, pooled_px)                                               # loss.backward()
        works = self.comm.allreduce_flat_async(enc.grad)                                     # overlaps what follows
        fires = self.gs_plugin.mode == "as_intended" and self.gs_plugin.exp_count != 0
        r_mean = None
        if self.comm.active:
            ops.colsum(feat, self._colsum, inv_batch)
            self.comm.exchange_head(self.head.grad, self._colsum, self.losses["loss_" + name], self._msg)
            r_mean = self._colsum
        self.last[f"head_grad_{name}_raw"] = self.head.weight_grad.clone() if self.keep_debug else None
        if fires:
            self.gs_plugin.before_update(self.head, feat, batch_step, len_dataloader, self.gs_plugin.exp_count,
                                         r_mean=r_mean)                                       # :437-438
        opt = self.optimizer
        opt.mark_ready("head")
        opt.step_group("head")                                                                # optimizer.step(): head
        pending.append((name, works))
        self.gs_plugin.exp_count += 1                                                         # :442

    keep_debug = False

    def train_step(self, spec: torch.Tensor, image: torch.Tensor, label: torch.Tensor, batch_step: int,
                   len_dataloader: int):
        """spec (B,H,W) or (B,1,H,W); image (B,3,T,H,W); label int64 (B,).  Returns device scalars
        {'loss','loss_a','loss_v'} (no host sync; call .item() when needed, main.py:472-476)."""
        m, opt = self.model, self.optimizer
        if spec.dim() == 3:
            spec = spec.unsqueeze(1)                                                          # main.py:431
        B = spec.shape[0]
        inv_batch = 1.0 / (B * self.comm.world)
        opt.zero_grad()                                                                       # main.py:164
        a, v = m.forward(spec.float(), image.float())                                         # main.py:431
        self.last["a"], self.last["v"] = a, v
        pending: list = []
        # ---- audio phase.  Its encoder SGD is deferred until its all-reduce has landed; the visual
        # phase does not read audio parameters, so enqueueing it first changes no result.
        self._phase("a", m.audio_net, a, m._pa, label, inv_batch, batch_step, len_dataloader, pending)
        opt.mark_ready("audio")
        if not self.comm.active:
            opt.step_group("audio")
        opt_legacy_audio = opt.legacy_zero_grad
        # ---- visual phase
        self._phase("v", m.visual_net, v, m._pv, label, inv_batch, batch_step, len_dataloader, pending)
        opt.mark_ready("visual")
        if self.comm.active:
            self.comm.wait(pending[0][1])
            opt.step_group("audio")
            self.comm.wait(pending[1][1])
        if opt_legacy_audio:                      # torch 1.8.1: audio grads are zero (not None) in the visual step (Q6)
            opt.grad_state["audio"] = "zero"
            opt.step_group("audio")
        opt.step_group("visual")
        opt.drop_grads()                                                                      # main.py:468-470
        torch.add(self.losses["loss_a"] * self.av_alpha, self.losses["loss_v"], alpha=1 - self.av_alpha,
                  out=self.losses["loss"])                                                    # main.py:472 (Q8)
        return self.losses
