"""FusedSGD: torch.optim.SGD(momentum, weight_decay) semantics (main.py:749) over flat buffers.

Two ways in, one engine (one `mla_sgd_step` launch per flat buffer = per encoder / head):

  * protocol mode -- `FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)`, the drop-in for
    `optim.SGD(...)` at main.py:749.  It is a real `torch.optim.Optimizer` (param_groups, `StepLR` at main.py:760 works,
    torch-format `state_dict()` with per-parameter `momentum_buffer`s in reference layout), but the parameters it
    receives are views of flat buffers (module.py): `step()` finds each parameter's owner and launches once per owner.
    Semantics per owner, exactly torch's per-parameter rule: all `.grad` None -> skipped (torch >= 2 after
    `zero_grad()`), gradients present -> weight decay + momentum + update; `zero_grad(set_to_none=False)` reproduces
    pinned torch 1.8.1, where zeroed gradients still receive weight decay + momentum.  Gradients somebody
    else assigned (not the published flat views) are copied into the flat gradient first; a partially-None owner
    falls back to one launch per parameter segment.
  * trainer mode -- `FusedSGD({"audio": enc_a, "visual": enc_v, "head": head}, ...)`: MLATrainer drives the groups
    explicitly (`mark_ready` / `step_group`) on their own streams.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Union

import torch

from . import ops
from ._lib import MLAHipError


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params: Union[Dict[str, object], Iterable], lr: float = 1e-3, momentum: float = 0.9,
                 weight_decay: float = 1e-4, legacy_zero_grad: bool = False):
        self.legacy_zero_grad = legacy_zero_grad
        if isinstance(params, dict):                          # trainer mode: name -> object with .flat / .grad
            self.groups = dict(params)
            plist = [p for g in self.groups.values() for p in self._owner_params(g)]
            self.protocol = False
        else:
            plist = list(params)
            owners: List[object] = []
            for p in plist:
                for q in (p["params"] if isinstance(p, dict) else [p]):
                    owner = getattr(q, "_mla_owner", None)
                    if owner is None:
                        raise MLAHipError("FusedSGD drives mla_hip parameters only (views of the kernels' flat buffers); "
                                          "got a foreign tensor of shape %s" % (tuple(q.shape),))
                    if not any(o is owner for o in owners):
                        owners.append(owner)
            self.groups = {"%s%d" % (type(o).__name__, i): o for i, o in enumerate(owners)}
            self.protocol = True
        if not plist:       # trainer-mode stand-ins without registered parameters (host-logic tests)
            plist = [torch.zeros(1, requires_grad=True)]
        super().__init__(plist, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.buf = {k: torch.zeros_like(g.flat) for k, g in self.groups.items()}
        self.initialized = {k: False for k in self.groups}
        self.seg_initialized: Dict[str, Optional[List[bool]]] = {k: None for k in self.groups}   # only after a partial step
        # trainer mode: gradient state per group: "none" | "zero" | "ready"
        self.grad_state = {k: "none" for k in self.groups}

    @staticmethod
    def _owner_params(g) -> list:
        return [p for _n, p, _gv in getattr(g, "_entries", [])]

    # ---- hyper-parameters live in param_groups (so lr schedulers work) ---------------------------------------
    def _hyper(self, name: str):
        owner = self.groups[name]
        first = self._owner_params(owner)
        for grp in self.param_groups:
            if not first or any(q is first[0] for q in grp["params"]):
                return grp["lr"], grp["momentum"], grp["weight_decay"]
        g0 = self.param_groups[0]
        return g0["lr"], g0["momentum"], g0["weight_decay"]

    @property
    def lr(self) -> float:
        return self.param_groups[0]["lr"]

    def set_lr(self, lr: float) -> None:
        for grp in self.param_groups:
            grp["lr"] = lr

    # ---- engine -------------------------------------------------------------------------------------------------
    def _launch(self, name: str, with_grad: bool) -> None:
        g = self.groups[name]
        lr, mom, wd = self._hyper(name)
        if self.seg_initialized[name] is not None:
            self._launch_segments(name, [with_grad] * len(g._entries))
            return
        ops.sgd_step(g.flat, g.grad if with_grad else None, self.buf[name], lr, mom, wd, first=not self.initialized[name])
        self.initialized[name] = True

    def _launch_segments(self, name: str, has_grad: List[Optional[bool]]) -> None:
        """Per-parameter launches (only when an owner's gradients are partially None): has_grad[i] None = skip."""
        g = self.groups[name]
        lr, mom, wd = self._hyper(name)
        if self.seg_initialized[name] is None:
            self.seg_initialized[name] = [self.initialized[name]] * len(g._entries)
        seg = self.seg_initialized[name]
        for i, (o, n) in enumerate(g.segments()):
            if has_grad[i] is None:
                continue
            ops.sgd_step(g.flat[o:o + n], g.grad[o:o + n] if has_grad[i] else None, self.buf[name][o:o + n], lr, mom, wd,
                         first=not seg[i])
            seg[i] = True
        self.initialized[name] = self.initialized[name] or all(seg)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self.protocol:                                       # trainer mode
            for k in self.groups:
                self.step_group(k)
            return loss
        for name, g in self.groups.items():                         # protocol mode: torch's per-parameter rule, per owner
            works = getattr(g, "_grad_works", None)
            if works:
                g.comm.wait(works)                                  # data parallel: encoder gradients reduced
                g._grad_works = []
            state = g.grads_alias_flat()
            if state is None:
                continue                                            # every p.grad is None -> skipped (torch >= 2 zero_grad)
            if state:
                self._launch(name, True)
                continue
            has = []
            for i, (_n, p, gv) in enumerate(g._entries):            # foreign or partial gradients
                if p.grad is None:
                    has.append(None)
                    continue
                if g._published is None or p.grad is not g._published[i]:
                    gv.copy_(p.grad)
                    p.grad = gv
                has.append(True)
            if g._published is None:
                g._published = [None] * len(g._entries)
            for i, (_n, p, gv) in enumerate(g._entries):
                if has[i]:
                    g._published[i] = gv
            if all(h for h in has):
                self._launch(name, True)
            else:
                self._launch_segments(name, has)
        return loss

    def zero_grad(self, set_to_none: bool = True) -> None:
        """main.py:164, 440, 452.  set_to_none=False (or legacy_zero_grad) = torch 1.8.1: tensors stay, zero-filled."""
        if not self.protocol:
            for k in self.groups:
                if self.grad_state[k] != "none":
                    self.grad_state[k] = "zero" if self.legacy_zero_grad else "none"
            return
        if set_to_none and not self.legacy_zero_grad:
            for g in self.groups.values():
                for _n, p, _gv in g._entries:
                    p.grad = None
            return
        for g in self.groups.values():
            state = g.grads_alias_flat()
            if state is None:
                continue
            if state:
                g.grad.zero_()                                      # one memset for the whole owner
            else:
                for _n, p, _gv in g._entries:
                    if p.grad is not None:
                        p.grad.zero_()

    # ---- trainer-mode state machine ---------------------------------------------------------------------------------
    def mark_ready(self, name: str) -> None:
        self.grad_state[name] = "ready"

    def step_group(self, name: str) -> None:
        state = self.grad_state[name]
        if state == "none":
            return                                                  # p.grad is None -> skipped by torch.optim.SGD
        self._launch(name, state == "ready")                        # "zero": zeroed grads (1.8.1): wd + momentum still apply

    def drop_grads(self) -> None:
        """main.py:468-470: `del p.grad` for every parameter."""
        for k in self.groups:
            self.grad_state[k] = "none"

    # ---- (de)serialisation: torch's format, momentum buffers in reference layout -------------------------------------
    def _mom_view(self, name: str, i: int) -> torch.Tensor:
        g = self.groups[name]
        gv = g._entries[i][2]
        return torch.as_strided(self.buf[name], gv.shape, gv.stride(), gv.storage_offset() - g.grad.storage_offset())

    def state_dict(self) -> dict:
        """torch.optim.SGD's layout (main.py:922): {'state': {index: {'momentum_buffer': tensor}}, 'param_groups': [...]}
        with parameters indexed in `model.parameters()` order; buffers are contiguous copies in the reference layout."""
        for name, g in self.groups.items():
            if hasattr(g, "_await_tail"):
                g._await_tail()
            seg = self.seg_initialized[name]
            for i, (_n, p, _gv) in enumerate(getattr(g, "_entries", [])):
                if (seg[i] if seg is not None else self.initialized[name]):
                    self.state[p]["momentum_buffer"] = self._mom_view(name, i).clone(memory_format=torch.contiguous_format)
                else:
                    self.state.pop(p, None)
        sd = super().state_dict()
        sd["mla_hip"] = {"initialized": dict(self.initialized)}
        return sd

    def load_state_dict(self, sd: dict) -> None:
        """Accepts a torch.optim.SGD state_dict of the reference (same parameter order) or one of our own."""
        sd = dict(sd)
        sd.pop("mla_hip", None)
        super().load_state_dict(sd)
        for name, g in self.groups.items():
            ents = getattr(g, "_entries", [])
            seg = []
            for i, (_n, p, _gv) in enumerate(ents):
                mb = self.state.get(p, {}).get("momentum_buffer")
                seg.append(mb is not None)
                if mb is not None:
                    self._mom_view(name, i).copy_(mb)
            self.initialized[name] = bool(seg) and all(seg)
            self.seg_initialized[name] = None if (all(seg) or not any(seg)) else seg
