"""nn.Module face of the flat-buffer objects: the object protocol main.py programs against.

The reference has no FFI; what `main.py` touches is the PyTorch protocol -- `model(...)` returning tensors with
autograd history, `named_parameters()` / `parameters()` / `state_dict()` / `load_state_dict()` under the
reference's dotted names and layouts, `p.grad` mutated in place by `GSPlugin.before_update`
(utils/utils.py:30-41), `model.apply(weight_init)` (main.py:719), `optim.SGD(model.parameters(), ...)`
(main.py:749).  The HIP kernels work on ONE flat fp32 buffer per encoder (HWIO / [in][out] layouts), so the
protocol objects here are *views*:

  * every reference parameter is an `nn.Parameter` whose storage IS the flat buffer -- a strided view in the
    reference's logical layout (OIHW conv weights = HWIO permuted, nn.Linear (out,in) = [in][out] transposed);
    in-place writes through it (load_state_dict, nn.init.*) land in the flat buffer, nothing is copied;
  * after a HIP backward the parameters' `.grad` are set to the same kind of views of the flat GRADIENT buffer,
    with autograd's accumulation rule (None -> set, otherwise add);
  * leaf holders subclass nn.Conv2d / nn.BatchNorm2d / nn.Linear (without allocating their own storage) so
    `isinstance` dispatch in the reference's `weight_init` (utils/utils.py:106-114) works unchanged.

The modules are pinned to the device and dtype they were built with: `.to()/.cuda()/.float()` that would
re-allocate storage raise (the views would silently detach from the kernels' buffers otherwise); no-op moves
(main.py:730, 734) pass.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from ._lib import MLAHipError


class Holder(nn.Module):
    """Interior node of a reference module path (`layer1`, `layer1.0`, `encoder.blocks.3`, ...): carries names only."""

    def forward(self, *a, **k):
        raise MLAHipError("this submodule only carries parameter names; call the owning encoder / classifier")


def _bare_init(self) -> None:
    nn.Module.__init__(self)


class Conv2dHolder(nn.Conv2d):
    """`isinstance(m, nn.Conv2d)` leaf whose `weight` (OIHW) is a view of the encoder's flat HWIO buffer."""

    def __init__(self, weight: nn.Parameter, stride: int, padding: int):
        _bare_init(self)
        co, ci, kh, kw = weight.shape
        self.in_channels, self.out_channels, self.kernel_size = ci, co, (kh, kw)
        self.stride, self.padding, self.dilation, self.groups = (stride, stride), (padding, padding), (1, 1), 1
        self.transposed, self.output_padding, self.padding_mode = False, (0, 0), "zeros"
        self._reversed_padding_repeated_twice = (padding,) * 4
        self.weight = weight
        self.register_parameter("bias", None)                                     # backbone.py:4-12: bias=False

    def forward(self, *a, **k):
        raise MLAHipError("convolutions run inside the encoder's launch plan; call the encoder / classifier")


class BatchNorm2dHolder(nn.BatchNorm2d):
    """`isinstance(m, nn.BatchNorm2d)` leaf: weight / bias views of the flat parameter buffer, running statistics
    views of the encoder's flat running buffer (momentum 0.1, eps 1e-5: backbone.py:22)."""

    def __init__(self, weight: nn.Parameter, bias: nn.Parameter, running_mean: torch.Tensor, running_var: torch.Tensor):
        _bare_init(self)
        self.num_features, self.eps, self.momentum = weight.shape[0], 1e-5, 0.1
        self.affine, self.track_running_stats = True, True
        self.weight, self.bias = weight, bias
        self.register_buffer("running_mean", running_mean)
        self.register_buffer("running_var", running_var)
        self.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long, device=weight.device))

    def forward(self, *a, **k):
        raise MLAHipError("batch norm runs inside the encoder's launch plan; call the encoder / classifier")


class FlatModule(nn.Module):
    """Base of every object that owns flat `flat` / `grad` buffers and exposes them under reference names."""

    def __init__(self):
        nn.Module.__init__(self)        # explicit: subclasses also inherit nn.Linear (SharedHead), whose ctor allocates
        self._entries: List[Tuple[str, nn.Parameter, torch.Tensor]] = []      # (reference name, parameter, grad view)
        self._published: Optional[List[torch.Tensor]] = None                  # grad views currently installed as p.grad
        self.comm = None                                                      # set by mla_hip.DataParallel

    # ---- registration -------------------------------------------------------------------------------------
    def _param_view(self, internal: torch.Tensor, grad_internal: torch.Tensor, to_ref: Callable[[torch.Tensor], torch.Tensor],
                    name: str) -> nn.Parameter:
        """Parameter = `to_ref(internal)` (a VIEW of the flat buffer, reference layout) + the matching gradient view."""
        v = to_ref(internal)
        if v.untyped_storage().data_ptr() != internal.untyped_storage().data_ptr():
            raise MLAHipError(f"{name}: reference view does not alias the flat buffer")
        p = nn.Parameter(v, requires_grad=True)
        p._mla_owner = self
        p._mla_index = len(self._entries)
        self._entries.append((name, p, to_ref(grad_internal)))
        return p

    @staticmethod
    def _descend(root: nn.Module, dotted: str) -> Tuple[nn.Module, str]:
        parts = dotted.split(".")
        m = root
        for part in parts[:-1]:
            if part not in m._modules:
                m.add_module(part, Holder())
            m = m._modules[part]
        return m, parts[-1]

    # ---- gradients: autograd's AccumulateGrad rule on top of kernels that overwrite the flat gradient ----------
    def grads_pending(self):
        """Call BEFORE a HIP backward: returns what must be added back afterwards (None in the common case where
        every .grad is None, i.e. after optimizer.zero_grad() / `del p.grad`, main.py:440, 468-470)."""
        keep = None
        alias_snapshot = None
        for i, (_n, p, gv) in enumerate(self._entries):
            g = p.grad
            if g is None:
                continue
            if keep is None:
                keep = []
            if self._published is not None and g is self._published[i]:
                if alias_snapshot is None:
                    alias_snapshot = self.grad.clone()                    # the kernels are about to overwrite it
                keep.append((i, None))
            else:
                keep.append((i, g))
        return None if keep is None else (keep, alias_snapshot)

    def publish_grads(self, pending=None) -> None:
        """Call AFTER a HIP backward filled `self.grad`: p.grad <- view (or accumulated)."""
        if pending is not None:
            keep, snap = pending
            aliased = [i for i, g in keep if g is None]
            if len(aliased) == len(self._entries):
                self.grad.add_(snap)                                      # the usual accumulation case: one flat add
            else:
                for i in aliased:
                    self._entries[i][2].add_(self._snap_view(snap, i))
            for i, g in keep:
                if g is not None:                                         # a gradient tensor somebody else assigned
                    self._entries[i][2].add_(g)
        pub = []
        for _n, p, gv in self._entries:
            p.grad = gv
            pub.append(gv)
        self._published = pub

    def _snap_view(self, snap: torch.Tensor, i: int) -> torch.Tensor:
        gv = self._entries[i][2]
        return torch.as_strided(snap, gv.shape, gv.stride(), gv.storage_offset() - self.grad.storage_offset())

    def segments(self) -> List[Tuple[int, int]]:
        """(offset, numel) of every registered parameter inside the flat buffers (each is one contiguous range)."""
        base = self.grad.storage_offset()
        return [(gv.storage_offset() - base, gv.numel()) for _n, _p, gv in self._entries]

    def grads_alias_flat(self) -> Optional[bool]:
        """True: every p.grad is the published view (one fused optimizer launch is legal); None: every p.grad is None;
        False: anything else (foreign / partial gradients)."""
        some, none, alias = False, False, True
        for i, (_n, p, _gv) in enumerate(self._entries):
            g = p.grad
            if g is None:
                none = True
            else:
                some = True
                if self._published is None or g is not self._published[i]:
                    alias = False
        if not some:
            return None
        return alias and not none

    # ---- device / dtype pinning -----------------------------------------------------------------------------
    def _apply(self, fn, recurse=True):
        probe = torch.empty(0, device=self.device, dtype=torch.float32)
        out = fn(probe)
        if out.device != probe.device or out.dtype != probe.dtype:
            raise MLAHipError(f"mla_hip modules are pinned to {probe.device} / float32 (their parameters are views of the "
                              f"kernels' flat buffers); requested {out.device} / {out.dtype}")
        return self
