"""The remaining protocol objects main.py constructs around the model:

  DataParallel      `model = torch.nn.DataParallel(model, device_ids=gpu_ids)` (main.py:732).  The reference's only
                    parallelism is single-process DataParallel; the MI355X design is one process per GPU over RCCL, so
                    this wrapper does not replicate / scatter / gather: it keeps the `.module` path and the call
                    signature, and attaches the process group (`Comm`) to every flat-buffer module so that the
                    autograd Functions and FusedSGD exchange gradients as SURVEY 8e prescribes.
  CrossEntropyLoss  `criterion = nn.CrossEntropyLoss()` (main.py:130) on the HIP kernels (torch's own criterion also
                    works on the logits `fc_out` returns; this one keeps the step free of ATen kernels).
  weight_init       utils/utils.py:106-114, setup_seed utils/utils.py:98-103: mirrors (the reference's own functions
                    work unchanged on mla_hip modules too: leaves are nn.Linear / nn.Conv2d / nn.BatchNorm2d instances).
"""
from __future__ import annotations

import random
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .autograd import SoftmaxCE
from .dist import Comm
from .module import FlatModule


class DataParallel(nn.Module):
    def __init__(self, module: nn.Module, device_ids=None, output_device=None, dim: int = 0, comm: Optional[Comm] = None):
        super().__init__()
        self.module = module
        self.device_ids = device_ids
        self.comm = comm if comm is not None else Comm()
        for m in module.modules():
            if isinstance(m, FlatModule):
                m.comm = self.comm
        if self.comm.active:                                   # replicas start identical (DataParallel broadcasts GPU 0's)
            for m in module.modules():
                if isinstance(m, FlatModule):
                    self.comm.broadcast_(m.flat)
                    if hasattr(m, "running"):
                        self.comm.broadcast_(m.running)
                    for p in getattr(m, "unused", {}).values():
                        self.comm.broadcast_(p.data)

    def forward(self, *inputs, **kwargs):
        return self.module(*inputs, **kwargs)


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() with the defaults main.py:130 uses (mean reduction, no weights, no smoothing)."""

    def forward(self, logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return SoftmaxCE.apply(logits, target)


_INIT_RULES = (
    # (leaf type, weight initialiser, bias constant or None)            utils/utils.py:106-114
    (nn.Linear, nn.init.xavier_normal_, 0.0),
    (nn.Conv2d, lambda w: nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu"), None),
    (nn.BatchNorm2d, lambda w: nn.init.constant_(w, 1.0), 0.0),
)


def weight_init(m: nn.Module) -> None:
    """`model.apply(weight_init)` (main.py:719): xavier-normal Linear with zero bias, kaiming-normal (fan_out, relu)
    Conv2d, BatchNorm weight 1 / bias 0.  The initialisers write through the parameter views into the flat buffers."""
    for leaf_type, init_w, bias_value in _INIT_RULES:
        if isinstance(m, leaf_type):
            init_w(m.weight)
            if bias_value is not None and m.bias is not None:
                nn.init.constant_(m.bias, bias_value)
            return


def setup_seed(seed: int) -> None:
    """utils/utils.py:98-103 (cudnn.deterministic has no meaning here: every kernel of the library is deterministic
    except the embedding scatter-add of the text encoder)."""
    for seeder in (random.seed, np.random.seed, torch.manual_seed, torch.cuda.manual_seed_all):
        seeder(seed)
