"""Data-parallel exchange for the MLA step: one process per GPU, RCCL over xGMI ("nccl" backend on
ROCm); `gloo` for the CPU tests.

The reference's only parallelism is torch.nn.DataParallel (main.py:732): per-replica BatchNorm
statistics, shared head + CE + projection on the GLOBAL batch on GPU 0, encoder gradients
reduce-added.  The MI355X-native equivalent keeps a full replica per rank and exchanges:

  * encoder gradients: ONE flat buffer per encoder (44.7 MB), all-reduce(SUM) in a few large
    buckets on RCCL's stream, asynchronous -- it overlaps the other modality's head + backward and
    is only waited for right before that encoder's SGD launch;
  * head path (latency-bound, on the critical path): dW, db, the feature column sum and the loss
    are packed into ONE 3.6K-float message and all-reduced once per modality phase, so every rank
    applies the identical projection and head SGD on global-batch quantities.

All local quantities are already scaled by 1/B_global (the CE mean is over the global batch), so
SUM is the only reduction needed.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def _high_priority_options(group):
    """RCCL communicator of the head exchange on a HIGH-PRIORITY stream: the 3.6 K-float message is on the step's critical path,
    and priority streams get hardware queues of their own, so its kernel neither waits behind compute packets nor shares a
    queue with an encoder chain (streams.py).  None (default options) for other backends or older torch."""
    try:
        if dist.get_backend(group) != "nccl":
            return None
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        return opts
    except Exception:       # pragma: no cover - backend without options
        return None


# One head communicator per parent group for the life of the process: dist.new_group is a COLLECTIVE and its communicator is
# never destroyed, so building one per Comm() leaked RCCL communicators / streams and turned e.g. `Evaluator(model)` on a subset
# of the ranks into a deadlock.  Keyed by the parent group object (None = the default group).
_HEAD_GROUPS: dict = {}


def _head_group_for(group):
    parent = group if group is not None else dist.distributed_c10d._get_default_group()
    entry = _HEAD_GROUPS.get(id(parent))          # the entry keeps `parent` alive, so its id is never reused while cached
    if entry is None or entry[0] is not parent:   # (a destroy_process_group / init_process_group cycle makes a new default group)
        ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
        entry = _HEAD_GROUPS[id(parent)] = (parent, dist.new_group(ranks=ranks, pg_options=_high_priority_options(group)))
    return entry[1]


class Comm:
    def __init__(self, group=None, bucket_bytes: int = 16 << 20, force: bool = False, dedicated_head_group: bool = True):
        """force=True issues every collective even with a single rank (used to rehearse the RCCL path on
        a one-GPU box); results are unchanged because SUM over one rank is the identity."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.active = (self.world > 1) or (force and dist.is_initialized())
        # The head path is latency-critical and must not queue behind the 44.7 MB encoder-gradient buckets: it
        # gets its own communicator (own RCCL stream).  new_group is collective: the FIRST Comm of a parent group must be built
        # by every rank at the same point (MLATrainer / DataParallel construction); later ones reuse the cached communicator.
        # The RCCL multi-rank path is unverified until a multi-GPU run exists.
        self.head_group = group
        if self.active and dedicated_head_group:
            self.head_group = _head_group_for(group)

    # ---- encoder gradients ------------------------------------------------------------------
    def allreduce_flat_async(self, flat: torch.Tensor) -> List:
        """all-reduce(SUM) `flat` in place in buckets; returns the work handles (empty if world == 1)."""
        if not self.active:
            return []
        works = []
        for o in range(0, flat.numel(), self.bucket_elems):
            works.append(dist.all_reduce(flat[o:o + self.bucket_elems], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return works

    @staticmethod
    def wait(works: List) -> None:
        for w in works:
            w.wait()

    # ---- head exchange ----------------------------------------------------------------------
    def exchange_head(self, head_grad_flat: torch.Tensor, colsum: torch.Tensor, loss: torch.Tensor,
                      scratch: Optional[torch.Tensor] = None) -> None:
        """In-place SUM over ranks of (dW|db), the feature column sum and the loss: one message."""
        if not self.active:
            return
        n0, n1 = head_grad_flat.numel(), colsum.numel()
        n = n0 + n1 + 1
        msg = scratch[:n] if scratch is not None else torch.empty(n, device=head_grad_flat.device, dtype=torch.float32)
        msg[:n0].copy_(head_grad_flat)
        msg[n0:n0 + n1].copy_(colsum)
        msg[n0 + n1:].copy_(loss.reshape(1))
        dist.all_reduce(msg, op=dist.ReduceOp.SUM, group=self.head_group)
        head_grad_flat.copy_(msg[:n0])
        colsum.copy_(msg[n0:n0 + n1])
        loss.reshape(1).copy_(msg[n0 + n1:])

    def allreduce_small(self, t: torch.Tensor) -> None:
        """Blocking in-place SUM of a small critical-path message (packed dW|db, feature mean) on the head communicator."""
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.head_group)

    def allgather_rows(self, t: torch.Tensor) -> torch.Tensor:
        """(B_local, C) -> (world * B_local, C) in rank order (evaluation under --dynamic needs global-batch logits, Q9)."""
        if not self.active:
            return t
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        dist.all_gather_into_tensor(out, t.contiguous(), group=self.head_group)
        return out

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> None:
        if self.active:
            dist.broadcast(t, src=src, group=self.group)
