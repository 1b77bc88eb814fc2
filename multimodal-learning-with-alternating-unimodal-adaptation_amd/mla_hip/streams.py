"""HIP streams on distinct hardware queues.

ROCm multiplexes a process's HIP streams onto a few hardware queues (4 by default, `GPU_MAX_HW_QUEUES`); two streams that
share a queue run their kernels strictly one after the other, whatever the events between them say.  Which streams share
depends on creation order, so the per-encoder stream pipeline of MLATrainer kept or lost its overlap depending on
how many streams the process had created before (measured: two streams created first -> both encoder chains on ONE queue,
38.6 instead of 34.9 ms per step; RCCL, a second model or a data loader do exactly that).  `distinct_streams` therefore
*measures* which streams serialise -- two 1-thread spin kernels on a pair of streams take 1x or 2x the time of one -- and
hands out streams that do not, most important first.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional

import torch

_SPIN_CYCLES = 300_000          # ~0.13 ms per spin kernel (well above the ~15 us launch noise of the host timer)
_MAX_STREAMS = 16               # streams created at most while looking for free queues
_pools: Dict[int, "_Pool"] = {}


class _Pool:
    def __init__(self, device: torch.device):
        self.device = device
        self.classes: List[List[torch.cuda.Stream]] = []      # streams grouped by the hardware queue they share
        self.created = 0
        self.t_single: Optional[float] = None

    def _time(self, a: torch.cuda.Stream, b: Optional[torch.cuda.Stream]) -> float:
        best = float("inf")
        for _ in range(2):
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            with torch.cuda.stream(a):
                torch.cuda._sleep(_SPIN_CYCLES)
            if b is not None:
                with torch.cuda.stream(b):
                    torch.cuda._sleep(_SPIN_CYCLES)
            torch.cuda.synchronize(self.device)
            best = min(best, time.perf_counter() - t0)
        return best

    def shares_queue(self, a: torch.cuda.Stream, b: torch.cuda.Stream) -> bool:
        if self.t_single is None:
            self.t_single = self._time(a, None)
        return self._time(a, b) > 1.6 * self.t_single

    def class_of(self, s: torch.cuda.Stream) -> Optional[int]:
        for k, cls in enumerate(self.classes):
            if self.shares_queue(cls[0], s):
                return k
        return None

    def new_stream(self) -> int:
        """Create one stream, file it under its hardware queue, return the class index."""
        s = torch.cuda.Stream(device=self.device)
        self.created += 1
        k = self.class_of(s)
        if k is None:
            self.classes.append([s])
            return len(self.classes) - 1
        self.classes[k].append(s)
        return k


def distinct_streams(n: int, device: torch.device, avoid: Optional[List[torch.cuda.Stream]] = None) -> List[torch.cuda.Stream]:
    """n streams, listed by importance: as many of the first ones as there are free hardware queues run concurrently with each
    other and with the `avoid` streams (default: the current stream); the rest share queues with later entries first.  Streams
    are drawn from a per-device pool that lives for the process, so repeated calls (second trainer, evaluation model) reuse them."""
    device = torch.device(device)
    if device.type != "cuda" or not hasattr(torch.cuda, "_sleep"):
        return [torch.cuda.Stream(device=device) for _ in range(n)]
    pool = _pools.setdefault(device.index if device.index is not None else torch.cuda.current_device(), _Pool(device))
    with torch.cuda.device(device):
        avoid = [torch.cuda.current_stream(device)] if avoid is None else list(avoid)
        while len(pool.classes) < n + len(avoid) and pool.created < _MAX_STREAMS:
            pool.new_stream()
        busy = set()
        for a in avoid:
            k = pool.class_of(a)
            if k is not None:
                busy.add(k)
        free = [k for k in range(len(pool.classes)) if k not in busy]
        if not free:
            return [torch.cuda.Stream(device=device) for _ in range(n)]
        out: List[torch.cuda.Stream] = []
        for i in range(n):
            if i < len(free):
                out.append(pool.classes[free[i]][0])
            else:          # out of free queues: the less important entries share a stream (they would serialise on the queue anyway)
                out.append(out[len(free) - 1 - (i - len(free)) % len(free)])
        return out
