"""Host -> device batch feed for the MLA step.

Batch tensor contract of the reference's datasets (dataset/dataset.py:111-161 AVDataset, 448-480, 753-803):
    CREMA-D:  (spec (1024,128) fp32, image (3,T,224,224) fp32, label int64, idx)     -> train_step(spec, image, label, ...)
    Food-101: (token (1,256) int64, padding_mask (1,256) fp32, image (3,256,256), label, idx)
    IEMOCAP:  (token, padding_mask, image, spec, label, idx)
The reference does a blocking `.to(device)` per tensor at the top of every iteration (main.py:146-162): 149 MB per
CREMA-D batch of 64 on the critical path.  `DeviceFeeder` copies each batch on a dedicated HIP stream into one of
`depth` device slots while the previous step computes; the compute stream only waits on the copy's event, so the PCIe
transfer (149 MB ~ 2.3-3 ms at the measured 50 GB/s) disappears behind the 37 ms step.

Slot lifetime: a slot is refilled once the stream the batch was consumed on has passed the end of that step (`done`
event).  Every encoder copies what its BACKWARD reads of the raw input (audio spectrogram for the stem weight gradient,
token ids for the embedding scatter) into its own workspace during forward(), and the trainer's calling stream waits for
all forwards before it returns, so no kernel touches a slot after `done` even though the encoder chains keep running on
their own streams (tests/test_step_gpu.py::test_device_feeder_with_stream_pipeline_is_bitwise_equivalent).
The feeder does not pin anything itself: copies are asynchronous when the loader hands over pinned tensors
(DataLoader(pin_memory=True), main.py:785) and host-blocking (but still off the compute stream) for pageable ones.
A source that RECYCLES pinned host memory (data.NpyBatcher's staging ring) exposes `copied(event)`: the feeder hands it the
event recorded behind each batch's copies, and the source waits on it before overwriting that memory (a copy that has been
issued has not necessarily run: the copy stream waits on device events of earlier steps and nothing here host-syncs).
"""
from __future__ import annotations

from typing import Iterable, Iterator, List, Sequence, Tuple

import torch


class DeviceFeeder:
    def __init__(self, batches: Iterable[Sequence[torch.Tensor]] = (), device="cuda", depth: int = 3):
        """`batches`: iterable of tuples of host tensors (any dtype); tensors keep their dtype and shape.  Keep one
        feeder per data loader for the whole run (the device slots are allocated once and reused by `feed()`).
        depth >= 3: while step i runs, batch i+1 is already on the device and batch i+2 is copied into the slot of step
        i-1.  Copies are issued on a dedicated stream straight from the loader's host tensors: asynchronous for pinned
        sources (DataLoader(pin_memory=True), main.py:785), ~2.3 ms of host time per 149 MB batch for pageable ones
        (the HIP runtime stages them at ~50 GB/s; this class pins nothing).  Slot reuse is fenced on the device by events
        (no host sync)."""
        self.batches = batches
        self.device = torch.device(device)
        self.depth = max(3, depth)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._dev: List[List[torch.Tensor]] = []
        self._free: List[torch.cuda.Event] = []          # slot may be overwritten once the step that used it is done

    def _issue(self, k: int, batch: Sequence[torch.Tensor]) -> torch.cuda.Event:
        if k >= len(self._dev):
            self._dev.append([torch.empty(t.shape, dtype=t.dtype, device=self.device) for t in batch])
            self._free.append(None)
        ready = torch.cuda.Event()
        with torch.cuda.stream(self.copy_stream):
            if self._free[k] is not None:
                self.copy_stream.wait_event(self._free[k])       # the step that read this slot has finished (device-side fence)
            for t, d in zip(batch, self._dev[k]):
                d.copy_(t, non_blocking=True)
            ready.record()
        copied = getattr(self.batches, "copied", None)   # sources that recycle pinned staging memory (data.NpyBatcher) refill a
        if copied is not None:                           # staging tuple only once the copies out of it have completed
            copied(ready)
        return ready

    def feed(self, batches: Iterable[Sequence[torch.Tensor]]) -> "DeviceFeeder":
        """Iterate over another epoch / loader with the same staging slots."""
        self.batches = batches
        return self

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, ...]]:
        it = iter(self.batches)
        inflight = []                                    # (slot, ready event)
        k = 0
        for _ in range(self.depth - 1):                  # prefetch
            try:
                b = next(it)
            except StopIteration:
                break
            inflight.append((k, self._issue(k, b)))
            k = (k + 1) % self.depth
        while inflight:
            slot, ready = inflight.pop(0)
            torch.cuda.current_stream().wait_event(ready)        # compute waits for this batch only
            yield tuple(self._dev[slot])                         # the consumer enqueues its step FIRST ...
            done = torch.cuda.Event()
            done.record()
            self._free[slot] = done
            try:                                                 # ... then the next batch is staged while that step runs
                b = next(it)
                inflight.append((k, self._issue(k, b)))
                k = (k + 1) % self.depth
            except StopIteration:
                pass
