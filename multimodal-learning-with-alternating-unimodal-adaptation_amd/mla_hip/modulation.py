"""OGM / OGM-GE gradient modulation: the `--modulation OGM | OGM_GE` surface (main.py:312-410).

    score_m  = sum_i softmax(out_m)[i][label_i]                         main.py:373-374 (314-316 with three modalities)
    ratio_v  = score_v / score_a,  ratio_a = 1 / ratio_v                 :376-377 (:319-321: score_m / sum of the others)
    coeff    = 1 - tanh(alpha * relu(ratio)) for the dominant modality, 1 for the others      :379-384 (:323-337)
    for every 4-D (conv) gradient of encoder m (name contains 'audio' / 'visual' resp. 'mae_a|v|t', :394-408 / :347-369):
        OGM:     grad *= coeff_m
        OGM_GE:  grad  = grad * coeff_m + N(0, grad.std() + 1e-8)

The reference does this with ~4 ATen kernels and one `.item()` host sync per tensor inside a Python loop over
named_parameters().  Here the coefficients never leave the device and each encoder's conv gradients -- one contiguous
region of its flat gradient buffer -- are modulated by one launch (OGM) or three (OGM-GE: fp64 statistics, finalize,
scale + counter-based Gaussian noise).  Works on both doors of the boundary: the flat buffers MLATrainer drives and the
`.grad` views the autograd path publishes are the same memory.  (Because those `.grad`s are ordinary tensors, the
reference's own loop -- `parms.grad *= coeff_a`, `parms.grad = parms.grad * coeff_a + noise` -- also runs unchanged on
mla_hip modules; FusedSGD honours re-assigned gradients.)
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import ops
from ._lib import MLAHipError


def conv_grad_segments(enc) -> List[tuple]:
    """(offset, numel) of every gradient the reference's `len(parms.grad.size()) == 4` test selects, in the encoder's flat
    gradient buffer: all conv weights of a ResNet-18 encoder, the 16x16 patch-embedding conv of the CAV-MAE audio encoder,
    nothing for the M3AE text / image encoders (their parameters are 1-D / 2-D / 3-D)."""
    segs = []
    for (name, p, _gv), (off, n) in zip(enc._entries, enc.segments()):
        if p.dim() == 4:
            segs.append((off, n))
    return segs


class _Plan:
    def __init__(self, enc, device):
        segs = conv_grad_segments(enc)
        self.n_seg = len(segs)
        if not segs:
            return
        chunk = ops.ogm_chunk_elems()
        first = [0]
        for _o, n in segs:
            first.append(first[-1] + (n + chunk - 1) // chunk)
        self.total_chunks = first[-1]
        self.seg = torch.tensor(segs, dtype=torch.int64, device=device)
        self.first = torch.tensor(first, dtype=torch.int32, device=device)
        self.ws = torch.empty(ops.ogm_ws_bytes(self.total_chunks, self.n_seg), dtype=torch.uint8, device=device)


class OGM:
    """`mode`: "OGM" or "OGM_GE" (args.modulation); `alpha`: args.alpha (main.py:38)."""

    def __init__(self, alpha: float = 0.3, mode: str = "OGM_GE", seed: int = 0, device="cuda"):
        if mode not in ("OGM", "OGM_GE"):
            raise ValueError("mode must be 'OGM' or 'OGM_GE'")
        self.alpha, self.mode, self.seed = float(alpha), mode, int(seed)
        self.device = torch.device(device)
        self.coeff = torch.ones(3, device=self.device, dtype=torch.float32)
        self.info = torch.zeros(6, device=self.device, dtype=torch.float32)       # scores [0:3], ratios [3:6]
        self.step = 0
        self._plans: Dict[int, _Plan] = {}

    def coefficients(self, outs: Sequence[torch.Tensor], label: torch.Tensor) -> torch.Tensor:
        """outs = (out_a, out_v) or (out_a, out_v, out_t) logits (B, C).  Returns the device tensor coeff[:M] (no host sync);
        scores / ratios for logging (main.py:386-390) are in `self.info`."""
        if len(outs) not in (2, 3):
            raise MLAHipError("OGM needs two or three modalities")
        ops.ogm_coeff([o.detach().contiguous() for o in outs], label, self.alpha, self.coeff, self.info)
        return self.coeff[:len(outs)]

    def modulate(self, encoders: Sequence, epoch: int = 0, modulation_starts: int = 0, modulation_ends: int = 50) -> None:
        """Scale (+ noise) the conv gradients of `encoders[m]` by coeff[m] (main.py:392-408).  Call after backward and
        before optimizer.step(); a no-op outside [modulation_starts, modulation_ends] like the reference."""
        if not (modulation_starts <= epoch <= modulation_ends):
            return
        for m, enc in enumerate(encoders):
            plan = self._plans.get(id(enc))
            if plan is None:
                plan = self._plans[id(enc)] = _Plan(enc, self.device)
            if plan.n_seg == 0:
                continue
            if hasattr(enc, "_await_tail"):
                enc._await_tail()
            works = getattr(enc, "_grad_works", None)
            if works:                       # data-parallel protocol path: the encoder-gradient all-reduce is still in flight
                enc.comm.wait(works)        # (autograd.py leaves it to FusedSGD.step()); modulate the REDUCED gradient
                enc._grad_works = []
            # Philox stream = (call counter, modality, tensor): the reference draws a fresh normal_() per tensor (main.py:399-407),
            # so conv k of the audio and of the visual ResNet-18 must not share a noise sequence
            ops.ogm_modulate(enc.grad, plan.seg, plan.first, plan.n_seg, plan.total_chunks, self.coeff[m:m + 1],
                             self.mode == "OGM_GE", self.seed, self.step * 8 + m, plan.ws if self.mode == "OGM_GE" else None)
        self.step += 1
