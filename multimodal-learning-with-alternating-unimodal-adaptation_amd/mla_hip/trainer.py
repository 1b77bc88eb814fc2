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

import contextlib
from typing import Optional

import torch

from .streams import distinct_streams

from . import ops
from .dist import Comm
from .optim import FusedSGD
from .plugin import GSPlugin


class MLATrainer:
    def __init__(self, model, lr: float = 1e-3, momentum: float = 0.9, weight_decay: float = 1e-4,
                 gs_mode: str = "as_intended", legacy_zero_grad: bool = False, av_alpha: float = 0.55,
                 comm: Optional[Comm] = None):
        """`model`: AVClassifier (ResNet-18 audio+visual) or M3AEClassifier (text+image); anything exposing
        `mla_encoders()`, `forward(*inputs) -> features` and `fusion_module.fc_out`."""
        self.model = model
        self.head = model.fusion_module.fc_out
        self.encoders = model.mla_encoders()                    # [(tag, group, encoder)], alternation order
        self.gs_plugin = GSPlugin(dim=self.head.in_features, device=model.device, mode=gs_mode)
        groups = {grp: enc for _t, grp, enc in self.encoders}
        groups["head"] = self.head
        self.optimizer = FusedSGD(groups, lr, momentum, weight_decay, legacy_zero_grad)
        self.av_alpha = av_alpha
        self.comm = comm if comm is not None else Comm()
        dev = model.device
        self._colsum = torch.empty(self.head.in_features, device=dev, dtype=torch.float32)
        self._msg = torch.empty(self.head.numel + self.head.in_features + 1, device=dev, dtype=torch.float32)
        self.losses = {k: torch.zeros(1, device=dev, dtype=torch.float32) for k in ["loss"] + ["loss_" + t for t, _g, _e in self.encoders]}
        self.last = {}
        # One HIP stream per encoder carries that encoder's whole chain -- forward, backward, gradient all-reduce, SGD --
        # and a second one its weight-gradient GEMMs; the stream train_step is called on carries only the head path
        # (head forward/backward, packed head exchange, GSPlugin, head SGD) and hands features / feature gradients
        # over with events.  No encoder forward depends on the head or on another encoder (Q7), the next modality only
        # needs the updated head, and the next STEP's first forward only needs that encoder's own SGD, so the encoder
        # chains run beside each other and across step boundaries (the last modality's backward overlaps the next
        # step's first forward) and fill each other's kernel tails.
        self._can_overlap = dev.type == "cuda" and hasattr(model, "forward_split")
        # encoder chains first (they must not share a hardware queue with each other or with the caller's stream), then the
        # weight-gradient side streams, which may double up when the queues run out (streams.py)
        ne = len(self.encoders)
        pool = distinct_streams(2 * ne, dev) if self._can_overlap else []
        self._estreams, self._wstreams = pool[:ne], pool[ne:]
        self.overlap_forward = False
        self.set_overlap(self._can_overlap)
        # Optional HIP-event brackets around the two data-parallel waits of a step (bench.py --gpus N): the packed head exchange
        # (critical path, calling stream) and the wait for an encoder's gradient all-reduce in front of its SGD launch (that
        # encoder's stream).  {"head_exchange": [(start, end)...], "grad_wait": [...]} or None (off: nothing is recorded).
        self.dist_events: Optional[dict] = None

    def join(self) -> None:
        """Make the current stream wait for every encoder chain (parameters, momentum, gradients, BN buffers final).
        Encoders do this themselves when they are used from another stream (forward, state_dict, ...); call it before
        touching optimizer state or raw buffers on the current stream without a device synchronize."""
        if self._estreams and torch.cuda.is_available():
            cur = torch.cuda.current_stream()
            for es in self._estreams:
                cur.wait_stream(es)

    def set_overlap(self, on: bool) -> None:
        """Per-encoder stream pipeline on / off (off: every kernel of the step on the current stream, in program order)."""
        self.join()
        self.overlap_forward = bool(on) and self._can_overlap
        for k, (_t, _g, enc) in enumerate(self.encoders):
            if hasattr(enc, "wgrad_stream"):
                enc.wgrad_stream = self._wstreams[k] if self.overlap_forward else None
            enc.tail_stream = self._estreams[k] if self.overlap_forward else None

    keep_debug = False

    def _phase(self, name: str, enc, feat: torch.Tensor, label: torch.Tensor, inv_batch: float,
               batch_step: int, len_dataloader: int, bstream):
        """One modality phase (main.py:432-442).  Critical path on the current stream: head forward/backward ->
        (data parallel: packed head exchange) -> GSPlugin -> head SGD.  The encoder backward (and its gradient
        all-reduce) is NOT on that path -- the next modality only needs the updated head -- so when `bstream` (the
        encoder's own stream) is given it is enqueued there and runs beside the following phases and the next step's
        forwards.  Returns the all-reduce work handles."""
        logits, loss, dX = self.head.forward_backward(feat, label, inv_batch, slot=name)               # :432-435
        self.last["out_" + name] = logits
        self.losses["loss_" + name].copy_(loss)
        if bstream is None:
            enc.backward_from_pooled(dX, enc._pa)                                             # loss.backward()
            works = self.comm.allreduce_flat_async(enc.grad)                                 # overlaps what follows
        else:
            bstream.wait_stream(torch.cuda.current_stream())                                  # dX (and the forward) are ready
            with torch.cuda.stream(bstream):
                enc.backward_from_pooled(dX, enc._pa)
                works = self.comm.allreduce_flat_async(enc.grad)
        fires = self.gs_plugin.mode == "as_intended" and self.gs_plugin.exp_count != 0
        r_mean = None
        if self.comm.active:
            ops.colsum(feat, self._colsum, inv_batch)
            ev0 = self._ev_begin()
            self.comm.exchange_head(self.head.grad, self._colsum, self.losses["loss_" + name], self._msg)
            self._ev_end("head_exchange", ev0)
            r_mean = self._colsum
        if self.keep_debug:      # test hooks: inputs of the projection (it is ill-conditioned, tests re-evaluate it in fp64)
            self.last[f"head_grad_{name}_raw"] = self.head.weight_grad.clone()
            self.last[f"Pl_before_{name}"] = self.gs_plugin.Pl.clone()
            self.last[f"r_mean_{name}"] = None if r_mean is None else r_mean.clone()
        if fires:
            self.gs_plugin.before_update(self.head, feat, batch_step, len_dataloader, self.gs_plugin.exp_count,
                                         r_mean=r_mean, grad=self.head.weight_grad)           # :437-438
        if self.keep_debug:
            self.last[f"head_grad_{name}"] = self.head.weight_grad.clone()
        opt = self.optimizer
        opt.mark_ready("head")
        opt.step_group("head")                                                                # optimizer.step(): head
        self.gs_plugin.exp_count += 1                                                         # :442
        return works

    def _on(self, stream):
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def _ev_begin(self):
        if self.dist_events is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()                                   # on the current stream = the stream the bracketed work is enqueued on
        return e

    def _ev_end(self, kind: str, start) -> None:
        if start is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.dist_events.setdefault(kind, []).append((start, e))

    def dist_event_ms(self) -> dict:
        """kind -> (count, mean ms) of the recorded brackets (call after a device synchronize); clears them."""
        out = {}
        for kind, pairs in (self.dist_events or {}).items():
            ts = [s.elapsed_time(e) for s, e in pairs]
            out[kind] = (len(ts), sum(ts) / max(len(ts), 1))
        if self.dist_events is not None:
            self.dist_events = {}
        return out

    def train_step(self, *batch):
        """AVClassifier:     train_step(spec, image, label, batch_step, len_dataloader)
        M3AEClassifier:   train_step(token, padding_mask, image, label, batch_step, len_dataloader)
        Modal3Classifier: train_step(token, padding_mask, image, spec, label, batch_step, len_dataloader)
        spec (B,H,W) or (B,1,H,W); image (B,3,T,H,W) / (B,3,256,256); label int64 (B,).  Returns device scalars
        {'loss','loss_a','loss_v'[,'loss_t']} (no host sync; call .item() when needed, main.py:472-476)."""
        *inputs, label, batch_step, len_dataloader = batch
        m, opt = self.model, self.optimizer
        if not getattr(m, "training", True):
            m.train()                                                                         # main.py:135 model.train()
        if len(inputs) == 2:                                                                  # ResNet audio+visual
            spec, image = inputs
            if spec.dim() == 3:
                spec = spec.unsqueeze(1)                                                      # main.py:431
            inputs = (spec.float(), image.float())
        B = label.shape[0]
        inv_batch = 1.0 / (B * self.comm.world)
        opt.zero_grad()                                                                       # main.py:164
        fwd_done = []
        if self.overlap_forward:
            main = torch.cuda.current_stream()
            fwds = m.forward_split(*inputs)                                                   # main.py:424-431 (joint forward, Q7)
            feats = []
            for es, f in zip(self._estreams, fwds):
                es.wait_stream(main)       # inputs are ready; the previous step's head phases have read this encoder's features
                with torch.cuda.stream(es):
                    feats.append(f())      # stream order on `es`: after this encoder's SGD of the previous step
                    ev = torch.cuda.Event()
                    ev.record()
                    fwd_done.append(ev)
        else:
            feats = m.forward_raw(*inputs)
        for k, ((tag, grp, enc), feat) in enumerate(zip(self.encoders, feats)):
            bs = self._estreams[k] if self.overlap_forward else None
            if bs is not None:
                torch.cuda.current_stream().wait_event(fwd_done[k])                           # this encoder's features have landed
            self.last[tag] = feat
            works = self._phase(tag, enc, feat, label, inv_batch, batch_step, len_dataloader, bs)
            opt.mark_ready(grp)
            with self._on(bs):
                ev0 = self._ev_begin() if works else None
                self.comm.wait(works)                                                         # encoder gradients reduced (data parallel)
                self._ev_end("grad_wait", ev0)
                opt.step_group(grp)                                                           # optimizer.step(): this encoder
            if opt.legacy_zero_grad:              # torch 1.8.1: earlier encoders hold zero (not None) grads in later steps (Q6)
                for j in range(k):
                    g2 = self.encoders[j][1]
                    opt.grad_state[g2] = "zero"
                    with self._on(self._estreams[j] if self.overlap_forward else None):
                        opt.step_group(g2)
        opt.drop_grads()                                                                      # main.py:468-470
        t0, t1 = self.encoders[0][0], self.encoders[1][0]
        torch.add(self.losses["loss_" + t0] * self.av_alpha, self.losses["loss_" + t1], alpha=1 - self.av_alpha,
                  out=self.losses["loss"])                                                    # main.py:472 (Q8)
        return self.losses


class Evaluator:
    """`valid()` of the reference under --gs_flag (main.py:486-679): eval-mode encoders, shared head applied to every
    modality, fixed (av_alpha | a/v/t_alpha) or entropy-gated (--dynamic, main.py:65-106) fusion, per-class accuracy
    counters -- kept on the device (one fusion/arg-max kernel per batch instead of the per-sample .cpu() loop of
    main.py:659-676).  Data parallel: the dynamic weights are ONE scalar per modality computed over the whole
    batch (softmax over dim 0), so they depend on the batch composition; the reference evaluates the global batch on GPU 0
    (DataParallel gathers the features, main.py:624-639).  With an active `comm` every rank all-gathers the (B_local, C)
    logits and labels of all ranks (rank order = the order DataParallel scatters / gathers in) on the head communicator
    and runs the fusion kernel on the global batch: every rank holds the same counters and weights as a single process
    evaluating the concatenated batch."""

    def __init__(self, model, dynamic: bool = False, av_alpha: float = 0.5, a_alpha: float = 0.35, v_alpha: float = 0.25,
                 t_alpha: float = 0.4, comm: Optional[Comm] = None):
        self.model = model
        self.comm = comm if comm is not None else Comm()
        self.head = model.fusion_module.fc_out
        self.M = len(model.mla_encoders())
        self.C = self.head.out_features
        self.dynamic = dynamic
        self.alphas = [av_alpha, 1.0 - av_alpha] if self.M == 2 else [a_alpha, v_alpha, t_alpha]      # main.py:647-651
        self.counts = torch.zeros(self.C * (2 + self.M), device=model.device, dtype=torch.int32)
        self.weights = torch.zeros(3, device=model.device, dtype=torch.float32)
        model.eval()                                                                                   # main.py:519

    def reset(self) -> None:
        self.counts.zero_()

    def update(self, *batch):
        """update(spec, image, label) | update(token, padding_mask, image, label) | update(token, pm, image, spec, label)."""
        *inputs, label = batch
        if len(inputs) == 2:
            spec, image = inputs
            if spec.dim() == 3:
                spec = spec.unsqueeze(1)
            inputs = (spec.float(), image.float())
        feats = self.model.forward_raw(*inputs)
        outs = [self.head.logits(f, slot="eval_" + str(k)) for k, f in enumerate(feats)]
        if self.comm.active:                                         # global batch (Q9): (world * B_local, C) in rank order
            g_outs = [self.comm.allgather_rows(o) for o in outs]
            g_label = self.comm.allgather_rows(label.contiguous())
            ops.eval_fuse(g_outs, g_label, self.counts, self.weights, self.dynamic, self.alphas)
        else:
            ops.eval_fuse(outs, label, self.counts, self.weights, self.dynamic, self.alphas)
        return outs

    def result(self):
        """(acc, acc_a, acc_v[, acc_t]) = sum(acc)/sum(num) as in main.py:677-679 (one host sync)."""
        c = self.counts.view(2 + self.M, self.C).sum(dim=1).cpu().tolist()
        num = max(c[0], 1)
        return tuple(x / num for x in c[1:])
