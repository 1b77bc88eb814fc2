"""autograd.Function wrappers: how `loss.backward()` (main.py:435, 447, 459) reaches the HIP backward plans.

The reference's step is written against autograd: `a, v = model(...)` must return tensors with history,
`fc_out(a)` a differentiable Linear, `criterion(out, label).backward()` must end with `.grad` populated on the
head and on ONE encoder.  Each Function below is a thin shell over the same launch plans `MLATrainer` drives:

  EncoderFeature   forward  = encoder forward + global pooling  -> (B, D) feature (fresh tensor)
                   backward = encoder.backward_from_pooled(d feature) -> flat gradient buffer -> p.grad views
  HeadLinear       forward  = mla_head_logits; backward = mla_head_bwd (dW, db, dX in one launch pair)
  SoftmaxCE        forward  = mla_ce_fwd_bwd (loss + d logits in one launch); backward = scale by the incoming grad

Graph recording is triggered by a private 1-element `anchor` tensor (requires_grad, not a Parameter): the encoder's
62 parameters are NOT autograd inputs, so no AccumulateGrad nodes / gradient copies exist -- the Function installs
the flat-buffer views as `.grad` itself, with autograd's accumulate rule (module.py).  Under `torch.no_grad()`
(evaluation, main.py:520) nothing is recorded and only the forward kernels run.

Data parallel (mla_hip.DataParallel, one process per GPU): HeadLinear.backward scales dW, db and dX by 1/world (the criterion
averaged over the LOCAL batch), so every local gradient -- the encoder's included, through dX -- is a share of the
global-batch mean; it all-reduces the packed (dW|db) before it is published; the encoder gradient all-reduce (SUM) is issued
asynchronously on RCCL's stream and waited for by FusedSGD.step().
"""
from __future__ import annotations

import torch

from . import ops


def make_anchor(device) -> torch.Tensor:
    return torch.zeros(1, device=device, dtype=torch.float32, requires_grad=True)


class EncoderFeature(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, enc, run, B, D):
        out = torch.empty((B, D), device=anchor.device, dtype=torch.float32)
        run(out)                                             # HIP forward plan; pooled feature written into `out`
        ctx.enc = enc
        ctx.token = enc._fwd_token = object()                # one live forward state per encoder (workspace is reused)
        return out

    @staticmethod
    def backward(ctx, dfeat):
        enc = ctx.enc
        if enc._fwd_token is not ctx.token:
            raise RuntimeError("mla_hip: backward through a stale encoder forward (the activation workspace holds the "
                               "most recent forward only; retain_graph / double backward are not supported)")
        comm = enc.comm        # data parallel: d feature already is this rank's share of the global-batch mean (HeadLinear.backward
        pending = enc.grads_pending()          # scales by 1 / world), so the flat encoder gradient only needs the SUM over ranks
        enc.backward_from_pooled(dfeat.contiguous(), enc._pa)
        if comm is not None and comm.active:
            enc._grad_works = comm.allreduce_flat_async(enc.grad)      # waited for by FusedSGD.step()
            if pending is not None:                                    # accumulation: add the older gradient after the exchange
                comm.wait(enc._grad_works)
                enc._grad_works = []
        enc.publish_grads(pending)
        return None, None, None, None, None


class HeadLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, head, x):
        xd = x.detach().contiguous()
        out = torch.empty((xd.shape[0], head.out_features), device=xd.device, dtype=torch.float32)
        ops.head_logits(xd, head.weight.detach(), head.bias.detach(), out)
        ctx.head, ctx.x = head, xd
        return out

    @staticmethod
    def backward(ctx, dlogits):
        head, x = ctx.head, ctx.x
        comm = head.comm
        active = comm is not None and comm.active
        pending = head.grads_pending()
        dX = torch.empty_like(x)
        ops.head_bwd(x, head.weight.detach(), dlogits.contiguous(), head.weight_grad, head.bias_grad, dX,
                     (1.0 / comm.world) if active else 1.0)
        if active:
            comm.allreduce_small(head.grad)                          # critical path: packed dW|db, one message
        head.publish_grads(pending)
        return None, None, dX


class SoftmaxCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        lg = logits.detach().contiguous()
        B, C = lg.shape
        loss = torch.empty(1, device=lg.device, dtype=torch.float32)
        dl = torch.empty_like(lg)
        ws = torch.empty(B, device=lg.device, dtype=torch.float32)
        ops.ce_fwd_bwd(lg, labels, loss, dl, ws, 1.0 / B)
        ctx.dl = dl
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dl = ctx.dl
        ops.scale_by_device_scalar(dl, g.reshape(1).contiguous())   # usually 1.0 (loss.backward()); no host sync
        return dl, None
