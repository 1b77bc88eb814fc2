"""`torch.ops.mla_hip.*`: the C-ABI launchers registered as PyTorch custom ops.

`torch.library` registrations (dispatch key CUDA = HIP on ROCm) over the ctypes launchers in ops.py: each op allocates
its outputs with torch, enqueues the kernel(s) on the current stream and returns -- no synchronisation, no fallback
implementation for any other dispatch key (calling one with CPU tensors raises torch's "no kernel" error, loudly).
Functional ops return new tensors; ops that update state take it as a mutable argument (`Tensor(a!)`), exactly what the
launcher does.  Layouts are the kernels': activations NHWC, conv weights HWIO, Linear weights [in][out].

The nn.Module / autograd.Function boundary (model.py, autograd.py) drives the same launchers through launch plans with
preallocated workspaces; these ops are the operator-level surface for code that composes the kernels itself.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import ops

_lib = torch.library.Library("mla_hip", "DEF")
_defined: List[str] = []


def _op(schema: str):
    name = schema.split("(", 1)[0]
    _lib.define(schema)

    def deco(fn):
        _lib.impl(name, fn, "CUDA")
        _defined.append(name)
        return fn
    return deco


def op_names() -> List[str]:
    return list(_defined)


def _f32(shape, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(shape, device=like.device, dtype=torch.float32)


# ---- convolution (nn.Conv2d, backbone.py:4-12) -------------------------------------------------------------------
@_op("conv2d_fwd(Tensor x, Tensor w_hwio, int stride, int pad, str math='f32') -> Tensor")
def conv2d_fwd(x, w_hwio, stride, pad, math="f32"):
    if math == "split":
        return ops.conv2d_fwd_split(x, ops.conv2d_wsplit(w_hwio, True), tuple(w_hwio.shape), stride, pad)[0]
    return ops.conv2d_fwd(x, w_hwio, stride, pad)[0]


@_op("conv2d_fwd_stats(Tensor x, Tensor w_hwio, int stride, int pad) -> (Tensor, Tensor, Tensor)")
def conv2d_fwd_stats(x, w_hwio, stride, pad):
    """conv + the BatchNorm batch statistics of its output from the conv epilogue: (y, mean, invstd)."""
    N, H, W, Cin = x.shape
    KH, KW, _, Cout = w_hwio.shape
    part = _f32(ops.conv2d_fwd_partial_elems(N, H, W, Cin, Cout, KH, KW, stride, pad), x)
    y, tiles = ops.conv2d_fwd(x, w_hwio, stride, pad, bn_partial=part)
    M = y.numel() // Cout
    mean, invstd = _f32(Cout, x), _f32(Cout, x)
    ops.bn_finalize(part, tiles, M, Cout, mean, invstd, None, None)
    return y, mean, invstd


@_op("conv2d_dgrad(Tensor dy, Tensor w_hwio, int[] x_shape, int stride, int pad, Tensor? residual=None, Tensor? relu_src=None) -> Tensor")
def conv2d_dgrad(dy, w_hwio, x_shape, stride, pad, residual=None, relu_src=None):
    return ops.conv2d_dgrad(dy, w_hwio, tuple(x_shape), stride, pad, _f32(w_hwio.numel(), dy), residual=residual,
                            relu_src=relu_src)


@_op("conv2d_wgrad(Tensor x, Tensor dy, int[] w_shape, int stride, int pad) -> Tensor")
def conv2d_wgrad(x, dy, w_shape, stride, pad):
    N, H, W, Cin = x.shape
    KH, KW, _, Cout = w_shape
    ws = _f32(ops.conv2d_wgrad_ws_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad) // 4 + 4, x)
    return ops.conv2d_wgrad(x, dy, _f32(tuple(w_shape), x), stride, pad, ws)


# ---- batch norm (+ ReLU + residual add), training mode (backbone.py:29-50) ------------------------------------------
@_op("bn_act_fwd(Tensor y, Tensor mean, Tensor invstd, Tensor gamma, Tensor beta, bool relu, Tensor? residual=None) -> Tensor")
def bn_act_fwd(y, mean, invstd, gamma, beta, relu, residual=None):
    C = y.shape[-1]
    return ops.bn_apply(y, mean, invstd, gamma, beta, torch.empty_like(y), y.numel() // C, C, relu, residual=residual)


@_op("bn_act_bwd(Tensor dout, Tensor y, Tensor mean, Tensor invstd, Tensor gamma) -> (Tensor, Tensor, Tensor)")
def bn_act_bwd(dout, y, mean, invstd, gamma):
    """(dy, dgamma, dbeta); `dout` must already carry the ReLU mask of the layer's output (the producing kernels fuse it)."""
    C = y.shape[-1]
    M = y.numel() // C
    dx, dg, db = torch.empty_like(y), _f32(C, y), _f32(C, y)
    ops.bn_bwd(dout, y, mean, invstd, gamma, dx, dg, db, _f32(ops.bn_bwd_ws_elems(M, C), y), M, C)
    return dx, dg, db


# ---- pooling (backbone.py:88, 152; basic_model.py:56-65) ----------------------------------------------------------------
@_op("maxpool3x3s2_fwd(Tensor x) -> (Tensor, Tensor)")
def maxpool3x3s2_fwd(x):
    N, H, W, C = x.shape
    oh, ow = ops.conv_out(H, 3, 2, 1), ops.conv_out(W, 3, 2, 1)
    y = _f32((N, oh, ow, C), x)
    idx = torch.empty((N, oh, ow, C), device=x.device, dtype=torch.uint8)
    ops.maxpool_fwd(x, y, idx)
    return y, idx


@_op("maxpool3x3s2_bwd(Tensor dy, Tensor idx, int[] x_shape, Tensor? relu_src=None) -> Tensor")
def maxpool3x3s2_bwd(dy, idx, x_shape, relu_src=None):
    dx = _f32(tuple(x_shape), dy)
    ops.maxpool_bwd(dy, idx, dx, tuple(x_shape), relu_src=relu_src)
    return dx


@_op("conv2d_dgrad_bn(Tensor dy, Tensor w_hwio, int[] x_shape, int stride, int pad, Tensor bn_x, Tensor bn_mean, Tensor bn_invstd, "
     "Tensor? residual=None, Tensor? relu_src=None) -> (Tensor, Tensor, int)")
def conv2d_dgrad_bn(dy, w_hwio, x_shape, stride, pad, bn_x, bn_mean, bn_invstd, residual=None, relu_src=None):
    """Input gradient whose epilogue also forms the reduction pass of the BatchNorm backward that consumes dx
    (bn_x = that BatchNorm's input): returns (dx, partial sums, tiles) for bn_bwd_from_partial."""
    N, H, W, Cin = x_shape
    wt_ws = _f32((w_hwio.numel(),), dy)
    part = _f32((ops.conv2d_dgrad_bn_partial_elems(N, H, W, Cin),), dy)
    dx, tiles = ops.conv2d_dgrad(dy, w_hwio, tuple(x_shape), stride, pad, wt_ws, residual=residual, relu_src=relu_src,
                                 bn_reqs=[(bn_x, bn_mean, bn_invstd, part)])
    return dx, part, tiles


@_op("bn_bwd_from_partial(Tensor dout, Tensor x, Tensor mean, Tensor invstd, Tensor gamma, Tensor partial, int tiles) -> (Tensor, Tensor, Tensor)")
def bn_bwd_from_partial(dout, x, mean, invstd, gamma, partial, tiles):
    C = x.shape[-1]
    M = x.numel() // C
    dx = torch.empty_like(x)
    dg, db = _f32((C,), x), _f32((C,), x)
    ops.bn_bwd_from_partial(dout, x, mean, invstd, gamma, dx, dg, db, partial, tiles, M, C)
    return dx, dg, db


@_op("bn_relu_maxpool_fwd(Tensor y, Tensor mean, Tensor invstd, Tensor gamma, Tensor beta) -> (Tensor, Tensor)")
def bn_relu_maxpool_fwd(y, mean, invstd, gamma, beta):
    """maxpool3x3s2(relu(bn(y))) for the stem (backbone.py:150-152) without materialising the ReLU output."""
    N, H, W, C = y.shape
    oh, ow = ops.conv_out(H, 3, 2, 1), ops.conv_out(W, 3, 2, 1)
    out = _f32((N, oh, ow, C), y)
    idx = torch.empty((N, oh, ow, C), device=y.device, dtype=torch.uint8)
    ops.bn_relu_maxpool_fwd(y, mean, invstd, gamma, beta, out, idx)
    return out, idx


@_op("bn_bwd_pooled(Tensor dpool, Tensor idx, Tensor y, Tensor mean, Tensor invstd, Tensor gamma, Tensor beta) -> (Tensor, Tensor, Tensor)")
def bn_bwd_pooled(dpool, idx, y, mean, invstd, gamma, beta):
    """Backward of bn_relu_maxpool_fwd: (dy, dgamma, dbeta)."""
    N, H, W, C = y.shape
    dy = torch.empty_like(y)
    dg, db = _f32((C,), y), _f32((C,), y)
    ws = _f32((ops.bn_bwd_ws_elems(N * H * W, C),), y)
    ops.bn_bwd_pooled(dpool, idx, y, mean, invstd, gamma, beta, dy, dg, db, ws)
    return dy, dg, db


@_op("avgpool_fwd(Tensor x, int groups) -> Tensor")
def avgpool_fwd(x, groups):
    """x (..., C) viewed as (groups, P, C) -> (groups, C): adaptive_avg_pool2d / 3d + flatten."""
    C = x.shape[-1]
    P = x.numel() // (groups * C)
    y = _f32((groups, C), x)
    ops.avgpool_fwd(x, y, groups, P, C)
    return y


@_op("avgpool_bwd(Tensor dy, int[] x_shape, Tensor? relu_src=None) -> Tensor")
def avgpool_bwd(dy, x_shape, relu_src=None):
    NB, C = dy.shape
    dx = _f32(tuple(x_shape), dy)
    ops.avgpool_bwd(dy, dx, NB, dx.numel() // (NB * C), C, relu_src=relu_src)
    return dx


# ---- shared head, cross entropy, projection, optimiser (main.py:432-440; utils/utils.py:24-41) ----------------------------
@_op("head_ce_fwd_bwd(Tensor X, Tensor W, Tensor b, Tensor labels, float inv_batch) -> (Tensor, Tensor, Tensor, Tensor, Tensor)")
def head_ce_fwd_bwd(X, W, b, labels, inv_batch):
    """(logits, loss[1], dW, db, dX)"""
    B, D = X.shape
    C = W.shape[0]
    logits, loss, dW, db, dX = _f32((B, C), X), _f32(1, X), _f32((C, D), X), _f32(C, X), _f32((B, D), X)
    ops.head_ce_fwd_bwd(X, W, b, labels, logits, loss, dW, db, dX, _f32(ops.head_ws_elems(B, C), X), inv_batch)
    return logits, loss, dW, db, dX


@_op("gs_project(Tensor(a!) Pl, Tensor X, Tensor(b!) G, float alpha) -> ()")
def gs_project(Pl, X, G, alpha):
    """GSPlugin.before_update body (utils/utils.py:34-41) on the batch features X (B, D): Pl and G updated in place."""
    D, C = Pl.shape[0], G.shape[0]
    r = _f32(D, X)
    ops.colsum(X, r, 1.0 / X.shape[0])
    ops.gs_project(Pl, r, G, alpha, _f32(ops.gs_ws_elems(D, C), X))


@_op("sgd_step(Tensor(a!) p, Tensor? g, Tensor(b!) buf, float lr, float momentum, float weight_decay, bool first) -> ()")
def sgd_step(p, g, buf, lr, momentum, weight_decay, first):
    ops.sgd_step(p, g, buf, lr, momentum, weight_decay, first)


# ---- transformer rows (models/m3ae.py:65-179; cav_mae.py:86-113) ------------------------------------------------------------
@_op("layernorm_fwd(Tensor x, Tensor w, Tensor b, float eps=1e-5) -> (Tensor, Tensor, Tensor)")
def layernorm_fwd(x, w, b, eps=1e-5):
    M, D = x.shape
    y, mean, rstd = torch.empty_like(x), _f32(M, x), _f32(M, x)
    ops.layernorm_fwd(x, w, b, y, mean, rstd, M, D, eps)
    return y, mean, rstd


@_op("layernorm_bwd(Tensor dy, Tensor x, Tensor w, Tensor mean, Tensor rstd) -> (Tensor, Tensor, Tensor)")
def layernorm_bwd(dy, x, w, mean, rstd):
    M, D = x.shape
    dx, dw, db = torch.empty_like(x), _f32(D, x), _f32(D, x)
    ops.layernorm_bwd(dy, x, w, mean, rstd, dx, dw, db, _f32(ops.colreduce_ws_elems(M, D), x), M, D)
    return dx, dw, db


@_op("linear_fwd(Tensor x, Tensor w_kn, Tensor? bias=None, Tensor? residual=None, bool gelu=False) -> Tensor")
def linear_fwd(x, w_kn, bias=None, residual=None, gelu=False):
    """y = x @ w_kn (+ bias) (+ residual); gelu=True returns gelu(y) (erf form, m3ae.py:77)."""
    M, K = x.shape
    N = w_kn.shape[1]
    y = _f32((M, N), x)
    yg = _f32((M, N), x) if gelu else None
    ops.linear_fwd(x, w_kn, bias, y, 1, M, K, N, residual=residual, y_gelu=yg)
    return yg if gelu else y


@_op("linear_dgrad(Tensor dy, Tensor w_kn, Tensor? residual=None, Tensor? gelu_src=None) -> Tensor")
def linear_dgrad(dy, w_kn, residual=None, gelu_src=None):
    M, N = dy.shape
    K = w_kn.shape[0]
    dx = _f32((M, K), dy)
    ops.linear_dgrad(dy, w_kn, dx, _f32(w_kn.numel(), dy), 1, M, K, N, residual=residual, gelu_src=gelu_src)
    return dx


@_op("linear_wgrad(Tensor x, Tensor dy) -> Tensor")
def linear_wgrad(x, dy):
    M, K = x.shape
    N = dy.shape[1]
    dw = _f32((K, N), x)
    ops.linear_wgrad(x, dy, dw, _f32(ops.linear_wgrad_ws_bytes(M, K, N) // 4 + 4, x), 1, M, K, N)
    return dw


@_op("attention_fwd(Tensor qkv, Tensor? pad_mask, int heads) -> (Tensor, Tensor)")
def attention_fwd(qkv, pad_mask, heads):
    """qkv (B, n, 3*D) -> (o (B, n, D), lse (B, heads, n)): softmax(q k^T * hd^-0.5 masked with -1e7) v (m3ae.py:102-125)."""
    B, n, D3 = qkv.shape
    D = D3 // 3
    o, lse = _f32((B, n, D), qkv), _f32((B, heads, n), qkv)
    ops.attention_fwd(qkv, pad_mask, o, lse, B, heads, n, D // heads)
    return o, lse


@_op("attention_bwd(Tensor do, Tensor qkv, Tensor o, Tensor lse, Tensor? pad_mask, int heads) -> Tensor")
def attention_bwd(do, qkv, o, lse, pad_mask, heads):
    B, n, D3 = qkv.shape
    D = D3 // 3
    dqkv = torch.empty_like(qkv)
    ops.attention_bwd(do, qkv, o, lse, pad_mask, dqkv, _f32((B, heads, n), qkv), B, heads, n, D // heads)
    return dqkv
