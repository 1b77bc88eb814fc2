"""Thin tensor-level wrappers over the C ABI (include/mla_hip.h).

Every function takes CUDA(HIP) fp32 contiguous tensors, enqueues on `stream` (a raw
hipStream_t handle; default = torch's current stream) and returns immediately.
Activations are NHWC, conv weights HWIO.  No fallbacks: errors raise MLAHipError.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import MLAHipError, check

BN_EPS = 1e-5        # nn.BatchNorm2d default (models/backbone.py:22)
BN_MOMENTUM = 0.1


class KernelTimer:
    """Optional HIP-event bracket around kernel launches (bench.py roofline).  Events are recorded on
    the stream the kernels are launched on (torch's current stream), so elapsed_time is the launch's
    device duration.  Off (None) in normal operation: zero overhead."""

    def __init__(self):
        self.records = []          # (kind, algorithmic work, start event, end event)

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, kind: str, work: float, start, moved: float = 0.0) -> None:
        """work = algorithmic FLOPs / bytes of the call; moved = bytes the kernels really move (HBM family)."""
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.records.append((kind, work, start, e, moved))

    def summary(self) -> dict:
        """kind -> {'launches', 'ms', 'ms_raw', 'work', 'moved', 'stalls'} (call after torch.cuda.synchronize()).
        The same call (kind, work) repeats every step; an elapsed time above 4x the median of its repeats and more than 1 ms
        over it is a host / profiler stall between the two event records, not kernel time (seen under rocprofv3: one 100 ms
        buffer flush inside a 0.2 ms bracket): it is replaced by that median and counted in 'stalls'."""
        groups: dict = {}
        for kind, work, s, e, moved in self.records:
            groups.setdefault((kind, work), []).append(s.elapsed_time(e))
        med = {k: sorted(v)[len(v) // 2] for k, v in groups.items()}
        out: dict = {}
        for kind, work, s, e, moved in self.records:
            d = out.setdefault(kind, {"launches": 0, "ms": 0.0, "ms_raw": 0.0, "work": 0.0, "moved": 0.0, "stalls": 0})
            t, m = s.elapsed_time(e), med[(kind, work)]
            d["ms_raw"] += t                       # as measured, stalled brackets included (reported beside the corrected sum)
            if len(groups[(kind, work)]) >= 3 and t > 4.0 * m and t > m + 1.0:
                t = m
                d["stalls"] += 1
            d["launches"] += 1
            d["ms"] += t
            d["work"] += work
            d["moved"] += moved
        return out


TIMER: Optional[KernelTimer] = None


def cur_stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor], dtype=torch.float32) -> Optional[int]:
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise MLAHipError(f"expected a contiguous {dtype} device tensor, got {t.dtype} {t.device} "
                          f"contiguous={t.is_contiguous()}")
    return t.data_ptr()


def conv_out(n: int, k: int, s: int, p: int) -> int:
    return (n + 2 * p - k) // s + 1


# ---- layout -------------------------------------------------------------------------------------
def video_to_nhwc(src: torch.Tensor, dst: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> torch.Tensor:
    B, C, T, H, W = src.shape
    if dst is None:
        dst = torch.empty((B * T, H, W, C), device=src.device, dtype=torch.float32)
    check(_lib.load().mla_video_to_nhwc(_p(src), _p(dst), B, C, T, H, W, stream or cur_stream()), "mla_video_to_nhwc")
    return dst


def nchw_to_nhwc(src: torch.Tensor, stream: Optional[int] = None) -> torch.Tensor:
    N, C, H, W = src.shape
    dst = torch.empty((N, H, W, C), device=src.device, dtype=torch.float32)
    check(_lib.load().mla_nchw_to_nhwc(_p(src), _p(dst), N, C, H, W, stream or cur_stream()), "mla_nchw_to_nhwc")
    return dst


def nhwc_to_nchw(src: torch.Tensor, stream: Optional[int] = None) -> torch.Tensor:
    N, H, W, C = src.shape
    dst = torch.empty((N, C, H, W), device=src.device, dtype=torch.float32)
    check(_lib.load().mla_nhwc_to_nchw(_p(src), _p(dst), N, C, H, W, stream or cur_stream()), "mla_nhwc_to_nchw")
    return dst


# ---- convolution --------------------------------------------------------------------------------
def conv2d_f32_cfg(cfg: int = -1) -> int:
    """Measurement hook: force the fp32 conv kernels' tile (0..3) or restore the automatic choice (-1)."""
    return int(_lib.load().mla_conv2d_f32_cfg(int(cfg)))


def conv2d_fwd_partial_elems(N, H, W, Cin, Cout, KH, KW, stride, pad) -> int:
    return int(_lib.load().mla_conv2d_fwd_partial_elems(N, H, W, Cin, Cout, KH, KW, stride, pad))


def conv2d_fwd(x: torch.Tensor, w_hwio: torch.Tensor, stride: int, pad: int, y: Optional[torch.Tensor] = None,
               bn_partial: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> Tuple[torch.Tensor, int]:
    """Returns (y, tiles).  If `bn_partial` is given it receives [tiles][2][Cout] column sums / sums of squares."""
    N, H, W, Cin = x.shape
    KH, KW, Cin2, Cout = w_hwio.shape
    if Cin2 != Cin:
        raise MLAHipError(f"conv2d_fwd: x has {Cin} channels, weight expects {Cin2}")
    if y is None:
        y = torch.empty((N, conv_out(H, KH, stride, pad), conv_out(W, KW, stride, pad), Cout), device=x.device,
                        dtype=torch.float32)
    tiles = ctypes.c_int(0)
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_conv2d_fwd(_p(x), _p(w_hwio), _p(y), N, H, W, Cin, Cout, KH, KW, stride, pad,
                                     _p(bn_partial), ctypes.addressof(tiles), stream or cur_stream()), "mla_conv2d_fwd")
    if t0 is not None:
        TIMER.end("conv_fwd", 2.0 * y.numel() * KH * KW * Cin, t0)
    return y, tiles.value


class _BnReduceReq(ctypes.Structure):     # include/mla_hip.h: mla_bn_reduce_req
    _fields_ = [("x", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("invstd", ctypes.c_void_p), ("partial", ctypes.c_void_p)]


def conv2d_dgrad_bn_partial_elems(N: int, H: int, W: int, Cin: int) -> int:
    return int(_lib.load().mla_conv2d_dgrad_bn_partial_elems(N, H, W, Cin))


def _bn_reqs(bn_reqs, x_shape):
    """bn_reqs: sequence of (x, mean, invstd, partial) -- BatchNorm layers whose backward consumes dx (<= 2)."""
    if not bn_reqs:
        return None, 0
    if len(bn_reqs) > 2:
        raise MLAHipError("conv2d_dgrad: at most two BatchNorm reduction requests")
    N, H, W, Cin = x_shape
    need = conv2d_dgrad_bn_partial_elems(N, H, W, Cin)
    arr = (_BnReduceReq * len(bn_reqs))()
    for q, (x, mean, invstd, partial) in enumerate(bn_reqs):
        if tuple(x.shape) != tuple(x_shape) or mean.numel() != Cin or invstd.numel() != Cin or partial.numel() < need:
            raise MLAHipError(f"conv2d_dgrad: BatchNorm request {q} does not match dx {tuple(x_shape)} (partial >= {need} floats)")
        arr[q].x, arr[q].mean, arr[q].invstd, arr[q].partial = _p(x), _p(mean), _p(invstd), _p(partial)
    return arr, len(bn_reqs)


def conv2d_dgrad(dy: torch.Tensor, w_hwio: torch.Tensor, x_shape, stride: int, pad: int, wt_ws: torch.Tensor,
                 dx: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                 relu_src: Optional[torch.Tensor] = None, stream: Optional[int] = None, bn_reqs=None,
                 class_mask: int = 0xF, residual_mask: int = 0xF):
    """Input gradient.  With bn_reqs (see _bn_reqs) the epilogue also forms the reduction pass of those BatchNorm
    backwards and the call returns (dx, tiles) for bn_bwd_from_partial.  class_mask / residual_mask: output parity classes
    (bit py * stride + px) to compute / to add `residual` in (include/mla_hip.h: mla_conv2d_dgrad_classes)."""
    N, H, W, Cin = x_shape
    KH, KW, _, Cout = w_hwio.shape
    if dx is None:
        dx = torch.empty((N, H, W, Cin), device=dy.device, dtype=torch.float32)
    if wt_ws.numel() < w_hwio.numel():
        raise MLAHipError("conv2d_dgrad: wt_ws too small")
    arr, nreq = _bn_reqs(bn_reqs, x_shape)
    tiles = ctypes.c_int(0)
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_conv2d_dgrad_classes(_p(dy), _p(w_hwio), _p(dx), N, H, W, Cin, Cout, KH, KW, stride, pad,
                                               _p(residual), _p(relu_src), _p(wt_ws), ctypes.addressof(arr) if nreq else None, nreq,
                                               ctypes.addressof(tiles), class_mask, residual_mask, stream or cur_stream()),
          "mla_conv2d_dgrad_classes")
    if t0 is not None:
        TIMER.end("conv_dgrad", 2.0 * dy.numel() * KH * KW * Cin, t0)
    return (dx, tiles.value) if nreq else dx


def conv2d_wsplit(w_hwio: torch.Tensor, transposed: bool, out: Optional[torch.Tensor] = None,
                  stream: Optional[int] = None) -> torch.Tensor:
    """Three bf16 planes of the conv weights (int16 storage, [3][taps][n][k]) for the split-bf16 kernels:
    transposed=True for conv2d_fwd_split (n=Cout, k=Cin), False for conv2d_dgrad_split (n=Cin, k=Cout)."""
    KH, KW, Cin, Cout = w_hwio.shape
    n = int(_lib.load().mla_conv2d_wsplit_bytes(Cin, Cout, KH, KW)) // 2
    if out is None:
        out = torch.empty(n, device=w_hwio.device, dtype=torch.int16)
    if out.numel() < n or out.dtype != torch.int16:
        raise MLAHipError("conv2d_wsplit: out must hold 3*KH*KW*Cin*Cout int16")
    check(_lib.load().mla_conv2d_wsplit(_p(w_hwio), _p(out, torch.int16), Cin, Cout, KH, KW, int(transposed),
                                        stream or cur_stream()), "mla_conv2d_wsplit")
    return out


def conv2d_wsplit_batch(params: torch.Tensor, wsplit: torch.Tensor, desc: torch.Tensor, total_blocks: int,
                        stream: Optional[int] = None) -> None:
    """One launch that re-splits every conv listed in `desc` (int32 device tensor, n x 8; see include/mla_hip.h)."""
    check(_lib.load().mla_conv2d_wsplit_batch(_p(params), _p(wsplit, torch.int16), _p(desc, torch.int32), desc.shape[0],
                                              int(total_blocks), stream or cur_stream()), "mla_conv2d_wsplit_batch")


def conv2d_fwd_split(x: torch.Tensor, wsplit_t: torch.Tensor, w_shape, stride: int, pad: int,
                     y: Optional[torch.Tensor] = None, bn_partial: Optional[torch.Tensor] = None,
                     stream: Optional[int] = None) -> Tuple[torch.Tensor, int]:
    """conv2d_fwd on the split-bf16 MFMA path; `wsplit_t` = conv2d_wsplit(w, True), w_shape = (KH, KW, Cin, Cout)."""
    N, H, W, Cin = x.shape
    KH, KW, Cin2, Cout = w_shape
    if Cin2 != Cin:
        raise MLAHipError(f"conv2d_fwd_split: x has {Cin} channels, weight expects {Cin2}")
    if y is None:
        y = torch.empty((N, conv_out(H, KH, stride, pad), conv_out(W, KW, stride, pad), Cout), device=x.device,
                        dtype=torch.float32)
    tiles = ctypes.c_int(0)
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_conv2d_fwd_split(_p(x), _p(wsplit_t, torch.int16), _p(y), N, H, W, Cin, Cout, KH, KW, stride,
                                           pad, _p(bn_partial), ctypes.addressof(tiles), stream or cur_stream()),
          "mla_conv2d_fwd_split")
    if t0 is not None:
        TIMER.end("conv_fwd", 2.0 * y.numel() * KH * KW * Cin, t0)
    return y, tiles.value


def conv2d_dgrad_split(dy: torch.Tensor, wsplit: torch.Tensor, w_shape, x_shape, stride: int, pad: int,
                       dx: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                       relu_src: Optional[torch.Tensor] = None, stream: Optional[int] = None, bn_reqs=None,
                       class_mask: int = 0xF, residual_mask: int = 0xF):
    """conv2d_dgrad on the split-bf16 MFMA path; `wsplit` = conv2d_wsplit(w, False).  bn_reqs as in conv2d_dgrad."""
    N, H, W, Cin = x_shape
    KH, KW, _, Cout = w_shape
    if dx is None:
        dx = torch.empty((N, H, W, Cin), device=dy.device, dtype=torch.float32)
    arr, nreq = _bn_reqs(bn_reqs, x_shape)
    tiles = ctypes.c_int(0)
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_conv2d_dgrad_split_classes(_p(dy), _p(wsplit, torch.int16), _p(dx), N, H, W, Cin, Cout, KH, KW, stride,
                                                     pad, _p(residual), _p(relu_src), ctypes.addressof(arr) if nreq else None, nreq,
                                                     ctypes.addressof(tiles), class_mask, residual_mask, stream or cur_stream()),
          "mla_conv2d_dgrad_split_classes")
    if t0 is not None:
        TIMER.end("conv_dgrad", 2.0 * dy.numel() * KH * KW * Cin, t0)
    return (dx, tiles.value) if nreq else dx


def conv2d_bnfold_supported(N: int, H: int, W: int, Cin: int, Cout: int, KH: int, KW: int, stride: int, pad: int) -> bool:
    """True where relu(bn(.)) can be folded into the operands of a convolution (64 -> 64 channels, 3x3 / 1 / 1, split arithmetic)."""
    return bool(_lib.load().mla_conv2d_bnfold_supported(N, H, W, Cin, Cout, KH, KW, stride, pad))


def conv2d_fwd_split_bnin(x: torch.Tensor, wsplit_t: torch.Tensor, w_shape, stride: int, pad: int, bn_in, y: Optional[torch.Tensor] = None,
                          bn_partial: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> Tuple[torch.Tensor, int]:
    """conv2d_fwd_split over relu(bn(x)); bn_in = (mean, invstd, gamma, beta) per input channel.  x is the BatchNorm's input."""
    N, H, W, Cin = x.shape
    KH, KW, Cin2, Cout = w_shape
    if Cin2 != Cin:
        raise MLAHipError(f"conv2d_fwd_split_bnin: x has {Cin} channels, weight expects {Cin2}")
    if y is None:
        y = torch.empty((N, conv_out(H, KH, stride, pad), conv_out(W, KW, stride, pad), Cout), device=x.device, dtype=torch.float32)
    tiles = ctypes.c_int(0)
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_conv2d_fwd_split_bnin(_p(x), _p(wsplit_t, torch.int16), _p(y), N, H, W, Cin, Cout, KH, KW, stride, pad,
                                                _p(bn_in[0]), _p(bn_in[1]), _p(bn_in[2]), _p(bn_in[3]), _p(bn_partial),
                                                ctypes.addressof(tiles), stream or cur_stream()), "mla_conv2d_fwd_split_bnin")
    if t0 is not None:
        TIMER.end("conv_fwd", 2.0 * y.numel() * KH * KW * Cin, t0)
    return y, tiles.value


def conv2d_wgrad_split_bnin(x: torch.Tensor, dy: torch.Tensor, dw_hwio: torch.Tensor, stride: int, pad: int, ws: torch.Tensor, bn_in,
                            stream: Optional[int] = None) -> torch.Tensor:
    """conv2d_wgrad_split with relu(bn(x)) as the input operand; bn_in = (mean, invstd, gamma, beta)."""
    N, H, W, Cin = x.shape
    KH, KW, _, Cout = dw_hwio.shape
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_conv2d_wgrad_split_bnin(_p(x), _p(dy), _p(dw_hwio), N, H, W, Cin, Cout, KH, KW, stride, pad, _p(bn_in[0]),
                                                  _p(bn_in[1]), _p(bn_in[2]), _p(bn_in[3]), _p(ws), ws.numel() * ws.element_size(),
                                                  stream or cur_stream()), "mla_conv2d_wgrad_split_bnin")
    if t0 is not None:
        TIMER.end("conv_wgrad", 2.0 * dy.numel() * KH * KW * Cin, t0)
    return dw_hwio


def conv2d_dgrad_split_bnmask(dy: torch.Tensor, wsplit: torch.Tensor, w_shape, x_shape, stride: int, pad: int, dx: torch.Tensor, bn_req,
                              mask_gamma: torch.Tensor, mask_beta: torch.Tensor, stream: Optional[int] = None):
    """conv2d_dgrad_split whose ReLU mask is relu(bn(bn_req.x)) > 0 -- the BatchNorm whose backward reduction the epilogue forms anyway
    (bn_req = (x, mean, invstd, partial)).  Returns (dx, tiles)."""
    N, H, W, Cin = x_shape
    KH, KW, _, Cout = w_shape
    arr, nreq = _bn_reqs([bn_req], x_shape)
    tiles = ctypes.c_int(0)
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_conv2d_dgrad_split_bnmask(_p(dy), _p(wsplit, torch.int16), _p(dx), N, H, W, Cin, Cout, KH, KW, stride, pad,
                                                    ctypes.addressof(arr), nreq, ctypes.addressof(tiles), _p(mask_gamma), _p(mask_beta),
                                                    stream or cur_stream()), "mla_conv2d_dgrad_split_bnmask")
    if t0 is not None:
        TIMER.end("conv_dgrad", 2.0 * dy.numel() * KH * KW * Cin, t0)
    return dx, tiles.value


def conv2d_stem_supported(Cin: int, Cout: int, KH: int, KW: int, stride: int, pad: int) -> bool:
    """True where the persistent split-arithmetic stem kernels apply (7x7 / 2 / 3, 1 or 3 -> 64 channels)."""
    return bool(_lib.load().mla_conv2d_stem_supported(Cin, Cout, KH, KW, stride, pad))


def conv2d_stem_waves(waves: int = -1) -> int:
    """Measurement hook: force 4 or 8 waves per stem-forward workgroup; 0 = automatic; -1: query."""
    return int(_lib.load().mla_conv2d_stem_waves(int(waves)))


def conv2d_stem_fwd_partial_elems() -> int:
    return int(_lib.load().mla_conv2d_stem_fwd_partial_elems())


def conv2d_stem_fwd_split(x: torch.Tensor, w_hwio: torch.Tensor, y: Optional[torch.Tensor] = None,
                          bn_partial: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> Tuple[torch.Tensor, int]:
    """The stem convolution (backbone.py:79-83, 149) on the split arithmetic, persistent patch-loader kernel; plain fp32 HWIO
    weights.  Returns (y, partial rows) like conv2d_fwd."""
    N, H, W, Cin = x.shape
    KH, KW, Cin2, Cout = w_hwio.shape
    if Cin2 != Cin:
        raise MLAHipError(f"conv2d_stem_fwd_split: x has {Cin} channels, weight expects {Cin2}")
    if y is None:
        y = torch.empty((N, conv_out(H, KH, 2, 3), conv_out(W, KW, 2, 3), Cout), device=x.device, dtype=torch.float32)
    if bn_partial is not None and bn_partial.numel() < conv2d_stem_fwd_partial_elems():
        raise MLAHipError("conv2d_stem_fwd_split: bn_partial too small")
    tiles = ctypes.c_int(0)
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_conv2d_stem_fwd_split(_p(x), _p(w_hwio), _p(y), N, H, W, Cin, Cout, KH, KW, 2, 3, _p(bn_partial),
                                                ctypes.addressof(tiles), stream or cur_stream()), "mla_conv2d_stem_fwd_split")
    if t0 is not None:
        TIMER.end("stem_fwd", 2.0 * y.numel() * KH * KW * Cin, t0)
    return y, tiles.value


def conv2d_stem_wgrad_split_ws_bytes(Cin: int) -> int:
    return int(_lib.load().mla_conv2d_stem_wgrad_split_ws_bytes(Cin))


def conv2d_stem_wgrad_split(x: torch.Tensor, dy: torch.Tensor, dw_hwio: torch.Tensor, stride: int, pad: int, ws: torch.Tensor,
                            stream: Optional[int] = None) -> torch.Tensor:
    """Stem weight gradient on the split arithmetic (same call shape as conv2d_wgrad)."""
    N, H, W, Cin = x.shape
    KH, KW, _, Cout = dw_hwio.shape
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_conv2d_stem_wgrad_split(_p(x), _p(dy), _p(dw_hwio), N, H, W, Cin, Cout, KH, KW, stride, pad,
                                                  _p(ws), ws.numel() * ws.element_size(), stream or cur_stream()),
          "mla_conv2d_stem_wgrad_split")
    if t0 is not None:
        TIMER.end("stem_wgrad", 2.0 * dy.numel() * KH * KW * Cin, t0)
    return dw_hwio


def conv2d_wgrad_tr(on: int = -1) -> int:
    """Measurement hook: 0 / 1 = per-tap / persistent all-taps split weight gradient for the 64 -> 64 3x3 convs; -1: query."""
    return int(_lib.load().mla_conv2d_wgrad_tr(int(on)))


def conv2d_patch(on: int = -1) -> int:
    """Measurement hook: 0 / 1 = per-tap gather-GEMM / LDS-patch kernel for the 3x3 stride-1 split forward and input gradient; -1: query."""
    return int(_lib.load().mla_conv2d_patch(int(on)))


def conv2d_dgrad_merge(on: int = -1) -> int:
    """Measurement hook: 0 / 1 = one launch per parity class / all classes in one launch for the stride-2 split input gradient; -1: query."""
    return int(_lib.load().mla_conv2d_dgrad_merge(int(on)))


def conv2d_two_phase(on: int = -1) -> int:
    """Measurement hook: 0 / 1 = single launch / whole rounds of a big tile + one launch for the remaining rows (split gather-GEMM); -1: query."""
    return int(_lib.load().mla_conv2d_two_phase(int(on)))


def conv2d_split_terms(terms: int = 0) -> int:
    """Select (3, 6, 8) or query (anything else) the bf16 product set of the split kernels; 6 = fp32-equivalent."""
    return int(_lib.load().mla_conv2d_split_terms(int(terms)))


def conv2d_split_cfg(cfg: int = -1) -> int:
    """Measurement hook: force the split kernels' tile (0..3) or restore the automatic choice (-1)."""
    return int(_lib.load().mla_conv2d_split_cfg(int(cfg)))


def conv2d_wgrad_ws_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad) -> int:
    return int(_lib.load().mla_conv2d_wgrad_ws_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad))


def conv2d_wgrad(x: torch.Tensor, dy: torch.Tensor, dw_hwio: torch.Tensor, stride: int, pad: int, ws: torch.Tensor,
                 stream: Optional[int] = None) -> torch.Tensor:
    N, H, W, Cin = x.shape
    KH, KW, _, Cout = dw_hwio.shape
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_conv2d_wgrad(_p(x), _p(dy), _p(dw_hwio), N, H, W, Cin, Cout, KH, KW, stride, pad,
                                       _p(ws), ws.numel() * ws.element_size(), stream or cur_stream()), "mla_conv2d_wgrad")
    if t0 is not None:
        TIMER.end("conv_wgrad", 2.0 * dy.numel() * KH * KW * Cin, t0)
    return dw_hwio


def conv2d_wgrad_split_ws_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad) -> int:
    return int(_lib.load().mla_conv2d_wgrad_split_ws_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad))


def conv2d_wgrad_split(x: torch.Tensor, dy: torch.Tensor, dw_hwio: torch.Tensor, stride: int, pad: int, ws: torch.Tensor,
                       stream: Optional[int] = None) -> torch.Tensor:
    """conv2d_wgrad on the split-bf16 MFMA path (Cin a multiple of 64); ws >= conv2d_wgrad_split_ws_bytes."""
    N, H, W, Cin = x.shape
    KH, KW, _, Cout = dw_hwio.shape
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_conv2d_wgrad_split(_p(x), _p(dy), _p(dw_hwio), N, H, W, Cin, Cout, KH, KW, stride, pad,
                                             _p(ws), ws.numel() * ws.element_size(), stream or cur_stream()),
          "mla_conv2d_wgrad_split")
    if t0 is not None:
        TIMER.end("conv_wgrad", 2.0 * dy.numel() * KH * KW * Cin, t0)
    return dw_hwio


# ---- batch norm ---------------------------------------------------------------------------------
def bn_stats_partial_elems(M: int, C: int) -> int:
    return int(_lib.load().mla_bn_stats_partial_elems(M, C))


def bn_stats_partial(x2d: torch.Tensor, M: int, C: int, partial: torch.Tensor, stream: Optional[int] = None) -> int:
    tiles = ctypes.c_int(0)
    check(_lib.load().mla_bn_stats_partial(_p(x2d), M, C, _p(partial), ctypes.addressof(tiles), stream or cur_stream()),
          "mla_bn_stats_partial")
    return tiles.value


def bn_finalize(partial: torch.Tensor, tiles: int, M: int, C: int, mean: torch.Tensor, invstd: torch.Tensor,
                running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor],
                eps: float = BN_EPS, momentum: float = BN_MOMENTUM, stream: Optional[int] = None) -> None:
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_bn_finalize(_p(partial), tiles, M, C, eps, momentum, _p(mean), _p(invstd), _p(running_mean),
                                      _p(running_var), stream or cur_stream()), "mla_bn_finalize")
    if t0 is not None:
        TIMER.end("bn_fwd", 0.0, t0)      # statistics finalize: its time belongs to the BN forward, its bytes are negligible


def bn_apply(x: torch.Tensor, mean, invstd, gamma, beta, out: torch.Tensor, M: int, C: int, relu: bool,
             residual: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> torch.Tensor:
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_bn_apply(_p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(residual), _p(out), M, C,
                                   int(relu), stream or cur_stream()), "mla_bn_apply")
    if t0 is not None:   # BN-fwd-train = 12 B/elem algorithmic; this path moves 8 (+4 with a residual)
        TIMER.end("bn_fwd", 12.0 * M * C, t0, moved=(12.0 if residual is not None else 8.0) * M * C)
    return out


def bn_bwd_ws_elems(M: int, C: int) -> int:
    return int(_lib.load().mla_bn_bwd_ws_elems(M, C))


def bn_bwd(dout, x, mean, invstd, gamma, dx, dgamma, dbeta, ws, M: int, C: int, relu_out=None, g_out=None,
           stream: Optional[int] = None) -> None:
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_bn_bwd(_p(dout), _p(relu_out), _p(x), _p(mean), _p(invstd), _p(gamma), _p(dx), _p(dgamma),
                                 _p(dbeta), _p(g_out), _p(ws), M, C, stream or cur_stream()), "mla_bn_bwd")
    if t0 is not None:   # BN-bwd = 20 B/elem (dy, x for the reductions; dy, x again; write dx)
        TIMER.end("bn_bwd", 20.0 * M * C, t0, moved=20.0 * M * C)


def bn_relu_maxpool_fwd(y, mean, invstd, gamma, beta, out, idx, stream: Optional[int] = None) -> None:
    """maxpool3x3s2(relu(bn(y))) without materialising the ReLU output (stem, backbone.py:150-152)."""
    N, H, W, C = y.shape
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_bn_relu_maxpool_fwd(_p(y), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(out),
                                              _p(idx, torch.uint8), N, H, W, C, stream or cur_stream()),
          "mla_bn_relu_maxpool_fwd")
    if t0 is not None:   # read y once, write the pooled quarter + its index bytes
        by = 4.0 * N * H * W * C + 5.0 * out.numel()
        TIMER.end("bn_fwd", by, t0, moved=by)


def bn_bwd_pooled(dpool, idx, y, mean, invstd, gamma, beta, dy, dgamma, dbeta, ws, stream: Optional[int] = None) -> None:
    """BatchNorm backward fed by the pooled gradient (max-pool scatter + ReLU mask recomputed on the fly)."""
    N, H, W, C = y.shape
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_bn_bwd_pooled(_p(dpool), _p(idx, torch.uint8), _p(y), _p(mean), _p(invstd), _p(gamma), _p(beta),
                                        _p(dy), _p(dgamma), _p(dbeta), _p(ws), N, H, W, C, stream or cur_stream()),
          "mla_bn_bwd_pooled")
    if t0 is not None:   # y twice, dy once, the pooled gradient + index twice
        by = 12.0 * N * H * W * C + 10.0 * dpool.numel()
        TIMER.end("bn_bwd", by, t0, moved=by)


def bn_bwd_from_partial(dout, x, mean, invstd, gamma, dx, dgamma, dbeta, partial, tiles: int, M: int, C: int,
                        stream: Optional[int] = None) -> None:
    """BatchNorm backward whose reduction pass was formed by the input-gradient kernel that wrote `dout` (bn_reqs)."""
    t0 = TIMER.begin() if TIMER is not None else None
    check(_lib.load().mla_bn_bwd_from_partial(_p(dout), _p(x), _p(mean), _p(invstd), _p(gamma), _p(dx), _p(dgamma), _p(dbeta),
                                              _p(partial), tiles, M, C, stream or cur_stream()), "mla_bn_bwd_from_partial")
    if t0 is not None:   # dy, x read once; dx written
        TIMER.end("bn_bwd", 12.0 * M * C, t0, moved=12.0 * M * C)


# ---- pooling ------------------------------------------------------------------------------------
def maxpool_fwd(x: torch.Tensor, y: torch.Tensor, idx: torch.Tensor, stream: Optional[int] = None) -> None:
    N, H, W, C = x.shape
    check(_lib.load().mla_maxpool3x3s2_fwd(_p(x), _p(y), _p(idx, torch.uint8), N, H, W, C, stream or cur_stream()),
          "mla_maxpool3x3s2_fwd")


def maxpool_bwd(dy, idx, dx, x_shape, relu_src=None, stream: Optional[int] = None) -> None:
    N, H, W, C = x_shape
    check(_lib.load().mla_maxpool3x3s2_bwd(_p(dy), _p(idx, torch.uint8), _p(relu_src), _p(dx), N, H, W, C,
                                           stream or cur_stream()), "mla_maxpool3x3s2_bwd")


def avgpool_fwd(x, y, NB: int, P: int, C: int, stream: Optional[int] = None) -> None:
    check(_lib.load().mla_avgpool_fwd(_p(x), _p(y), NB, P, C, stream or cur_stream()), "mla_avgpool_fwd")


def avgpool_bwd(dy, dx, NB: int, P: int, C: int, relu_src=None, stream: Optional[int] = None) -> None:
    check(_lib.load().mla_avgpool_bwd(_p(dy), _p(relu_src), _p(dx), NB, P, C, stream or cur_stream()), "mla_avgpool_bwd")


# ---- head / projection / optimiser ----------------------------------------------------------------
def head_ws_elems(B: int, C: int) -> int:
    return int(_lib.load().mla_head_ws_elems(B, C))


def head_ce_fwd_bwd(X, W, b, labels, logits, loss, dW, db, dX, ws, inv_batch: float, stream: Optional[int] = None) -> None:
    B, D = X.shape
    C = W.shape[0]
    check(_lib.load().mla_head_ce_fwd_bwd(_p(X), _p(W), _p(b), _p(labels, torch.int64), _p(logits), _p(loss), _p(dW),
                                          _p(db), _p(dX), _p(ws), B, D, C, inv_batch, stream or cur_stream()),
          "mla_head_ce_fwd_bwd")


def ce_fwd_bwd(logits, labels, loss, dlogits, ws, inv_batch: float, stream: Optional[int] = None) -> None:
    """nn.CrossEntropyLoss() forward + d logits (main.py:130, 434); ws: B floats."""
    B, C = logits.shape
    check(_lib.load().mla_ce_fwd_bwd(_p(logits), _p(labels, torch.int64), _p(loss), _p(dlogits), _p(ws), B, C, inv_batch,
                                     stream or cur_stream()), "mla_ce_fwd_bwd")


def head_bwd(X, W, dlogits, dW, db, dX, scale: float = 1.0, stream: Optional[int] = None) -> None:
    """autograd of fc_out for a given d logits: dW, db, dX (all times `scale`)."""
    B, D = X.shape
    C = W.shape[0]
    if tuple(dlogits.shape) != (B, C):
        raise MLAHipError(f"head_bwd: dlogits {tuple(dlogits.shape)} does not match ({B}, {C})")
    check(_lib.load().mla_head_bwd(_p(X), _p(W), _p(dlogits), _p(dW), _p(db), _p(dX), B, D, C, scale, stream or cur_stream()),
          "mla_head_bwd")


def scale_by_device_scalar(x, scalar, stream: Optional[int] = None) -> None:
    check(_lib.load().mla_scale_by_device_scalar(_p(x), _p(scalar), x.numel(), stream or cur_stream()),
          "mla_scale_by_device_scalar")


def colsum(X, r, scale: float, stream: Optional[int] = None) -> None:
    B, D = X.shape
    check(_lib.load().mla_colsum(_p(X), _p(r), B, D, scale, stream or cur_stream()), "mla_colsum")


def gs_ws_elems(D: int, C: int) -> int:
    return int(_lib.load().mla_gs_ws_elems(D, C))


def gs_project(Pl, r, G, alpha: float, ws, stream: Optional[int] = None) -> None:
    D = Pl.shape[0]
    C = G.shape[0]
    check(_lib.load().mla_gs_project(_p(Pl), _p(r), _p(G), D, C, alpha, _p(ws), stream or cur_stream()), "mla_gs_project")


def sgd_step(p, g, buf, lr: float, momentum: float, wd: float, first: bool, stream: Optional[int] = None) -> None:
    check(_lib.load().mla_sgd_step(_p(p), _p(g), _p(buf), p.numel(), lr, momentum, wd, int(first),
                                   stream or cur_stream()), "mla_sgd_step")


# ---- transformer encoders (M3AE / CAV-MAE) -----------------------------------------------------------
LN_EPS = 1e-5     # nn.LayerNorm default (models/m3ae.py:138)


def linear_fwd(x, w_kn, bias, y, groups: int, rows: int, K: int, N: int, x_group_rows: Optional[int] = None, x_off: int = 0,
               y_group_rows: Optional[int] = None, y_off: int = 0, residual=None, y_gelu=None, stream: Optional[int] = None,
               wsplit=None):
    """y[g][y_off+r] = x[g][x_off+r] @ w_kn (+bias) (+residual); y_gelu (optional) also receives gelu(y).
    wsplit: conv2d_wsplit(w_kn.view(1, 1, K, N), True) selects the split-bf16 arithmetic."""
    if wsplit is not None:
        check(_lib.load().mla_linear_fwd_split(_p(x), _p(wsplit, torch.int16), _p(bias), _p(residual), _p(y), _p(y_gelu), groups,
                                               rows, x_group_rows or rows, x_off, y_group_rows or rows, y_off, K, N,
                                               stream or cur_stream()), "mla_linear_fwd_split")
        return
    check(_lib.load().mla_linear_fwd(_p(x), _p(w_kn), _p(bias), _p(residual), _p(y), _p(y_gelu), groups, rows,
                                     x_group_rows or rows, x_off, y_group_rows or rows, y_off, K, N,
                                     stream or cur_stream()), "mla_linear_fwd")


def linear_dgrad(dy, w_kn, dx, wt_ws, groups: int, rows: int, K: int, N: int, residual=None, gelu_src=None,
                 stream: Optional[int] = None, wsplit=None):
    """dx = dy @ w_kn^T (+residual) (* gelu'(gelu_src)); dense rows.  wsplit: conv2d_wsplit(w_kn.view(1, 1, K, N), False)."""
    if wsplit is not None:
        check(_lib.load().mla_linear_dgrad_split(_p(dy), _p(wsplit, torch.int16), _p(dx), _p(residual), _p(gelu_src), groups,
                                                 rows, rows, 0, rows, 0, K, N, stream or cur_stream()), "mla_linear_dgrad_split")
        return
    check(_lib.load().mla_linear_dgrad(_p(dy), _p(w_kn), _p(dx), _p(residual), _p(gelu_src), _p(wt_ws), groups, rows,
                                       rows, 0, rows, 0, K, N, stream or cur_stream()), "mla_linear_dgrad")


def linear_wgrad_ws_bytes(M: int, K: int, N: int, split: bool = False) -> int:
    if split:
        return int(_lib.load().mla_linear_wgrad_split_ws_bytes(M, K, N))
    return int(_lib.load().mla_linear_wgrad_ws_bytes(M, K, N))


def linear_wgrad(x, dy, dw_kn, ws, groups: int, rows: int, K: int, N: int, x_group_rows: Optional[int] = None, x_off: int = 0,
                 stream: Optional[int] = None, split: bool = False, dbias: Optional[torch.Tensor] = None):
    """dw_kn = x^T dy (K, N).  split + dbias: the bias gradient (column sums of dy) comes out of the same pass."""
    if split:
        check(_lib.load().mla_linear_wgrad_split_bias(_p(x), _p(dy), _p(dw_kn), _p(dbias), groups, rows, x_group_rows or rows, x_off,
                                                      K, N, _p(ws), ws.numel() * ws.element_size(), stream or cur_stream()),
              "mla_linear_wgrad_split_bias")
        return
    if dbias is not None:
        raise MLAHipError("linear_wgrad: the fused bias gradient exists on the split arithmetic only (use colsum_rows)")
    check(_lib.load().mla_linear_wgrad(_p(x), _p(dy), _p(dw_kn), groups, rows, x_group_rows or rows, x_off, K, N, _p(ws),
                                       ws.numel() * ws.element_size(), stream or cur_stream()), "mla_linear_wgrad")


def colreduce_ws_elems(M: int, C: int) -> int:
    return int(_lib.load().mla_colreduce_ws_elems(M, C))


def colsum_rows(x, out, ws, M: int, C: int, stream: Optional[int] = None):
    check(_lib.load().mla_colsum_rows(_p(x), _p(out), _p(ws), M, C, stream or cur_stream()), "mla_colsum_rows")


def layernorm_fwd(x, w, b, y, mean, rstd, M: int, D: int, eps: float = LN_EPS, stream: Optional[int] = None):
    check(_lib.load().mla_layernorm_fwd(_p(x), _p(w), _p(b), _p(y), _p(mean), _p(rstd), M, D, eps, stream or cur_stream()),
          "mla_layernorm_fwd")


def layernorm_bwd(dy, x, w, mean, rstd, dx, dw, db, ws, M: int, D: int, add=None, stream: Optional[int] = None):
    check(_lib.load().mla_layernorm_bwd(_p(dy), _p(x), _p(w), _p(mean), _p(rstd), _p(add), _p(dx), _p(dw), _p(db), _p(ws),
                                        M, D, stream or cur_stream()), "mla_layernorm_bwd")


def bgemm(A, B, C, batches: int, heads: int, M: int, N: int, K: int, a_strides, b_strides, c_strides, alpha: float = 1.0,
          a_off: int = 0, b_off: int = 0, c_off: int = 0, stream: Optional[int] = None):
    """C[z] = alpha * A[z] @ B[z]; strides in elements (batch, head, row, col-or-k); *_off: element offsets into the buffers."""
    L4 = ctypes.c_long * 4
    sa, sb, sc = L4(*a_strides), L4(*b_strides), L4(*c_strides)      # keep alive across the call (no temporaries!)
    pa, pb, pc = _p(A) + 4 * a_off, _p(B) + 4 * b_off, _p(C) + 4 * c_off
    if min(a_off, b_off, c_off) < 0 or a_off >= A.numel() or b_off >= B.numel() or c_off >= C.numel():
        raise MLAHipError("bgemm: element offset outside its buffer")
    check(_lib.load().mla_bgemm(pa, pb, pc, batches, heads, M, N, K, ctypes.addressof(sa), ctypes.addressof(sb),
                                ctypes.addressof(sc), A.numel() - a_off, B.numel() - b_off, C.numel() - c_off, alpha,
                                stream or cur_stream()), "mla_bgemm")


def softmax_fwd(S, pad_mask, B: int, H: int, n: int, stream: Optional[int] = None):
    check(_lib.load().mla_softmax_fwd(_p(S), _p(pad_mask), B, H, n, stream or cur_stream()), "mla_softmax_fwd")


def softmax_bwd(P, dP, B: int, H: int, n: int, stream: Optional[int] = None):
    check(_lib.load().mla_softmax_bwd(_p(P), _p(dP), B, H, n, stream or cur_stream()), "mla_softmax_bwd")


def attention_fwd(qkv, pad_mask, o, lse, B: int, H: int, n: int, hd: int, stream: Optional[int] = None):
    """o = softmax(mask(q k^T * hd^-0.5)) v, fused (models/m3ae.py:102-125); lse (B, H, n) is kept for the backward."""
    check(_lib.load().mla_attention_fwd(_p(qkv), _p(pad_mask), _p(o), _p(lse), B, H, n, hd, stream or cur_stream()),
          "mla_attention_fwd")


def attention_bwd(do, qkv, o, lse, pad_mask, dqkv, dvec, B: int, H: int, n: int, hd: int, stream: Optional[int] = None):
    check(_lib.load().mla_attention_bwd(_p(do), _p(qkv), _p(o), _p(lse), _p(pad_mask), _p(dqkv), _p(dvec), B, H, n, hd,
                                        stream or cur_stream()), "mla_attention_bwd")


def tokens_assemble(x0, table, ids, pos, type_emb, cls, B: int, L: int, D: int, stream: Optional[int] = None):
    V = table.shape[0] if table is not None else 0
    check(_lib.load().mla_tokens_assemble(_p(x0), _p(table), _p(ids, torch.int64), _p(pos), _p(type_emb), _p(cls), B, L, D, V,
                                          stream or cur_stream()), "mla_tokens_assemble")


def tokens_assemble_bwd_ws_bytes(B: int, L: int, D: int) -> int:
    return int(_lib.load().mla_tokens_assemble_bwd_ws_bytes(B, L, D))


def tokens_assemble_bwd(dx0, colsum_all, ids, dcls, dtype, dtable, B: int, L: int, D: int, stream: Optional[int] = None,
                        ws: Optional[torch.Tensor] = None):
    """ws (uint8, >= tokens_assemble_bwd_ws_bytes): scratch of the deterministic embedding scatter (text path only)."""
    V = dtable.shape[0] if dtable is not None else 0
    if dtable is not None and ws is None:
        ws = torch.empty(tokens_assemble_bwd_ws_bytes(B, L, D), device=dx0.device, dtype=torch.uint8)
    check(_lib.load().mla_tokens_assemble_bwd(_p(dx0), _p(colsum_all), _p(ids, torch.int64), _p(dcls), _p(dtype), _p(dtable),
                                              B, L, D, V, _p(ws, torch.uint8) if ws is not None else None,
                                              ws.numel() if ws is not None else 0, stream or cur_stream()), "mla_tokens_assemble_bwd")


def patchify(img, out, P: int = 16, transposed_hw: Optional[tuple] = None, stream: Optional[int] = None):
    """img (B,C,H,W) -> out (B*(H/P)*(W/P), C*P*P).  transposed_hw=(H, W): img is stored (B, W, H) with C == 1."""
    if transposed_hw is None:
        B, C, H, W = img.shape
        tr = 0
    else:
        B, C = img.shape[0], 1
        H, W = transposed_hw
        tr = 1
    check(_lib.load().mla_patchify(_p(img), _p(out), B, C, H, W, P, tr, stream or cur_stream()), "mla_patchify")


# ---- evaluation path ------------------------------------------------------------------------------------
def head_logits(X, W, b, logits, stream: Optional[int] = None):
    B, D = X.shape
    check(_lib.load().mla_head_logits(_p(X), _p(W), _p(b), _p(logits), B, D, W.shape[0], stream or cur_stream()), "mla_head_logits")


def eval_fuse(outs, labels, counts, weights_out, dynamic: bool, alphas, stream: Optional[int] = None):
    M = len(outs)
    B, C = outs[0].shape
    o = [_p(t) for t in outs] + [None] * (3 - M)
    al = list(alphas) + [0.0] * (3 - len(alphas))
    check(_lib.load().mla_eval_fuse(o[0], o[1], o[2], _p(labels, torch.int64), _p(counts, torch.int32), _p(weights_out), M, B, C,
                                    int(dynamic), al[0], al[1], al[2], stream or cur_stream()), "mla_eval_fuse")


def bn_invstd(var, invstd, eps: float = BN_EPS, stream: Optional[int] = None):
    check(_lib.load().mla_bn_invstd(_p(var), _p(invstd), var.numel(), eps, stream or cur_stream()), "mla_bn_invstd")


# ---- OGM / OGM-GE gradient modulation (main.py:312-410) ---------------------------------------------------------------------
def ogm_coeff(outs, labels, alpha: float, coeff, info=None, stream: Optional[int] = None):
    M = len(outs)
    B, C = outs[0].shape
    ptrs = [_p(t) for t in outs] + [None] * (3 - M)
    check(_lib.load().mla_ogm_coeff(ptrs[0], ptrs[1], ptrs[2], _p(labels, torch.int64), M, B, C, alpha, _p(coeff), _p(info),
                                    stream or cur_stream()), "mla_ogm_coeff")


def ogm_chunk_elems() -> int:
    return int(_lib.load().mla_ogm_chunk_elems())


def ogm_ws_bytes(total_chunks: int, n_seg: int) -> int:
    return int(_lib.load().mla_ogm_ws_bytes(total_chunks, n_seg))


def ogm_modulate(grad, seg_desc, first_chunk, n_seg: int, total_chunks: int, coeff, ge: bool, seed: int, step: int, ws,
                 stream: Optional[int] = None):
    check(_lib.load().mla_ogm_modulate(_p(grad), _p(seg_desc, torch.int64), _p(first_chunk, torch.int32), n_seg, total_chunks,
                                       _p(coeff), int(ge), seed, step, _p(ws, torch.uint8) if ws is not None else None,
                                       ws.numel() if ws is not None else 0, stream or cur_stream()), "mla_ogm_modulate")
