"""Stem conv forward / weight gradient at the CREMA-D shapes: persistent split-arithmetic kernels (stem_split.hip) vs the fp32-MFMA
gather-GEMM (conv_igemm.hip), interleaved rounds in one process (cdna_hip_programming.md 5.4 rule 24)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
from mla_hip import ops  # noqa: E402


def timed(fn, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, (N, H, W, Cin) in (("audio", (B, 1024, 128, 1)), ("visual", (3 * B, 224, 224, 3))):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((N, H, W, Cin), device="cuda", generator=g)
    w = torch.randn((7, 7, Cin, 64), device="cuda", generator=g) * 0.1
    OH, OW = ops.conv_out(H, 7, 2, 3), ops.conv_out(W, 7, 2, 3)
    y = torch.empty((N, OH, OW, 64), device="cuda")
    dy = torch.randn((N, OH, OW, 64), device="cuda", generator=g)
    dw = torch.empty_like(w)
    part = torch.zeros(max(ops.conv2d_fwd_partial_elems(N, H, W, Cin, 64, 7, 7, 2, 3), ops.conv2d_stem_fwd_partial_elems()), device="cuda")
    ws = torch.empty(max(ops.conv2d_wgrad_ws_bytes(N, H, W, Cin, 64, 7, 7, 2, 3), ops.conv2d_stem_wgrad_split_ws_bytes(Cin)) // 4 + 4, device="cuda")
    def fwd_w(wv):
        def f():
            ops.conv2d_stem_waves(wv)
            ops.conv2d_stem_fwd_split(x, w, y=y, bn_partial=part)
        return f
    fns = {"fwd split w8": fwd_w(8), "fwd split w4": fwd_w(4),
           "fwd f32": lambda: ops.conv2d_fwd(x, w, 2, 3, y=y, bn_partial=part),
           "wgrad split": lambda: ops.conv2d_stem_wgrad_split(x, dy, dw, 2, 3, ws),
           "wgrad f32": lambda: ops.conv2d_wgrad(x, dy, dw, 2, 3, ws)}
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(3):
        for k, f in fns.items():
            res[k].append(timed(f))
    flop = 2.0 * N * OH * OW * 64 * 49 * Cin
    byts = 4.0 * (x.numel() + y.numel())
    for k, v in res.items():
        t = min(v)
        print(f"{name:7s} {k:12s} min {t:8.1f} us  median {sorted(v)[1]:8.1f} us  {flop / t / 1e6:7.1f} TFLOP/s  {byts / t / 1e3:7.1f} GB/s (x + y bytes)", flush=True)
