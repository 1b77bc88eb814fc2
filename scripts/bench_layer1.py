"""64 -> 64 channel 3x3 forward (+ BatchNorm statistics) and input gradient (+ residual, ReLU mask, fused BatchNorm-backward reduction)
at the CREMA-D layer1 shapes, as the training step launches them: the driver of the PMC passes over the persistent patch kernel."""
import os, sys
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


which = sys.argv[1:] or ["audio", "visual"]
for name, (N, H, W) in (("audio", (64, 256, 32)), ("visual", (192, 56, 56))):
    if name not in which:
        continue
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((N, H, W, 64), device="cuda", generator=g)
    w = torch.randn((3, 3, 64, 64), device="cuda", generator=g) * 0.05
    dy = torch.randn((N, H, W, 64), device="cuda", generator=g)
    res = torch.randn((N, H, W, 64), device="cuda", generator=g)
    msk = torch.randn((N, H, W, 64), device="cuda", generator=g)
    z = torch.randn((N, H, W, 64), device="cuda", generator=g)
    mean, invstd = torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")
    wT, wS = ops.conv2d_wsplit(w, True), ops.conv2d_wsplit(w, False)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    part = torch.zeros(ops.conv2d_fwd_partial_elems(N, H, W, 64, 64, 3, 3, 1, 1), device="cuda")
    rpart = torch.zeros(ops.conv2d_dgrad_bn_partial_elems(N, H, W, 64), device="cuda")
    fns = {
        "fwd+stats": lambda: ops.conv2d_fwd_split(x, wT, w.shape, 1, 1, y=y, bn_partial=part),
        "dgrad+res+mask+bn": lambda: ops.conv2d_dgrad_split(dy, wS, w.shape, x.shape, 1, 1, dx=dx, residual=res, relu_src=msk,
                                                             bn_reqs=[(z, mean, invstd, rpart)]),
    }
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    flop = 2.0 * N * H * W * 64 * 64 * 9
    for k, f in fns.items():
        t = min(timed(f) for _ in range(3))
        print(f"{name:7s} {k:18s} {t:8.1f} us  {flop / t / 1e6:7.1f} TFLOP/s", flush=True)
