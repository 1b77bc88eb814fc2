"""64 -> 64 channel 3x3 weight gradient at the CREMA-D layer1 shapes: persistent all-taps kernel vs per-tap kernel (same process)."""
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

for name, (N, H, W) in (("audio l1", (64, 256, 32)), ("visual l1", (192, 56, 56))):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((N, H, W, 64), device="cuda", generator=g)
    dy = torch.randn((N, H, W, 64), device="cuda", generator=g)
    dw = torch.empty((3, 3, 64, 64), device="cuda")
    ws = torch.empty(ops.conv2d_wgrad_split_ws_bytes(N, H, W, 64, 64, 3, 3, 1, 1) // 4 + 4, device="cuda")
    def run(on):
        def f():
            ops.conv2d_wgrad_tr(on)
            ops.conv2d_wgrad_split(x, dy, dw, 1, 1, ws)
        return f
    fns = {"all-taps": run(1), "per-tap": run(0)}
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(3):
        for k, f in fns.items():
            res[k].append(timed(f))
    flop = 2.0 * N * H * W * 64 * 64 * 9
    for k, v in res.items():
        print(f"{name:10s} {k:9s} min {min(v):8.1f} us  median {sorted(v)[1]:8.1f} us  {flop / min(v) / 1e6:7.1f} TFLOP/s", flush=True)
ops.conv2d_wgrad_tr(1)
