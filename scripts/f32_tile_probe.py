"""fp32 igemm: fwd + dgrad time per layer shape for each forced tile (0 128x128, 1 256x64, 2 64x64, 3 128x64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
from mla_hip import ops
B = 64
shapes = [("a.l1", B, 256, 32, 64, 64, 3, 1, 1), ("v.l1", 3 * B, 56, 56, 64, 64, 3, 1, 1), ("v.l2", 3 * B, 28, 28, 128, 128, 3, 1, 1),
          ("v.l3", 3 * B, 14, 14, 256, 256, 3, 1, 1), ("v.l4", 3 * B, 7, 7, 512, 512, 3, 1, 1), ("a.l4", B, 32, 4, 512, 512, 3, 1, 1)]
def timeit(fn, rep=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(rep): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / rep
for rnd in range(2):
  for tag, N, H, W, Cin, Cout, k, s, p in shapes:
    x = torch.randn((N, H, W, Cin), device="cuda"); w = torch.randn((k, k, Cin, Cout), device="cuda") * 0.05
    y, _ = ops.conv2d_fwd(x, w, s, p); dy = torch.randn_like(y); wt = torch.empty(w.numel(), device="cuda"); dx = torch.empty_like(x)
    gf = 2.0 * y.numel() * k * k * Cin / 1e9
    line = f"{tag:5s}"
    for cfg in (-1, 0, 1, 2, 3):
        if cfg >= 0 and Cout % (128 if cfg == 0 else 64) != 0: continue
        ops.conv2d_f32_cfg(cfg)
        tf = timeit(lambda: ops.conv2d_fwd(x, w, s, p, y=y)); td = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, s, p, wt, dx=dx))
        line += f" | cfg {cfg:2d}: {gf/tf:6.1f} / {gf/td:6.1f} TF"
    ops.conv2d_f32_cfg(-1)
    print(line, flush=True)
