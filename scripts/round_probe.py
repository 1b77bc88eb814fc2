"""Time of 1, 2, 3, 4 full rounds of 256 tiles (split igemm, forced tile): launch ramp vs marginal round."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
from mla_hip import ops
cfg = int(os.environ.get("CFG", "1")); C = int(os.environ.get("C", "128")); k = int(os.environ.get("KS", "3"))
bm = {0: 256, 1: 128, 2: 128, 3: 64, 4: 256}[cfg]; bn = 128 if cfg < 2 else 64
ops.conv2d_split_cfg(cfg)
def timeit(fn, rep=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(rep): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / rep * 1e3
w = torch.randn((k, k, C, C), device="cuda") * 0.05
wT = ops.conv2d_wsplit(w, True)
per_round = 256 * (bn // bn) // (C // bn)     # M tiles per round of 256 workgroups
for rounds in (1, 2, 3, 4, 8):
    M = rounds * per_round * bm
    N_, H, W = M // (32 * 16), 32, 16
    x = torch.randn((N_, H, W, C), device="cuda")
    y, _ = ops.conv2d_fwd_split(x, wT, w.shape, 1, k // 2)
    t = timeit(lambda: ops.conv2d_fwd_split(x, wT, w.shape, 1, k // 2, y=y))
    gf = 2.0 * M * C * k * k * C / 1e9
    print(f"cfg {cfg} C {C} k {k}: {rounds} rounds ({M // bm * (C // bn)} tiles): {t:7.1f} us  {gf / t / 1e3:6.1f} TF", flush=True)
