"""Per-layer conv microbenchmark at the CREMA-D B=64 shapes: TFLOP/s of fwd / dgrad / wgrad per distinct conv."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
from mla_hip import ops

B = int(os.environ.get("B", "64"))
REP = int(os.environ.get("REP", "10"))
only = sys.argv[1] if len(sys.argv) > 1 else ""
# (tag, N, H, W, Cin, Cout, k, s, p, count per step)
shapes = []
for mod, N, H, W in (("a", B, 1024, 128), ("v", 3 * B, 224, 224)):
    cin = 1 if mod == "a" else 3
    shapes.append((f"{mod}.stem", N, H, W, cin, 64, 7, 2, 3, 1))
    h, w = H // 4, W // 4
    shapes.append((f"{mod}.l1", N, h, w, 64, 64, 3, 1, 1, 4))
    c = 64
    for li, planes in ((2, 128), (3, 256), (4, 512)):
        shapes.append((f"{mod}.l{li}.s2", N, h, w, c, planes, 3, 2, 1, 1))
        shapes.append((f"{mod}.l{li}.ds", N, h, w, c, planes, 1, 2, 0, 1))
        h, w = (h + 1) // 2, (w + 1) // 2
        shapes.append((f"{mod}.l{li}", N, h, w, planes, planes, 3, 1, 1, 3))
        c = planes

def timeit(fn):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REP): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / REP

tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
print(f"{'shape':10s} {'M':>8s} {'N':>4s} {'K':>5s} {'GF':>7s} | {'fwd ms':>7s} {'TF':>6s} | {'dgrad':>7s} {'TF':>6s} | {'wgrad':>7s} {'TF':>6s}")
for tag, N, H, W, Cin, Cout, k, s, p, cnt in shapes:
    if only and only not in tag: continue
    x = torch.randn((N, H, W, Cin), device="cuda")
    w = torch.randn((k, k, Cin, Cout), device="cuda") * 0.05
    y, _ = ops.conv2d_fwd(x, w, s, p)
    part = torch.empty(ops.conv2d_fwd_partial_elems(N, H, W, Cin, Cout, k, k, s, p), device="cuda")
    dy = torch.randn_like(y)
    gf = 2.0 * y.numel() * k * k * Cin / 1e9
    t_f = timeit(lambda: ops.conv2d_fwd(x, w, s, p, y=y, bn_partial=part))
    t_d = float("nan")
    if Cin % 64 == 0:
        wt = torch.empty(w.numel(), device="cuda"); dx = torch.empty_like(x)
        t_d = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, s, p, wt, dx=dx))
    ws = torch.empty(ops.conv2d_wgrad_ws_bytes(N, H, W, Cin, Cout, k, k, s, p) // 4 + 4, device="cuda")
    dw = torch.empty_like(w)
    t_w = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, s, p, ws))
    M = y.numel() // Cout
    print(f"{tag:10s} {M:8d} {Cout:4d} {k*k*Cin:5d} {gf:7.2f} | {t_f:7.3f} {gf/t_f:6.1f} | {t_d:7.3f} {gf/t_d:6.1f} | {t_w:7.3f} {gf/t_w:6.1f}   x{cnt}")
    tot["fwd"] += t_f * cnt; tot["wgrad"] += t_w * cnt
    if t_d == t_d: tot["dgrad"] += t_d * cnt
print("per-step totals (ms):", {k: round(v, 2) for k, v in tot.items()}, "sum", round(sum(tot.values()), 2))
