"""Driver of the stream-K experiment on the transformer Linear shapes (needs scripts/experiments/r03_streamk.patch applied and
libmla_hip.so rebuilt; run from the repo root).  M3AE / CAV-MAE: M = B x 257 rows; forward and input gradient on the split arithmetic with
the schedule off / by cost model / forced, same process.  Output of round 3: profiles/r03_streamk_linear.txt."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


M = int(os.environ.get("MLA_ROWS", 64 * 257))
ws = torch.empty(ops.conv2d_streamk_ws_bytes() // 4, device="cuda")
ops.conv2d_streamk_workspace(ws)
tot = {0: 0.0, 1: 0.0, 2: 0.0}
for name, K, N, gelu, res in (("qkv", 768, 2304, False, False), ("proj", 768, 768, False, True), ("fc1", 768, 3072, True, False),
                              ("fc2", 3072, 768, False, True)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((M, K), device="cuda", generator=g)
    w = torch.randn((K, N), device="cuda", generator=g) * 0.03
    b = torch.randn((N,), device="cuda", generator=g)
    r = torch.randn((M, N), device="cuda", generator=g) if res else None
    y, yg = torch.empty((M, N), device="cuda"), (torch.empty((M, N), device="cuda") if gelu else None)
    dy = torch.randn((M, N), device="cuda", generator=g)
    dx = torch.empty((M, K), device="cuda")
    gs = torch.randn((M, K), device="cuda", generator=g)
    wT, wS = ops.conv2d_wsplit(w.view(1, 1, K, N), True), ops.conv2d_wsplit(w.view(1, 1, K, N), False)
    fwd = lambda: ops.linear_fwd(x, w, b, y, 1, M, K, N, residual=r, y_gelu=yg, wsplit=wT)
    dgr = lambda: ops.linear_dgrad(dy, w, dx, None, 1, M, K, N, gelu_src=gs if name == "fc2" else None, wsplit=wS)
    flop = 2.0 * M * K * N
    line = f"{name:5s} K={K:4d} N={N:4d} {flop / 1e9:6.1f} GF |"
    for what, f in (("fwd", fwd), ("dgrad", dgr)):
        ts = []
        for mode in (0, 1, 2):
            ops.conv2d_streamk(mode)
            f()
            torch.cuda.synchronize()
            t = min(timed(f) for _ in range(3))
            ts.append(t)
            tot[mode] += t
        line += f" {what} off {ts[0]:7.1f} model {ts[1]:7.1f} forced {ts[2]:7.1f} us ({flop / ts[1] / 1e6:5.1f} TF) |"
    print(line, flush=True)
ops.conv2d_streamk(1)
ops.conv2d_streamk_workspace(None)
print({k: round(v / 1e3, 3) for k, v in tot.items()}, "ms per encoder layer (fwd + dgrad): off / model / forced")
