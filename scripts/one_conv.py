"""Run one conv shape repeatedly (for rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
from mla_hip import ops
N, H, W, Cin, Cout, k, s, p = [int(v) for v in sys.argv[1:9]]
what = sys.argv[9] if len(sys.argv) > 9 else "fwd"
x = torch.randn((N, H, W, Cin), device="cuda"); w = torch.randn((k, k, Cin, Cout), device="cuda") * 0.05
y, _ = ops.conv2d_fwd(x, w, s, p); dy = torch.randn_like(y)
wt = torch.empty(w.numel(), device="cuda"); dx = torch.empty_like(x); dw = torch.empty_like(w)
ws = torch.empty(ops.conv2d_wgrad_ws_bytes(N, H, W, Cin, Cout, k, k, s, p) // 4 + 4, device="cuda")
for _ in range(5):
    if what == "fwd": ops.conv2d_fwd(x, w, s, p, y=y)
    elif what == "dgrad": ops.conv2d_dgrad(dy, w, x.shape, s, p, wt, dx=dx)
    else: ops.conv2d_wgrad(x, dy, dw, s, p, ws)
torch.cuda.synchronize()
