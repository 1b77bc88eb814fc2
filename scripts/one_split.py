"""Run one split-bf16 conv shape a few times (for rocprofv3 --pmc): N H W Cin Cout k s p cfg [fwd|dgrad|wgrad]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
from mla_hip import ops
N, H, W, Cin, Cout, k, s, p, cfg = [int(v) for v in sys.argv[1:10]]
what = sys.argv[10] if len(sys.argv) > 10 else "fwd"
x = torch.randn((N, H, W, Cin), device="cuda"); w = torch.randn((k, k, Cin, Cout), device="cuda") * 0.05
ops.conv2d_split_cfg(cfg)
wT, wN = ops.conv2d_wsplit(w, True), ops.conv2d_wsplit(w, False)
y, _ = ops.conv2d_fwd_split(x, wT, w.shape, s, p); dy = torch.randn_like(y)
dx = torch.empty_like(x); dw = torch.empty_like(w)
ws = torch.empty(ops.conv2d_wgrad_split_ws_bytes(N, H, W, Cin, Cout, k, k, s, p) // 4 + 4, device="cuda")
for _ in range(3):
    if what == "fwd": ops.conv2d_fwd_split(x, wT, w.shape, s, p, y=y)
    elif what == "dgrad": ops.conv2d_dgrad_split(dy, wN, w.shape, x.shape, s, p, dx=dx)
    else: ops.conv2d_wgrad_split(x, dy, dw, s, p, ws)
torch.cuda.synchronize()
