"""One stem kernel variant in a loop (for PMC passes): python scripts/bench_stem_one.py {audio|visual} {fwd|wgrad} [waves]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
from mla_hip import ops  # noqa: E402
which, kind = sys.argv[1], sys.argv[2]
if len(sys.argv) > 3:
    ops.conv2d_stem_waves(int(sys.argv[3]))
N, H, W, Cin = (64, 1024, 128, 1) if which == "audio" else (192, 224, 224, 3)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((N, H, W, Cin), device="cuda", generator=g)
w = torch.randn((7, 7, Cin, 64), device="cuda", generator=g) * 0.1
OH, OW = ops.conv_out(H, 7, 2, 3), ops.conv_out(W, 7, 2, 3)
y = torch.empty((N, OH, OW, 64), device="cuda")
dy = torch.randn((N, OH, OW, 64), device="cuda", generator=g)
dw = torch.empty_like(w)
part = torch.zeros(ops.conv2d_stem_fwd_partial_elems(), device="cuda")
ws = torch.empty(ops.conv2d_stem_wgrad_split_ws_bytes(Cin) // 4 + 4, device="cuda")
for _ in range(12):
    if kind == "fwd":
        ops.conv2d_stem_fwd_split(x, w, y=y, bn_partial=part)
    else:
        ops.conv2d_stem_wgrad_split(x, dy, dw, 2, 3, ws)
torch.cuda.synchronize()
