"""Forward-conv TFLOP/s over a list of N,H,W,Cin,Cout,k,s,p shapes (stdin-free: shapes given as argv 'N,H,W,C,K,k,s,p')."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
from mla_hip import ops
REP = 20
for spec in sys.argv[1:]:
    N, H, W, Cin, Cout, k, s, p = [int(v) for v in spec.split(",")]
    x = torch.randn((N, H, W, Cin), device="cuda"); w = torch.randn((k, k, Cin, Cout), device="cuda") * 0.05
    if os.environ.get("ZERO"): x.zero_(); w.zero_()
    y, _ = ops.conv2d_fwd(x, w, s, p)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REP): ops.conv2d_fwd(x, w, s, p, y=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / REP
    gf = 2.0 * y.numel() * k * k * Cin / 1e9
    print(f"{spec:28s} M={y.numel()//Cout:8d} {ms:7.3f} ms {gf/ms:6.1f} TF", flush=True)
