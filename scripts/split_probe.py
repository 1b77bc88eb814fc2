"""fp32-MFMA vs split-bf16 gather-GEMM: error against an fp64 reference and TFLOP/s, per ResNet-18 layer shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
import torch.nn.functional as F
from mla_hip import ops

B = int(os.environ.get("B", "64"))
REP = int(os.environ.get("REP", "20"))
ACC = os.environ.get("ACC", "1") == "1"
terms_list = [int(t) for t in os.environ.get("TERMS", "6").split(",")]
cfgs = [int(t) for t in os.environ.get("CFGS", "-1").split(",")]
shapes = [("l1", 3 * B, 56, 56, 64, 64, 3, 1, 1), ("l2.s2", 3 * B, 56, 56, 64, 128, 3, 2, 1), ("l2.ds", 3 * B, 56, 56, 64, 128, 1, 2, 0),
          ("l2", 3 * B, 28, 28, 128, 128, 3, 1, 1), ("l3", 3 * B, 14, 14, 256, 256, 3, 1, 1), ("l4", 3 * B, 7, 7, 512, 512, 3, 1, 1),
          ("a.l1", B, 256, 32, 64, 64, 3, 1, 1), ("a.l4", B, 32, 4, 512, 512, 3, 1, 1)]

def timeit(fn):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REP): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / REP

def err(a, ref):   # max abs error scaled by the rms of the reference, and rms error / rms
    d = (a.double() - ref)
    r = ref.pow(2).mean().sqrt()
    return float(d.abs().max() / r), float(d.pow(2).mean().sqrt() / r)

for tag, N, H, W, Cin, Cout, k, s, p in shapes:
    x = torch.randn((N, H, W, Cin), device="cuda"); w = torch.randn((k, k, Cin, Cout), device="cuda") * 0.05
    y32, _ = ops.conv2d_fwd(x, w, s, p)
    dy = torch.randn_like(y32)
    wt = torch.empty(w.numel(), device="cuda")
    dx32 = ops.conv2d_dgrad(dy, w, x.shape, s, p, wt)
    gf = 2.0 * y32.numel() * k * k * Cin / 1e9
    wsT, wsN = ops.conv2d_wsplit(w, True), ops.conv2d_wsplit(w, False)
    line = f"{tag:6s} gf {gf:6.2f} | f32 fwd {gf/timeit(lambda: ops.conv2d_fwd(x, w, s, p, y=y32)):6.1f} TF dgrad {gf/timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, s, p, wt, dx=dx32)):6.1f} TF"
    if ACC:   # fp64 reference on a sub-batch (memory)
        nb = min(N, 8)
        xr = x[:nb].double().permute(0, 3, 1, 2); wr = w.double().permute(3, 2, 0, 1)
        yr = F.conv2d(xr, wr, stride=s, padding=p).permute(0, 2, 3, 1)
        dyr = dy[:nb].double().permute(0, 3, 1, 2)
        dxr = torch.nn.grad.conv2d_input(xr.shape, wr, dyr, stride=s, padding=p).permute(0, 2, 3, 1)
        e_y32, e_dx32 = err(y32[:nb], yr), err(dx32[:nb], dxr)
        line += f" | err f32 y {e_y32[0]:.2e}/{e_y32[1]:.2e} dx {e_dx32[0]:.2e}/{e_dx32[1]:.2e}"
    ws = torch.empty(max(ops.conv2d_wgrad_ws_bytes(N, H, W, Cin, Cout, k, k, s, p), ops.conv2d_wgrad_split_ws_bytes(N, H, W, Cin, Cout, k, k, s, p)) // 4 + 4, device="cuda")
    dw32 = ops.conv2d_wgrad(x, dy, torch.empty_like(w), s, p, ws).clone()
    dws = ops.conv2d_wgrad_split(x, dy, torch.empty_like(w), s, p, ws).clone()
    t32 = timeit(lambda: ops.conv2d_wgrad(x, dy, dw32, s, p, ws)); tsp = timeit(lambda: ops.conv2d_wgrad_split(x, dy, dws, s, p, ws))
    line += f" | wgrad f32 {gf/t32:6.1f} split {gf/tsp:6.1f} TF maxdiff {float((dws-dw32).abs().max()/dw32.abs().max()):.1e}"
    print(line, flush=True)
    for terms, cfg in [(t, c) for t in terms_list for c in cfgs]:
        if cfg >= 0 and Cout % (128 if cfg in (0, 1) else 64) != 0: continue
        ops.conv2d_split_terms(terms); ops.conv2d_split_cfg(cfg)
        ys, _ = ops.conv2d_fwd_split(x, wsT, w.shape, s, p)
        dxs = ops.conv2d_dgrad_split(dy, wsN, w.shape, x.shape, s, p)
        t_f = timeit(lambda: ops.conv2d_fwd_split(x, wsT, w.shape, s, p, y=ys))
        t_d = timeit(lambda: ops.conv2d_dgrad_split(dy, wsN, w.shape, x.shape, s, p, dx=dxs))
        line = f"       split{terms} cfg {cfg:2d} fwd {gf/t_f:6.1f} TF dgrad {gf/t_d:6.1f} TF"
        if ACC:
            e_y, e_dx = err(ys[:nb], yr), err(dxs[:nb], dxr)
            line += f" | err y {e_y[0]:.2e}/{e_y[1]:.2e} dx {e_dx[0]:.2e}/{e_dx[1]:.2e}"
        print(line, flush=True)
    ops.conv2d_split_terms(6); ops.conv2d_split_cfg(-1)
