"""Per-layer time of the three conv contractions at the CREMA-D shapes (B = 64), both arithmetics.  MATH=split|f32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
from mla_hip import ops
from mla_hip.encoder import conv_specs
math = os.environ.get("MATH", "split")
B = int(os.environ.get("B", "64"))
if "MLA_PATCH" in os.environ:                      # same-box A/B switches of the round-3 kernels
    ops.conv2d_patch(int(os.environ["MLA_PATCH"]))
if "MLA_CONV_TWO_PHASE" in os.environ:
    ops.conv2d_two_phase(int(os.environ["MLA_CONV_TWO_PHASE"]))
if "MLA_DGRAD_MERGE" in os.environ:
    ops.conv2d_dgrad_merge(int(os.environ["MLA_DGRAD_MERGE"]))
if "MLA_WGRAD_TR" in os.environ:
    ops.conv2d_wgrad_tr(int(os.environ["MLA_WGRAD_TR"]))


def timeit(fn, rep=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(rep): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / rep * 1e3


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
for mod, (N, H, W) in (("audio", (B, 1024, 128)), ("visual", (B * 3, 224, 224))):
    h, w = H, W
    dims = {}
    for name, cin, cout, k, s, p in conv_specs(mod):
        if name == "conv1":
            ih, iw = H, W
        elif name.endswith("conv1") or name.endswith("downsample.0"):
            ih, iw = dims["in"]
        else:
            ih, iw = dims["mid"]
        oh, ow = ops.conv_out(ih, k, s, p), ops.conv_out(iw, k, s, p)
        if name == "conv1":
            dims["in"] = (ops.conv_out(oh, 3, 2, 1), ops.conv_out(ow, 3, 2, 1))      # after the max-pool
        elif name.endswith("conv1"):
            dims["mid"] = (oh, ow)
        elif name.endswith("conv2"):
            dims["in"] = (oh, ow)
        if cin % 64 != 0:
            continue
        x = torch.randn((N, ih, iw, cin), device="cuda")
        wt = torch.randn((k, k, cin, cout), device="cuda") * 0.05
        y = torch.empty((N, oh, ow, cout), device="cuda")
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        dw = torch.empty_like(wt)
        gf = 2.0 * y.numel() * k * k * cin / 1e9
        if math == "split":
            wT, wS = ops.conv2d_wsplit(wt, True), ops.conv2d_wsplit(wt, False)
            ws = torch.empty(ops.conv2d_wgrad_split_ws_bytes(N, ih, iw, cin, cout, k, k, s, p) // 4 + 4, device="cuda")
            tf = timeit(lambda: ops.conv2d_fwd_split(x, wT, wt.shape, s, p, y=y))
            td = timeit(lambda: ops.conv2d_dgrad_split(dy, wS, wt.shape, x.shape, s, p, dx=dx))
            tw = timeit(lambda: ops.conv2d_wgrad_split(x, dy, dw, s, p, ws))
        else:
            wtw = torch.empty(wt.numel(), device="cuda")
            ws = torch.empty(ops.conv2d_wgrad_ws_bytes(N, ih, iw, cin, cout, k, k, s, p) // 4 + 4, device="cuda")
            tf = timeit(lambda: ops.conv2d_fwd(x, wt, s, p, y=y))
            td = timeit(lambda: ops.conv2d_dgrad(dy, wt, x.shape, s, p, wtw, dx=dx))
            tw = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, s, p, ws))
        tot["fwd"] += tf; tot["dgrad"] += td; tot["wgrad"] += tw
        print(f"{mod[0]}.{name:24s} M={y.numel()//cout:7d} {cin:3d}->{cout:3d} k{k} s{s} {gf:6.1f} GF | fwd {tf:7.1f} us {gf/tf*1e3:6.1f} TF | "
              f"dgrad {td:7.1f} us {gf/td*1e3:6.1f} TF | wgrad {tw:7.1f} us {gf/tw*1e3:6.1f} TF", flush=True)
print({k: round(v / 1e3, 3) for k, v in tot.items()}, "ms per step (non-stem convs)")
