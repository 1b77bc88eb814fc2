import os, sys, torch
sys.path.insert(0, "multimodal-learning-with-alternating-unimodal-adaptation_amd")
from mla_hip import ops
from mla_hip.encoder import conv_specs
def timeit(fn, rep=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(rep): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / rep * 1e3
names = {-1: "auto", 0: "256x128", 1: "128x128", 2: "128x64", 3: "64x64", 4: "256x64"}
B = 64
tot_auto = tot_best = 0.0
for mod, (N, H, W) in (("a", (B, 1024, 128)), ("v", (B * 3, 224, 224))):
    dims = {}
    seen = set()
    for name, cin, cout, k, s, p in conv_specs("audio" if mod == "a" else "visual"):
        if name == "conv1": ih, iw = H, W
        elif name.endswith("conv1") or name.endswith("downsample.0"): ih, iw = dims["in"]
        else: ih, iw = dims["mid"]
        oh, ow = ops.conv_out(ih, k, s, p), ops.conv_out(iw, k, s, p)
        if name == "conv1": dims["in"] = (ops.conv_out(oh, 3, 2, 1), ops.conv_out(ow, 3, 2, 1))
        elif name.endswith("conv1"): dims["mid"] = (oh, ow)
        elif name.endswith("conv2"): dims["in"] = (oh, ow)
        if cin % 64 != 0: continue
        key = (ih, iw, cin, cout, k, s)
        mult = 1
        if key in seen: continue
        seen.add(key)
        x = torch.randn((N, ih, iw, cin), device="cuda"); wt = torch.randn((k, k, cin, cout), device="cuda") * 0.05
        y = torch.empty((N, oh, ow, cout), device="cuda"); dy = torch.randn_like(y); dx = torch.empty_like(x)
        wT, wS = ops.conv2d_wsplit(wt, True), ops.conv2d_wsplit(wt, False)
        res = {}
        for kind, fn in (("fwd", lambda: ops.conv2d_fwd_split(x, wT, wt.shape, s, p, y=y)), ("dgrad", lambda: ops.conv2d_dgrad_split(dy, wS, wt.shape, x.shape, s, p, dx=dx))):
            r = {}
            for cfg in (-1, 0, 1, 2, 3, 4):
                if cfg >= 0 and (cout if kind == "fwd" else cin) % (128 if cfg <= 1 else 64) != 0: continue
                ops.conv2d_split_cfg(cfg)
                r[cfg] = timeit(fn)
            ops.conv2d_split_cfg(-1)
            best = min((v, c) for c, v in r.items() if c >= 0)
            tot_auto += r[-1]; tot_best += best[0]
            flag = "  <-- auto is %.1f %% slower" % (100 * (r[-1] / best[0] - 1)) if r[-1] > 1.03 * best[0] else ""
            print(f"{mod}.{name:22s} {kind:5s} auto {r[-1]:7.1f} | " + " ".join(f"{names[c]} {v:7.1f}" for c, v in r.items() if c >= 0) + flag)
print(f"sum auto {tot_auto/1e3:.3f} ms, sum best {tot_best/1e3:.3f} ms (unique shapes only)")
