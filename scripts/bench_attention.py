"""Fused attention kernels alone at the config-4 / config-5 shapes: us per call and TFLOP/s (executed products incl. recomputation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
from mla_hip import ops


def timeit(fn, rep=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(rep): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / rep * 1e3


for B, H, n, masked in ((64, 12, 257, True), (64, 12, 257, False), (64, 12, 256, True), (64, 12, 256, False), (64, 12, 288, False), (64, 12, 258, False), (32, 12, 512, False)):
    D = H * 64
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn((B, n, 3 * D), device="cuda", generator=g) * 0.7
    do = torch.randn((B, n, D), device="cuda", generator=g)
    pm = None
    if masked:
        lens = torch.randint(8, n, (B,), device="cuda", generator=g)
        pm = (torch.arange(n, device="cuda")[None, :] >= lens[:, None]).float().contiguous()
        pm[:, 0] = 0
    o, lse = torch.empty((B, n, D), device="cuda"), torch.empty((B, H, n), device="cuda")
    dqkv, dvec = torch.empty_like(qkv), torch.empty((B, H, n), device="cuda")
    unit = 2.0 * B * H * n * n * 64 / 1e9
    tf = timeit(lambda: ops.attention_fwd(qkv, pm, o, lse, B, H, n, 64))
    tb = timeit(lambda: ops.attention_bwd(do, qkv, o, lse, pm, dqkv, dvec, B, H, n, 64))
    print(f"B={B} H={H} n={n} masked={masked}: fwd {tf:7.1f} us ({2 * unit / tf * 1e3:5.1f} TF) | bwd {tb:7.1f} us ({7 * unit / tb * 1e3:5.1f} TF executed, "
          f"{4 * unit / tb * 1e3:5.1f} useful)", flush=True)
