import os, sys, time
sys.path.insert(0, "/root/repo/multimodal-learning-with-alternating-unimodal-adaptation_amd")
import torch
from mla_hip import AVClassifier, MLATrainer
class Args: fusion_method, dataset, gs_flag, modulation = "concat", "CREMAD", True, "Normal"
B = 64
g = torch.Generator(device="cuda").manual_seed(0)
spec = torch.randn((B, 1024, 128), device="cuda", generator=g) * 4.48 - 5.08
image = torch.randn((B, 3, 3, 224, 224), device="cuda", generator=g)
label = torch.randint(0, 6, (B,), device="cuda", generator=g)
res = {}
trs = {}
for ov in (False, True):
    m = AVClassifier(Args(), seed=1); tr = MLATrainer(m); tr.set_overlap(ov)
    trs[ov] = tr
for rnd in range(3):
    for ov in (False, True):
        tr = trs[ov]
        for s in range(2): tr.train_step(spec, image, label, s, 100)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(8): tr.train_step(spec, image, label, s, 100)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
        print(f"round {rnd} overlap={ov}: {dt*1e3:.2f} ms/step  {B/dt:.1f} samples/s  loss {tr.losses['loss'].item():.5f}")
