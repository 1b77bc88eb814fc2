"""PCIe-inclusive rate of the CREMA-D MLA step (config 2): batches start in host memory every step.
(a) blocking .to(device) per tensor like main.py:159-162; (b) DeviceFeeder (pinned, copy stream, double-buffered)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
from mla_hip import AVClassifier, MLATrainer, DeviceFeeder
class Args: fusion_method, dataset, gs_flag, modulation = "concat", "CREMAD", True, "Normal"
B, steps = 64, 12
model = AVClassifier(Args(), seed=1); tr = MLATrainer(model)
host = [(torch.randn(B, 1024, 128) * 4.48 - 5.08, torch.randn(B, 3, 3, 224, 224), torch.randint(0, 6, (B,))) for _ in range(3)]
def gen(n):
    for i in range(n): yield host[i % 3]
for s, (spec, image, label) in enumerate(gen(3)):                      # warm-up
    tr.train_step(spec.cuda(), image.cuda(), label.cuda(), s, 100)
torch.cuda.synchronize(); t0 = time.perf_counter()
for s, (spec, image, label) in enumerate(gen(steps)):
    tr.train_step(spec.to("cuda"), image.to("cuda"), label.to("cuda"), s, 100)       # main.py:159-162 (pageable, blocking)
torch.cuda.synchronize(); dt_a = (time.perf_counter() - t0) / steps
feeder = DeviceFeeder()
for s, (spec, image, label) in enumerate(feeder.feed(gen(4))):          # allocates the pinned slots (once per run)
    tr.train_step(spec, image, label, s, 100)
torch.cuda.synchronize(); t0 = time.perf_counter()
for s, (spec, image, label) in enumerate(feeder.feed(gen(steps))):
    tr.train_step(spec, image, label, s, 100)
torch.cuda.synchronize(); dt_b = (time.perf_counter() - t0) / steps
dev = [tuple(t.cuda() for t in b) for b in host]
torch.cuda.synchronize(); t0 = time.perf_counter()
for s in range(steps):
    spec, image, label = dev[s % 3]; tr.train_step(spec, image, label, s, 100)
torch.cuda.synchronize(); dt_c = (time.perf_counter() - t0) / steps
print(f"inputs resident in HBM      : {dt_c*1e3:.2f} ms/step  {B/dt_c:.1f} samples/s")
print(f"blocking .to(device) (ref)  : {dt_a*1e3:.2f} ms/step  {B/dt_a:.1f} samples/s")
print(f"DeviceFeeder (pinned, async): {dt_b*1e3:.2f} ms/step  {B/dt_b:.1f} samples/s")
