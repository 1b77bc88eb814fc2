"""Config 5 of BASELINE.json (IEMOCAP, --modal3: CAV-MAE audio + M3AE image + M3AE text, 3-way alternation), per-GPU batch 32,
on one MI355X: samples/s of the MLA step.  Not the headline bench line; numbers go to DESIGN.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
from mla_hip import Modal3Classifier, MLATrainer

B = int(os.environ.get("B", "32")); steps = int(os.environ.get("STEPS", "5")); depth = int(os.environ.get("DEPTH", "12"))
math_ = os.environ.get("MATH", "f32")
class Args: fusion_method, dataset, gs_flag, modulation, modal3 = "concat", "IEMOCAP", True, "Normal", True
model = Modal3Classifier(Args(), depth=depth, seed=1, conv_math=math_)
tr = MLATrainer(model)
g = torch.Generator(device="cuda").manual_seed(0)
token = torch.randint(0, 30522, (B, 1, 256), device="cuda", generator=g)
lens = torch.randint(8, 257, (B,), device="cuda", generator=g)
pm = (torch.arange(256, device="cuda")[None, :] >= lens[:, None]).float().view(B, 1, 256)
image = torch.randn((B, 3, 256, 256), device="cuda", generator=g)
spec = torch.randn((B, 1024, 128), device="cuda", generator=g) * 4.4849 - 5.081
label = torch.randint(0, 4, (B,), device="cuda", generator=g)
for s in range(2):
    tr.train_step(token, pm, image, spec, label, s, 100)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(steps):
    tr.train_step(token, pm, image, spec, label, s + 2, 100)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"Modal3 MLA step ({math_}): B={B} depth={depth}: {dt*1e3:.1f} ms/step, {B/dt:.1f} samples/s, loss {tr.losses['loss'].item():.4f}")
