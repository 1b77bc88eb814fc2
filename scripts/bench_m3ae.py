"""Config 4 of BASELINE.json (Food-101, --lorb m3ae, text+image M3AE ViT-B, batch 64) on one MI355X:
samples/s of the MLA step.  Not the headline bench line (bench.py measures config 2); numbers go to DESIGN.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
from mla_hip import M3AEClassifier, MLATrainer, ops

B = int(os.environ.get("B", "64")); steps = int(os.environ.get("STEPS", "5")); depth = int(os.environ.get("DEPTH", "12"))
class Args: fusion_method, dataset, gs_flag, modulation = "concat", "Food101", True, "Normal"
model = M3AEClassifier(Args(), depth=depth, seed=1, conv_math=os.environ.get("MATH", "f32"))
tr = MLATrainer(model)
if os.environ.get("F32_CFG"):
    ops.conv2d_f32_cfg(int(os.environ["F32_CFG"]))      # measurement hook: force one fp32 tile (0: 128x128, 1: 256x64, 2: 64x64, 3: 128x64)
if os.environ.get("OVERLAP") == "0":
    tr.set_overlap(False)
g = torch.Generator(device="cuda").manual_seed(0)
token = torch.randint(0, 30522, (B, 1, 256), device="cuda", generator=g)
lens = torch.randint(8, 257, (B,), device="cuda", generator=g)
pm = (torch.arange(256, device="cuda")[None, :] >= lens[:, None]).float().view(B, 1, 256)
image = torch.randn((B, 3, 256, 256), device="cuda", generator=g)
label = torch.randint(0, 101, (B,), device="cuda", generator=g)
for s in range(2):
    tr.train_step(token, pm, image, label, s, 100)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(steps):
    tr.train_step(token, pm, image, label, s + 2, 100)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
flops = 2 * 3 * 2 * B * 257 * depth * (12 * 768 * 768 + 2 * 257 * 768)     # 12 d^2 + 2 N d MAC per token per layer
print(f"M3AE MLA step ({os.environ.get('MATH', 'f32')}): B={B} depth={depth}: {dt*1e3:.1f} ms/step, {B/dt:.1f} samples/s, ~{flops/dt/1e12:.1f} TFLOP/s (algorithmic), loss {tr.losses['loss'].item():.4f}")
