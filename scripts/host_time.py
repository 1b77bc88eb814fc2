"""Host-side enqueue time of one MLA step (CPU-bound floor), CREMA-D B=64."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
from mla_hip import AVClassifier, MLATrainer
class Args: fusion_method, dataset, gs_flag, modulation = "concat", "CREMAD", True, "Normal"
B = int(os.environ.get("B", "64"))
m = AVClassifier(Args(), seed=1); tr = MLATrainer(m)
spec = torch.randn(B, 1024, 128, device="cuda"); image = torch.randn(B, 3, 3, 224, 224, device="cuda"); label = torch.randint(0, 6, (B,), device="cuda")
for s in range(3): tr.train_step(spec, image, label, s, 100)
torch.cuda.synchronize()
ts = []
for s in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.train_step(spec, image, label, s, 100)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("host enqueue ms / total ms per step:", [(round(a * 1e3, 2), round(b * 1e3, 2)) for a, b in ts])
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for s in range(3): tr.train_step(spec, image, label, s, 100)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
