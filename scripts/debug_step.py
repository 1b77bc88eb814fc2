"""Debug: run one full-size step with a device sync + log line after every C-ABI call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
from mla_hip import _lib, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
log = open(os.path.join(ROOT, "gpurun_out", "debug_step.log"), "w")
orig = _lib.check
t_last = [time.perf_counter()]
def check(rc, who):
    torch.cuda.synchronize()
    t = time.perf_counter()
    log.write(f"{who} rc={rc} dt={1e3*(t-t_last[0]):.2f} ms\n"); log.flush()
    t_last[0] = t
    orig(rc, who)
_lib.check = check; ops.check = check
from mla_hip import AVClassifier, MLATrainer
class Args: fusion_method, dataset, gs_flag, modulation = "concat", "CREMAD", True, "Normal"
model = AVClassifier(Args(), seed=1)
tr = MLATrainer(model)
spec = torch.randn((B, 1024, 128), device="cuda") * 4.48 - 5.08
image = torch.randn((B, 3, 3, 224, 224), device="cuda")
label = torch.randint(0, 6, (B,), device="cuda")
log.write("inputs ready\n"); log.flush()
for s in range(2):
    t0 = time.perf_counter()
    tr.train_step(spec, image, label, s, 100)
    torch.cuda.synchronize()
    log.write(f"=== step {s}: {1e3*(time.perf_counter()-t0):.1f} ms, loss {tr.losses['loss'].item():.4f}\n"); log.flush()
