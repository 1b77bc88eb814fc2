"""Free-running loss trajectory: HIP (f32 / split) vs CPU oracle, deviation per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
from oracle import mla_oracle as O
from test_step_gpu import build, inputs
seed, B, steps = 77, int(os.environ.get("B", "8")), int(os.environ.get("STEPS", "12"))
lr = float(os.environ.get("LR", "1e-3"))
shape = ((96, 64), 2, (64, 64))
_, _, st = build(seed, "as_intended", False)
ref = []
for s in range(steps):
    spec, image, label = inputs(seed, s, B, *shape)
    r = O.mla_step(st, spec, image, label, s, steps, lr=lr)
    ref.append((float(r["loss"]), r["out_a"].clone(), r["out_v"].clone()))
print("oracle losses", [round(r[0], 4) for r in ref])
for cm in ("f32", "split"):
    model, tr, _ = build(seed, "as_intended", False, cm)
    tr.optimizer.set_lr(lr)
    dev = []
    for s in range(steps):
        spec, image, label = inputs(seed, s, B, *shape)
        losses = tr.train_step(spec.cuda(), image.cuda(), label.cuda(), s, steps)
        dl = abs(float(losses["loss"]) - ref[s][0])
        do = max(float((tr.last["out_a"].cpu() - ref[s][1]).abs().max()), float((tr.last["out_v"].cpu() - ref[s][2]).abs().max()))
        dev.append((round(dl, 6), round(do, 6)))
    print(cm, dev)
