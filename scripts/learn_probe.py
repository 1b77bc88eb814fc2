"""Loss trajectories of the HIP trainer and of the CPU oracle on the same learnable synthetic task (chaotic after a few steps, so
only the qualitative behaviour is comparable): does the as_intended projection + lr destabilise the head in BOTH?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
from oracle import mla_oracle as O
from mla_hip import AVClassifier, MLATrainer

lr = float(os.environ.get("LR", "1e-2")); mode = os.environ.get("GS", "as_intended"); steps = int(os.environ.get("STEPS", "80"))
seed = 3
pa, pv, hd = O.make_resnet18_params("audio", seed), O.make_resnet18_params("visual", seed + 1), O.make_head_params(512, 6, seed + 2)
model = AVClassifier(type("A", (), dict(fusion_method="concat", dataset="CREMAD", gs_flag=True, modulation="Normal"))(), seed=0)
sd = {f"audio_net.{k}": v for k, v in pa.items()}; sd.update({f"visual_net.{k}": v for k, v in pv.items()})
sd.update({f"fusion_module.fc_out.{k}": v for k, v in hd.items()})
model.load_state_dict(sd)
tr = MLATrainer(model, lr=lr, momentum=0.9, weight_decay=1e-4, gs_mode=mode)
st = O.MLAState(pa, pv, hd)
g = torch.Generator().manual_seed(0)
B = 16
torch.set_num_threads(8)
for s in range(steps):
    label = torch.randint(0, 6, (B,), generator=g)
    spec = torch.randn((B, 128, 64), generator=g) + (label.float() - 2.5)[:, None, None] * 0.8
    image = torch.randn((B, 3, 2, 64, 64), generator=g)
    image[:, 0] += (label.float() - 2.5)[:, None, None, None] * 0.6
    image[:, 1] -= (label.float() % 2)[:, None, None, None] * 0.8
    losses = tr.train_step(spec.cuda(), image.cuda(), label.cuda(), s % 10, 10)
    ref = O.mla_step(st, spec, image, label, s % 10, 10, gs_mode=mode, lr=lr) if "lr" in O.mla_step.__code__.co_varnames else O.mla_step(st, spec, image, label, s % 10, 10, gs_mode=mode)
    if s % 8 == 0 or s == steps - 1:
        print(f"step {s:3d}  HIP a {losses['loss_a'].item():9.4f} v {losses['loss_v'].item():8.4f} |feat a| {tr.last['a'].abs().max().item():8.2f} |W| {model.fusion_module.fc_out.weight.abs().max().item():7.3f}"
              f"   oracle a {float(ref['loss_a']):9.4f} v {float(ref['loss_v']):8.4f} |W| {st.head['weight'].abs().max().item():7.3f}", flush=True)
