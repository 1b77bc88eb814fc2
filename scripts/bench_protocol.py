"""The reference's step loop (main.py:431-476) executed through the nn.Module / autograd / optimizer protocol at
BASELINE configs[1] (batch 64), next to the fused MLATrainer: how much the protocol shell costs.  Numbers go to DESIGN.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd"))
import torch
import mla_hip

B, steps = int(os.environ.get("B", "64")), int(os.environ.get("STEPS", "10"))
math = os.environ.get("MATH", "f32")


class Args:
    fusion_method, dataset, gs_flag, modulation = "concat", "CREMAD", True, "Normal"


g = torch.Generator(device="cuda").manual_seed(0)
spec = torch.randn((B, 1024, 128), device="cuda", generator=g) * 4.4849 - 5.081
image = torch.randn((B, 3, 3, 224, 224), device="cuda", generator=g)
label = torch.randint(0, 6, (B,), device="cuda", generator=g)

model = mla_hip.DataParallel(mla_hip.AVClassifier(Args(), seed=1, conv_math=math), device_ids=[0])
optimizer = mla_hip.FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
gs_plugin = mla_hip.GSPlugin()
criterion = mla_hip.CrossEntropyLoss() if os.environ.get("CRIT", "mla") == "mla" else torch.nn.CrossEntropyLoss()
model.train()


def step(batch_step, len_dataloader=105, av_alpha=0.55):
    optimizer.zero_grad()
    a, v = model(spec.unsqueeze(1).float(), image.float())
    out_a = model.module.fusion_module.fc_out(a)
    loss_a = criterion(out_a, label)
    loss_a.backward()
    gs_plugin.before_update(model.module.fusion_module.fc_out, a, batch_step, len_dataloader, gs_plugin.exp_count)
    optimizer.step()
    optimizer.zero_grad()
    gs_plugin.exp_count += 1
    out_v = model.module.fusion_module.fc_out(v)
    loss_v = criterion(out_v, label)
    loss_v.backward()
    gs_plugin.before_update(model.module.fusion_module.fc_out, v, batch_step, len_dataloader, gs_plugin.exp_count)
    optimizer.step()
    optimizer.zero_grad()
    gs_plugin.exp_count += 1
    for n, p in model.named_parameters():
        if p.grad != None:
            del p.grad
    return (loss_a * av_alpha + loss_v * (1 - av_alpha)).item()       # the .item() of main.py:472: one host sync per step


for s in range(3):
    step(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(steps):
    loss = step(3 + s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
tr = mla_hip.MLATrainer(mla_hip.AVClassifier(Args(), seed=1, conv_math=math))
res = {}
for name, ov in (("serialized", False), ("pipelined", True)):
    tr.set_overlap(ov)
    for s in range(3):
        tr.train_step(spec, image, label, s, 105)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        tr.train_step(spec, image, label, 3 + s, 105)
    torch.cuda.synchronize()
    res[name] = (time.perf_counter() - t0) / steps
print(f"protocol path ({math}): {dt*1e3:.2f} ms/step = {B/dt:.0f} samples/s (loss {loss:.4f}); MLATrainer serialized "
      f"{res['serialized']*1e3:.2f} ms, pipelined {res['pipelined']*1e3:.2f} ms = {B/res['pipelined']:.0f} samples/s")
