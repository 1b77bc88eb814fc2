"""The drop-in boundary: the reference's own step loop, restated VERBATIM from main.py:431-454, 468-476,
runs against mla_hip objects through the PyTorch object protocol -- `model(...)` with autograd history,
`model.module.fusion_module.fc_out(a)`, `criterion(out, label)`, `loss.backward()`, `gs_plugin.before_update(fc_out, a,
...)` reading `w.grad` via `named_parameters()`, `optimizer.step()` / `zero_grad()` over `model.parameters()`,
`del p.grad` -- and matches the golden outputs the imported reference produced (tests/golden/make_golden.py) at the
same tolerances as the fused `MLATrainer.train_step` path (tests/test_step_gpu.py).

Pinned semantics per fixture: projection mode (Q1) and zero_grad flavour (Q6: `legacy` = torch 1.8.1, tensors kept and
zero-filled == `optimizer.zero_grad(set_to_none=False)`).
"""
import io
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import mla_oracle as O  # noqa: E402
from util import assert_close, assert_close_robust  # noqa: E402

TOL = 2e-4


class Args:
    fusion_method, dataset, gs_flag, modulation = "concat", "CREMAD", True, "Normal"
    lorb, clip, modal3 = "base", False, False


def build_protocol(seed, gs_mode, conv_math="f32", wrapper="mla", lr=1e-3):
    import mla_hip
    model = mla_hip.AVClassifier(Args(), seed=0, conv_math=conv_math)
    pa, pv = O.make_resnet18_params("audio", seed), O.make_resnet18_params("visual", seed + 1)
    hd = O.make_head_params(512, 6, seed + 2)
    sd = {f"audio_net.{k}": v for k, v in pa.items()}
    sd.update({f"visual_net.{k}": v for k, v in pv.items()})
    sd.update({f"fusion_module.fc_out.{k}": v for k, v in hd.items()})
    missing, unexpected = model.load_state_dict(sd, strict=False)                      # main.py:727
    assert not missing and not unexpected
    model.to(torch.device("cuda:0"))                                                   # main.py:730
    model = (mla_hip.DataParallel if wrapper == "mla" else torch.nn.DataParallel)(model, device_ids=[0])   # :732
    model.cuda()                                                                       # :734
    optimizer = mla_hip.FusedSGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)   # :749
    gs_plugin = mla_hip.GSPlugin(mode=gs_mode)                                         # :819
    return model, optimizer, gs_plugin


def reference_loop_body(args, model, optimizer, gs_plugin, criterion, spec, image, label, batch_step, len_dataloader,
                        av_alpha, rec, legacy):
    """main.py:431-454 and 468-476, verbatim (the `rec[...]` lines and the `legacy` switch are the only additions)."""
    zero_grad = (lambda: optimizer.zero_grad(set_to_none=False)) if legacy else optimizer.zero_grad
    a, v = model(spec.unsqueeze(1).float(), image.float())
    out_a = model.module.fusion_module.fc_out(a)

    loss_a = criterion(out_a, label)
    loss_a.backward()
    rec["head_grad_a_raw"] = model.module.fusion_module.fc_out.weight.grad.clone()

    gs_plugin.before_update(model.module.fusion_module.fc_out, a,
                            batch_step, len_dataloader, gs_plugin.exp_count)
    rec["head_grad_a"] = model.module.fusion_module.fc_out.weight.grad.clone()
    optimizer.step()
    zero_grad()

    gs_plugin.exp_count += 1

    out_v = model.module.fusion_module.fc_out(v)

    loss_v = criterion(out_v, label)
    loss_v.backward()
    rec["head_grad_v_raw"] = model.module.fusion_module.fc_out.weight.grad.clone()

    gs_plugin.before_update(model.module.fusion_module.fc_out, v,
                            batch_step, len_dataloader, gs_plugin.exp_count)
    rec["head_grad_v"] = model.module.fusion_module.fc_out.weight.grad.clone()
    rec["grads"] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    optimizer.step()
    zero_grad()

    gs_plugin.exp_count += 1

    for n, p in model.named_parameters():
        if p.grad != None:
            del p.grad

    _loss = (loss_a * av_alpha + loss_v * (1 - av_alpha)).item()
    rec.update(a=a.detach(), v=v.detach(), out_a=out_a.detach(), out_v=out_v.detach(), loss=_loss,
               loss_a=loss_a.item(), loss_v=loss_v.item())


def inputs(seed, s, B, spec_hw, T, img_hw):
    spec = O.portable_normal(seed + 100 + s, (B,) + tuple(spec_hw), stream=1, mean=-5.081, std=4.4849)
    image = O.portable_normal(seed + 100 + s, (B, 3, T) + tuple(img_hw), stream=2)
    label = O.portable_labels(seed + 100 + s, B, 6)
    return spec.cuda(), image.cuda(), label.cuda()


@pytest.mark.parametrize("conv_math", ["split", "f32"])
@pytest.mark.parametrize("criterion_kind,wrapper", [("mla", "mla"), ("torch", "torch")])
@pytest.mark.parametrize("tag", ["small_intended", "small_published", "small_legacy"])
def test_reference_loop_verbatim_vs_golden(tag, criterion_kind, wrapper, conv_math, golden_dir):
    import mla_hip
    fx = np.load(os.path.join(golden_dir, f"mla_{tag}.npz"))
    B, sh, sw, T, ih, iw, steps, seed, ldl = [int(v) for v in fx["meta"]]
    gs_mode, legacy = str(fx["gs_mode"]), bool(int(fx["legacy"]))
    model, optimizer, gs_plugin = build_protocol(seed, gs_mode, conv_math=conv_math, wrapper=wrapper)
    criterion = mla_hip.CrossEntropyLoss() if criterion_kind == "mla" else nn.CrossEntropyLoss()   # main.py:130
    model.train()
    for s in range(steps):
        spec, image, label = inputs(seed, s, B, (sh, sw), T, (ih, iw))
        rec = {}
        optimizer.zero_grad()                                                                       # main.py:164
        reference_loop_body(Args(), model, optimizer, gs_plugin, criterion, spec, image, label, s, ldl, 0.55, rec, legacy)
        tol = TOL if s == 0 else 1e-3            # step 1 is free-running (see tests/test_step_gpu.py)
        for k in ("a", "v", "out_a", "out_v", "head_grad_a_raw", "head_grad_v_raw", "head_grad_a", "head_grad_v"):
            assert_close(rec[k], fx[f"s{s}.{k}"], atol=tol, name=f"{tag} s{s} {k}")
        for k in ("loss", "loss_a", "loss_v"):
            assert abs(rec[k] - float(fx[f"s{s}.{k}"])) <= tol, (k, rec[k], float(fx[f"s{s}.{k}"]))
        # the visual phase leaves gradients on the head and the visual encoder only (torch >= 2 zero_grad); legacy keeps zeros
        names = set(rec["grads"])
        assert any(n.startswith("module.visual_net.") for n in names) and "module.fusion_module.fc_out.weight" in names
        assert any(n.startswith("module.audio_net.") for n in names) == legacy
        for n, p in model.named_parameters():
            assert p.grad is None                                                                   # main.py:468-470 ran
        sd = model.state_dict()                                                                     # DataParallel keys (main.py:921)
        assert_close(sd["module.fusion_module.fc_out.weight"], fx[f"s{s}.head.weight"], atol=tol, name="head weight")
        assert_close(sd["module.fusion_module.fc_out.bias"], fx[f"s{s}.head.bias"], atol=tol, name="head bias")
        for enc in ("audio_net", "visual_net"):
            assert_close(sd[f"module.{enc}.bn1.running_mean"], fx[f"s{s}.{enc}.bn1.running_mean"], atol=1e-5, rtol=1e-5, name="running_mean")
            assert_close(sd[f"module.{enc}.bn1.running_var"], fx[f"s{s}.{enc}.bn1.running_var"], atol=1e-5, rtol=1e-5, name="running_var")
            assert int(sd[f"module.{enc}.bn1.num_batches_tracked"]) == s + 1
            assert_close_robust(sd[f"module.{enc}.conv1.weight"], fx[f"s{s}.{enc}.conv1.weight"], rel_l2=2e-3, elem_tol=2e-3, frac=0.9,
                                name=f"{enc} conv1.weight")
            w = sd[f"module.{enc}.layer4.1.conv2.weight"]
            assert_close(w.flatten()[:64], fx[f"s{s}.{enc}.layer4.1.conv2.weight.head"], atol=2e-6, name="layer4 weight slice")
        for key in fx.files:                                  # encoder gradient digests of the reference's named_parameters()
            if key.startswith(f"s{s}.grad.visual.") and key.endswith(".abssum"):
                name = "module.visual_net." + key[len(f"s{s}.grad.visual."):-len(".abssum")]
                got = rec["grads"][name].double().abs().sum().item()
                assert abs(got - float(fx[key])) <= 5e-3 * float(fx[key]) + 1e-9, (key, got, float(fx[key]))
        Pl = gs_plugin.Pl.cpu()
        assert_close(Pl[:8, :8], fx[f"s{s}.Pl.corner"], atol=1e-6, rtol=1e-4, name="Pl corner")
        assert abs(torch.trace(Pl).item() - float(fx[f"s{s}.Pl.trace"])) < 1e-4


@pytest.mark.parametrize("conv_math", ["f32", "split"])
def test_protocol_path_equals_fused_trainer(conv_math):
    """Same kernels behind both doors: three steps through the verbatim loop equal three `MLATrainer.train_step`s --
    encoder parameters, momentum and BN buffers bit for bit (identical launch plans), head / Pl / losses to 1e-6 (the
    protocol path computes CE and the Linear backward in separate kernels, as autograd splits them)."""
    import mla_hip
    seed, B = 53, 4
    model, optimizer, gs_plugin = build_protocol(seed, "as_intended", conv_math)
    criterion = mla_hip.CrossEntropyLoss()
    ref_model = build_protocol(seed, "as_intended", conv_math)[0].module
    tr = mla_hip.MLATrainer(ref_model, lr=1e-3, momentum=0.9, weight_decay=1e-4, gs_mode="as_intended")
    model.train()
    for step in range(3):
        spec, image, label = inputs(seed, step, B, (128, 64), 2, (64, 64))
        rec = {}
        optimizer.zero_grad()
        reference_loop_body(Args(), model, optimizer, gs_plugin, criterion, spec, image, label, step, 10, 0.55, rec, False)
        losses = tr.train_step(spec, image, label, step, 10)
        torch.cuda.synchronize()
        assert abs(rec["loss"] - losses["loss"].item()) < 1e-6
    tr.join()
    m = model.module
    # step 0 is bit-identical; afterwards the head differs in the last bits (separate CE / Linear-backward kernels), which
    # enters the encoders through d feature
    assert_close(m.fusion_module.fc_out.flat, ref_model.fusion_module.fc_out.flat, atol=1e-6, name="head")
    assert_close(gs_plugin.Pl, tr.gs_plugin.Pl, atol=1e-6, name="Pl")
    assert gs_plugin.exp_count == tr.gs_plugin.exp_count == 6
    for a, b in ((m.audio_net, ref_model.audio_net), (m.visual_net, ref_model.visual_net)):
        assert_close(a.flat, b.flat, atol=1e-6, name="encoder parameters")
        assert_close(a.running, b.running, atol=1e-6, name="BN running statistics")


def test_weight_init_scheduler_and_eval_through_the_protocol():
    """main.py:719 `model.apply(weight_init)` with the REFERENCE-shaped function (isinstance dispatch on nn.Linear /
    nn.Conv2d / nn.BatchNorm2d) re-initialises the flat buffers through the parameter views; StepLR (main.py:760) drives
    FusedSGD's learning rate; `model.eval()` + `torch.no_grad()` (main.py:519-520) gives history-free outputs."""
    import mla_hip
    model, optimizer, _ = build_protocol(3, "as_intended")
    before = model.module.audio_net.flat.clone()

    def weight_init(m):                                   # what utils/utils.py:106-114 does, via torch's own initialisers
        if isinstance(m, nn.Linear):
            nn.init.xavier_normal_(m.weight)
            nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)

    mla_hip.setup_seed(0)
    model.apply(weight_init)
    enc = model.module.audio_net
    assert not torch.equal(before, enc.flat)
    w = enc.p["layer3.0.conv1.weight"]                                   # HWIO view of the same storage
    std = (2.0 / (256 * 9)) ** 0.5                                       # kaiming-normal, fan_out = Cout*k*k
    assert abs(w.std().item() - std) < 0.02 * std and abs(w.mean().item()) < 0.05 * std
    assert torch.all(enc.p["layer3.0.bn1.weight"] == 1) and torch.all(enc.p["layer3.0.bn1.bias"] == 0)
    head = model.module.fusion_module.fc_out
    assert torch.all(head.bias == 0) and abs(head.weight.std().item() - (2.0 / 518) ** 0.5) < 0.1 * (2.0 / 518) ** 0.5
    sched = torch.optim.lr_scheduler.StepLR(optimizer, 2, 0.1)            # main.py:760
    for _ in range(2):
        optimizer.step()
        sched.step()
    assert abs(optimizer.lr - 1e-4) < 1e-12
    spec, image, label = inputs(3, 0, 2, (64, 32), 2, (32, 32))
    model.eval()
    with torch.no_grad():
        a, v = model(spec.unsqueeze(1).float(), image.float())
        out = model.module.fusion_module.fc_out(a)
    assert not a.requires_grad and not out.requires_grad and out.shape == (2, 6)
    with pytest.raises(mla_hip.MLAHipError):
        model.module.double()                                              # pinned to fp32 on its device


def test_checkpoint_dict_roundtrip_and_bitwise_resume(tmp_path):
    """SURVEY 8f-3 / main.py:916-927 + 721-728: save the reference-shaped dict {'saved_epoch', 'modulation', 'alpha',
    'fusion', 'acc', 'model', 'optimizer', 'scheduler'} with torch.save, load it with torch.load(weights_only=True), and
    (a) warm-start a fresh model through the reference's recipe (strip `module.`, delete the head, strict=False);
    (b) fully resume -- model + optimizer + scheduler + the plugin state the reference forgets (Pl, exp_count) -- and get a
    bit-identical next step."""
    import mla_hip
    seed, B = 77, 3
    criterion = mla_hip.CrossEntropyLoss()

    def run(model, optimizer, gs_plugin, steps, start=0):
        out = None
        for s in range(start, start + steps):
            spec, image, label = inputs(seed, s, B, (96, 64), 2, (48, 48))
            rec = {}
            optimizer.zero_grad()
            reference_loop_body(Args(), model, optimizer, gs_plugin, criterion, spec, image, label, s, 10, 0.55, rec, False)
            out = rec
        return out

    model, optimizer, gs_plugin = build_protocol(seed, "as_intended")
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, 70, 0.1)
    model.train()
    run(model, optimizer, gs_plugin, 2)
    saved_dict = {'saved_epoch': 0, 'modulation': "Normal", 'alpha': 0.1, 'fusion': "concat", 'acc': 0.5,
                  'model': model.state_dict(), 'optimizer': optimizer.state_dict(), 'scheduler': scheduler.state_dict(),
                  'gs_plugin': {'Pl': gs_plugin.Pl, 'exp_count': gs_plugin.exp_count}}      # last key: our addition
    path = os.path.join(tmp_path, "ckpt.pth")
    torch.save(saved_dict, path)
    want = run(model, optimizer, gs_plugin, 1, start=2)
    want_state = {k: v.clone() for k, v in model.state_dict().items()}

    loaded_dict = torch.load(path, weights_only=True)
    assert set(loaded_dict) >= {'saved_epoch', 'modulation', 'alpha', 'fusion', 'acc', 'model', 'optimizer', 'scheduler'}
    assert all(k.startswith("module.") for k in loaded_dict['model'])
    osd = loaded_dict['optimizer']                                            # torch.optim.SGD's layout
    assert set(osd['param_groups'][0]) >= {'lr', 'momentum', 'weight_decay', 'params'}
    assert len(osd['param_groups'][0]['params']) == 122 and len(osd['state']) == 122
    assert osd['state'][2]['momentum_buffer'].shape == (64, 1, 7, 7)          # audio_net.conv1.weight, OIHW

    # (a) main.py:721-728
    fresh, _, _ = build_protocol(seed + 1, "as_intended")
    fresh = fresh.module
    state_dict = loaded_dict['model']
    state_dict = {key[7:]: state_dict[key] for key in state_dict}
    del state_dict["fusion_module.fc_out.weight"]
    del state_dict["fusion_module.fc_out.bias"]
    missing, unexcepted = fresh.load_state_dict(state_dict, strict=False)
    assert sorted(missing) == ["fusion_module.fc_out.bias", "fusion_module.fc_out.weight"] and not unexcepted
    assert torch.equal(fresh.audio_net.state_dict()["layer2.0.downsample.0.weight"].cpu(),
                       loaded_dict['model']["module.audio_net.layer2.0.downsample.0.weight"].cpu())

    # (b) full resume, bit-identical next step
    model2, optimizer2, gs2 = build_protocol(seed + 5, "as_intended")
    scheduler2 = torch.optim.lr_scheduler.StepLR(optimizer2, 70, 0.1)
    model2.load_state_dict(loaded_dict['model'])
    optimizer2.load_state_dict(loaded_dict['optimizer'])
    scheduler2.load_state_dict(loaded_dict['scheduler'])
    gs2.Pl.copy_(loaded_dict['gs_plugin']['Pl'])
    gs2.exp_count = loaded_dict['gs_plugin']['exp_count']
    model2.train()
    got = run(model2, optimizer2, gs2, 1, start=2)
    assert got["loss"] == want["loss"] and torch.equal(got["out_v"], want["out_v"])
    for k, v in model2.state_dict().items():
        assert torch.equal(v, want_state[k]), k
    assert torch.equal(gs2.Pl, gs_plugin.Pl)


def test_gradient_accumulation_and_foreign_gradients():
    """autograd's rule on top of kernels that overwrite: a second backward without zero_grad ADDS to .grad; a gradient
    tensor assigned by the user (OGM-style `parms.grad = parms.grad * coeff + noise`, main.py:398-400) is honoured by
    FusedSGD.step()."""
    import mla_hip
    seed = 19
    model, optimizer, _ = build_protocol(seed, "as_published")
    criterion = mla_hip.CrossEntropyLoss()
    spec, image, label = inputs(seed, 0, 2, (64, 32), 2, (32, 32))
    model.train()
    enc = model.module.audio_net

    def one_backward():
        a, _v = model(spec.unsqueeze(1).float(), image.float())
        criterion(model.module.fusion_module.fc_out(a), label).backward()

    one_backward()
    g1 = {n: p.grad.clone() for n, p in enc.named_parameters()}
    nbt = dict(enc.num_batches_tracked)
    # freeze BN running stats influence: gradients do not depend on running stats, so a second identical forward/backward doubles
    one_backward()
    for n, p in enc.named_parameters():
        assert_close(p.grad, 2 * g1[n], atol=1e-6, rtol=1e-5, name=f"accumulated {n}")
    assert all(enc.num_batches_tracked[k] == nbt[k] + 1 for k in nbt)
    # foreign gradient on one parameter: the optimizer must use it
    p = dict(enc.named_parameters())["layer1.0.conv1.weight"]
    before = p.detach().clone()
    p.grad = p.grad * 0.5 + 1.0
    want_g = p.grad.clone()
    optimizer.step()
    got = before - p.detach()                                # first step: buf = g + wd*p; p -= lr*buf
    assert_close(got, 1e-3 * (want_g + 1e-4 * before), atol=1e-7, rtol=1e-4, name="foreign gradient honoured")
    # partially-None owner: torch.optim.SGD skips exactly the parameters whose .grad is None (per-segment launches here),
    # and their momentum buffers stay where they were
    one_backward()
    params = dict(enc.named_parameters())
    skip, keep = params["layer2.0.conv1.weight"], params["layer2.0.conv2.weight"]
    skip.grad = None
    w_skip, w_keep, g_keep = skip.detach().clone(), keep.detach().clone(), keep.grad.clone()
    mom_keep = optimizer._mom_view([k for k, o in optimizer.groups.items() if o is enc][0], keep._mla_index).clone()
    optimizer.step()
    assert torch.equal(skip.detach(), w_skip), "a parameter without gradient must not move"
    buf = 0.9 * mom_keep + (g_keep + 1e-4 * w_keep)                    # second step of this parameter: momentum applies
    assert_close(w_keep - keep.detach(), 1e-3 * buf, atol=1e-8, rtol=1e-4, name="per-segment step")
    sd = optimizer.state_dict()
    assert len(sd["state"]) == 122 - 60                                     # head + audio encoder stepped; the visual encoder never did


def test_torch_library_ops(golden_dir):
    """the launchers are registered as `torch.ops.mla_hip.*` (dispatch key CUDA only).  Spot parity of the
    op-level surface against the oracle / the reference's own GSPlugin trajectory."""
    import math
    import mla_hip
    from mla_hip import torch_ops
    T = torch.ops.mla_hip
    assert {"conv2d_fwd", "conv2d_dgrad", "conv2d_wgrad", "bn_act_fwd", "bn_act_bwd", "maxpool3x3s2_fwd", "avgpool_fwd",
            "head_ce_fwd_bwd", "gs_project", "sgd_step", "layernorm_fwd", "linear_fwd", "attention_fwd"} <= set(torch_ops.op_names())
    # conv + BN + ReLU forward and the three gradients
    N, H, W, Cin, Cout, k, s, p = 2, 20, 12, 64, 128, 3, 2, 1
    x = O.portable_normal(1, (N, Cin, H, W), stream=1)
    w = O.portable_normal(1, (Cout, Cin, k, k), stream=2, std=math.sqrt(2.0 / (Cin * k * k)))
    xd, wd = x.permute(0, 2, 3, 1).contiguous().cuda(), w.permute(2, 3, 1, 0).contiguous().cuda()
    y_ref = O.conv2d_fwd(x, w, s, p)
    for math_ in ("f32", "split"):
        y = T.conv2d_fwd(xd, wd, s, p, math_)
        assert_close(y.permute(0, 3, 1, 2), y_ref, atol=0, rtol=2e-5, name=f"conv2d_fwd[{math_}]")
    y, mean, invstd = T.conv2d_fwd_stats(xd, wd, s, p)
    assert_close(mean, y_ref.mean(dim=(0, 2, 3)), atol=1e-5, name="fused BN mean")
    assert_close(invstd, 1.0 / torch.sqrt(y_ref.var(dim=(0, 2, 3), unbiased=False) + 1e-5), atol=0, rtol=1e-4, name="fused BN invstd")
    gamma, beta = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    out = T.bn_act_fwd(y, mean, invstd, gamma, beta, True)
    ref = torch.relu(torch.nn.functional.batch_norm(y_ref, None, None, gamma.cpu(), beta.cpu(), True, 0.1, 1e-5))
    assert_close(out.permute(0, 3, 1, 2), ref, atol=2e-5, name="bn_act_fwd")
    dy = O.portable_normal(1, tuple(y_ref.shape), stream=3)
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    assert_close(T.conv2d_dgrad(dyd, wd, list(xd.shape), s, p).permute(0, 3, 1, 2), O.conv2d_dgrad(dy, w, x.shape, s, p), atol=0,
                 rtol=2e-5, name="conv2d_dgrad")
    assert_close(T.conv2d_wgrad(xd, dyd, list(wd.shape), s, p).permute(3, 2, 0, 1), O.conv2d_wgrad(x, dy, w.shape, s, p), atol=0,
                 rtol=2e-5, name="conv2d_wgrad")
    # head + CE
    X, Wh, bh = torch.randn(8, 512), torch.randn(6, 512) * 0.05, torch.randn(6) * 0.1
    lab = torch.randint(0, 6, (8,))
    logits, loss, dW, db, dX = T.head_ce_fwd_bwd(X.cuda(), Wh.cuda(), bh.cuda(), lab.cuda(), 1.0 / 8)
    rl, rloss, rdW, rdb, rdX = O.head_ce_fwd_bwd(X, Wh, bh, lab)
    for got, want, nm in ((logits, rl, "logits"), (loss.reshape(()), rloss, "loss"), (dW, rdW, "dW"), (db, rdb, "db"), (dX, rdX, "dX")):
        assert_close(got, want, atol=1e-5, name=nm)
    # GS projection against the reference's own trajectory (tests/golden/gs_kat_d512.npz: utils/utils.py:34-41 outputs)
    fx = np.load(os.path.join(golden_dir, "gs_kat_d512.npz"))
    D_, C_, B_, calls, kseed = [int(v) for v in fx["meta"]]
    Pl = torch.eye(D_, device="cuda")
    for i in range(calls):
        Xb = O.portable_normal(kseed + i, (B_, D_), stream=5, mean=0.3, std=0.7).abs().cuda()
        G = O.portable_normal(kseed + i, (C_, D_), stream=6, std=0.05).cuda()
        if i > 0:                                                   # train_exp_counter != 0 (utils/utils.py:29)
            T.gs_project(Pl, Xb, G, mla_hip.GSPlugin.alpha(i % 7, 7))
        assert_close(G, fx[f"c{i}.G"], atol=1e-8, rtol=2e-5, name=f"gs_project call {i}")
    # SGD
    pp, g, buf = torch.randn(1000, device="cuda"), torch.randn(1000, device="cuda"), torch.zeros(1000, device="cuda")
    p0 = pp.clone()
    T.sgd_step(pp, g, buf, 1e-2, 0.9, 1e-4, True)
    assert_close(pp, p0 - 1e-2 * (g + 1e-4 * p0), atol=1e-7, rtol=1e-6, name="sgd first step")
    with pytest.raises((NotImplementedError, RuntimeError)):
        T.sgd_step(pp.cpu(), g.cpu(), buf.cpu(), 1e-2, 0.9, 1e-4, True)        # no CPU kernel: loud


@pytest.mark.parametrize("which", ["m3ae", "modal3"])
def test_reference_loop_verbatim_transformer_classifiers(which):
    """main.py:419-476 with --lorb m3ae (and --modal3): `a, v = model(token, padding_mask, image)` resp.
    `a, v, t = model(token, padding_mask, image, spec)`, then the per-modality blocks incl. the third one (main.py:455-466),
    executed verbatim on M3AEClassifier / Modal3Classifier protocol objects -- against the fused MLATrainer on an identical
    model (same launch plans): features / losses / raw head gradients to 1e-5, the updated head and every encoder to 1e-6."""
    import mla_hip
    from test_dist_gpu import T_B, _build_transformer, _transformer_case

    class args:
        lorb, modal3, clip = "m3ae", which == "modal3", False
    A, sd, inputs, label = _transformer_case(which)
    model_t, tr, _i, _l = _build_transformer(which)
    tr.set_overlap(False)
    cls = mla_hip.M3AEClassifier if which == "m3ae" else mla_hip.Modal3Classifier
    from test_dist_gpu import T_DEPTH, T_VOCAB
    model = cls(A(), depth=T_DEPTH, text_vocab_size=T_VOCAB, seed=0)
    model.load_state_dict(sd)
    model = mla_hip.DataParallel(model, device_ids=[0])
    optimizer = mla_hip.FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    gs_plugin = mla_hip.GSPlugin()                       # eye(512) by default; takes D = 768 from the head it is handed (Q3)
    criterion = mla_hip.CrossEntropyLoss()
    dev = [x.cuda() for x in inputs]
    label = label.cuda()
    if which == "m3ae":
        token, padding_mask, image = dev
    else:
        token, padding_mask, image, spec = dev
    model.train()
    for batch_step in range(2):
        len_dataloader = 10
        optimizer.zero_grad()
        # ---- main.py:419-466, verbatim
        if args.lorb == "large":
            a, v = model(spec, image)
        elif args.lorb == "m3ae":
            if args.modal3:
                a, v, t = model(token, padding_mask, image, spec)
            else:
                a, v = model(token, padding_mask, image)
        out_a = model.module.fusion_module.fc_out(a)

        loss_a = criterion(out_a, label)
        loss_a.backward()

        gs_plugin.before_update(model.module.fusion_module.fc_out, a,
                                batch_step, len_dataloader, gs_plugin.exp_count)
        optimizer.step()
        optimizer.zero_grad()

        gs_plugin.exp_count += 1

        out_v = model.module.fusion_module.fc_out(v)

        loss_v = criterion(out_v, label)
        loss_v.backward()

        gs_plugin.before_update(model.module.fusion_module.fc_out, v,
                                batch_step, len_dataloader, gs_plugin.exp_count)
        optimizer.step()
        optimizer.zero_grad()

        gs_plugin.exp_count += 1
        if args.modal3:
            out_t = model.module.fusion_module.fc_out(t)

            loss_t = criterion(out_t, label)
            loss_t.backward()

            gs_plugin.before_update(model.module.fusion_module.fc_out, t,
                                    batch_step, len_dataloader, gs_plugin.exp_count)
            optimizer.step()
            optimizer.zero_grad()

            gs_plugin.exp_count += 1

        for n, p in model.named_parameters():
            if p.grad != None:
                del p.grad
        # ---- the fused trainer on the twin model
        losses = tr.train_step(*dev, label, batch_step, len_dataloader)
        torch.cuda.synchronize()
        assert_close(a, tr.last["a"], atol=1e-5, name=f"step {batch_step} feature a")
        assert_close(v, tr.last["v"], atol=1e-5, name=f"step {batch_step} feature v")
        assert abs(loss_a.item() - losses["loss_a"].item()) < 1e-5 and abs(loss_v.item() - losses["loss_v"].item()) < 1e-5
        if args.modal3:
            assert_close(t, tr.last["t"], atol=1e-5, name="feature t")
            assert abs(loss_t.item() - losses["loss_t"].item()) < 1e-5
    assert gs_plugin.exp_count == tr.gs_plugin.exp_count == (6 if args.modal3 else 4) and gs_plugin.Pl.shape == (768, 768)
    # the projection of this path is ill-conditioned on transformer features: compare what feeds it and the
    # encoders tightly, the projected quantities loosely
    for (tag, _g, enc_t), enc_p in zip(model_t.mla_encoders(), [e for _t, _g2, e in model.module.mla_encoders()]):
        assert_close(enc_p.flat, enc_t.flat, atol=2e-6, name=f"encoder {tag} parameters after 2 steps")
    assert_close(model.module.fusion_module.fc_out.flat, model_t.fusion_module.fc_out.flat, atol=5e-5, name="head after 2 steps")
    names = [n for n, _p in model.module.named_parameters()]
    assert names[0] == "fusion_module.fc_out.weight" and any(n.startswith("mae_v.encoder.blocks.1.attention.qkv_linear") for n in names)
