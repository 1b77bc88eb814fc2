"""OGM / OGM-GE gradient modulation (main.py:312-410) on the HIP kernels vs the reference's own vectors
(tests/golden/ogm_kat.npz: coefficients / scores / ratios, OGM-scaled conv gradients, OGM-GE noise scale) and vs the
reference's loop executed verbatim on mla_hip modules.  Scaling is compared exactly (given the coefficient), the
coefficient to 1e-6 (tanhf vs ATen's tanh), the Gaussian noise by its moments."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import mla_oracle as O  # noqa: E402
from util import assert_close  # noqa: E402


class Args:
    fusion_method, dataset, gs_flag, modulation = "concat", "CREMAD", True, "OGM_GE"


def test_ogm_coefficients_vs_reference_vectors(golden_dir):
    from mla_hip import OGM
    fx = np.load(os.path.join(golden_dir, "ogm_kat.npz"))
    label = torch.from_numpy(fx["label"]).cuda()
    for cname in ("ref", "swapped", "sharp"):
        oa, ov = torch.from_numpy(fx[f"{cname}.out_a"]).cuda(), torch.from_numpy(fx[f"{cname}.out_v"]).cuda()
        for alpha in (0.1, 0.3, 0.8):
            og = OGM(alpha=alpha, mode="OGM")
            cf = og.coefficients([oa, ov], label)
            want = fx[f"{cname}.alpha{alpha}"]                       # coeff_a, coeff_v, score_a, score_v, ratio_a, ratio_v
            assert_close(cf, want[:2], atol=1e-6, name=f"{cname} alpha={alpha} coefficients")
            assert_close(og.info[:2], want[2:4], atol=0, rtol=2e-6, name="scores")
            assert_close(og.info[3:5], want[4:6], atol=0, rtol=4e-6, name="ratios")
    oa, ov, ot = [torch.from_numpy(fx[k]).cuda() for k in ("ref.out_a", "ref.out_v", "three.out_t")]
    for tag, outs in (("avt", [oa, ov, ot]), ("tva", [ot * 3, ov, oa])):
        og = OGM(alpha=0.3, mode="OGM")
        cf = og.coefficients(outs, label)
        assert_close(cf, fx[f"three.{tag}"][:3], atol=1e-6, name=f"three modalities {tag}")
        assert_close(og.info[3:6], fx[f"three.{tag}"][3:], atol=0, rtol=4e-6, name="ratios")


def _model_with_fixture_grads(fx):
    """AVClassifier whose conv1 / layer1.0.conv1 / bn1 gradients are the reference's (fixture), the rest seeded noise."""
    import mla_hip
    model = mla_hip.AVClassifier(Args(), seed=0)
    g = torch.Generator(device="cuda").manual_seed(9)
    for enc, name in ((model.audio_net, "audio_net"), (model.visual_net, "visual_net")):
        enc.grad.copy_(torch.randn(enc.grad.shape, device="cuda", generator=g) * 0.01)
        for k in ("conv1.weight", "layer1.0.conv1.weight", "bn1.weight"):
            t = torch.from_numpy(fx[f"grad.{name}.{k}"]).cuda()
            enc.g[k].copy_(t.permute(2, 3, 1, 0) if t.dim() == 4 else t)            # reference OIHW -> flat HWIO
        enc.publish_grads()
    return model


@pytest.mark.parametrize("mode", ["OGM", "OGM_GE"])
def test_ogm_modulation_vs_reference_vectors_and_verbatim_loop(golden_dir, mode):
    import mla_hip
    fx = np.load(os.path.join(golden_dir, "ogm_kat.npz"))
    label = torch.from_numpy(fx["label"]).cuda()
    oa, ov = torch.from_numpy(fx["ref.out_a"]).cuda(), torch.from_numpy(fx["ref.out_v"]).cuda()
    model = _model_with_fixture_grads(fx)
    before = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    og = mla_hip.OGM(alpha=0.3, mode=mode, seed=5)
    cf = og.coefficients([oa, ov], label)
    og.modulate([model.audio_net, model.visual_net], epoch=60, modulation_ends=50)            # outside the window: no-op
    assert all(torch.equal(p.grad, before[n]) for n, p in model.named_parameters() if p.grad is not None)
    og.modulate([model.audio_net, model.visual_net], epoch=3)
    after = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    coeff = {"audio_net": cf[0], "visual_net": cf[1]}
    for n, g0 in before.items():
        enc = n.split(".")[0]
        if g0.dim() != 4:
            assert torch.equal(after[n], g0), f"{n}: only 4-D gradients are modulated (main.py:397)"
            continue
        scaled = g0 * coeff[enc]
        if mode == "OGM":
            assert torch.equal(after[n], scaled), n                                            # exact given the coefficient
        else:
            noise = (after[n] - scaled).double()
            sd = g0.double().std().item() + 1e-8
            tol = 4.0 / (g0.numel() ** 0.5)                                                     # 4 sigma of the estimators
            assert abs(noise.std().item() / sd - 1.0) < max(tol, 2e-3), (n, noise.std().item(), sd)
            assert abs(noise.mean().item()) < max(tol, 2e-3) * sd * 1.5, n
            if g0.numel() >= 30000:                                                             # Gaussian shape: kurtosis 3, skew 0
                z = noise / noise.std()
                assert abs((z ** 4).mean().item() - 3.0) < 0.15 and abs((z ** 3).mean().item()) < 0.06, n
    if mode == "OGM_GE":      # independent draws per tensor AND per modality (the reference calls normal_() per tensor, main.py:399-407)
        def z_of(n):
            return ((after[n] - before[n] * coeff[n.split(".")[0]]).double() / (before[n].double().std() + 1e-8)).flatten()
        for k in ("layer1.0.conv1.weight", "layer3.1.conv2.weight"):          # same shape, same position in both encoders
            za, zv = z_of("audio_net." + k), z_of("visual_net." + k)
            corr = (za * zv).mean().item() / (za.std().item() * zv.std().item())
            assert abs(corr) < 5.0 / za.numel() ** 0.5, (k, corr)
        za, zb = z_of("audio_net.layer1.0.conv1.weight"), z_of("audio_net.layer1.0.conv2.weight")
        assert abs((za * zb).mean().item()) < 5.0 / za.numel() ** 0.5
    for enc in ("audio_net", "visual_net"):                                                     # the reference's own numbers
        for k in ("conv1.weight", "layer1.0.conv1.weight", "bn1.weight"):
            if mode == "OGM":
                assert_close(after[f"{enc}.{k}"], fx[f"ogm.{enc}.{k}"], atol=0, rtol=2e-6, name=f"OGM-scaled {enc}.{k} vs reference")
            elif k != "bn1.weight":
                noise = (after[f"{enc}.{k}"] - before[f"{enc}.{k}"] * coeff[enc]).double()
                want = float(fx[f"ge_std.{enc}.{k}"])
                assert abs(noise.std().item() - want) < 0.03 * want, (enc, k, noise.std().item(), want)
    if mode == "OGM_GE":                                                  # reproducible, and fresh noise on the next call
        model2 = _model_with_fixture_grads(fx)
        og2 = mla_hip.OGM(alpha=0.3, mode=mode, seed=5)
        og2.coefficients([oa, ov], label)
        og2.modulate([model2.audio_net, model2.visual_net], epoch=3)
        assert torch.equal(model2.audio_net.grad, model.audio_net.grad)
        a1 = model2.audio_net.g["layer1.0.conv1.weight"].clone()
        model2.audio_net.grad.copy_(_model_with_fixture_grads(fx).audio_net.grad)
        og2.modulate([model2.audio_net, model2.visual_net], epoch=3)
        assert not torch.equal(model2.audio_net.g["layer1.0.conv1.weight"], a1)
    # the reference's loop, verbatim (main.py:392-408), on the same mla_hip modules: identical scaling
    model3 = torch.nn.DataParallel(_model_with_fixture_grads(fx), device_ids=[0])
    coeff_a, coeff_v = cf[0].clone(), cf[1].clone()

    class args:
        modulation, modulation_starts, modulation_ends = mode, 0, 50
    epoch = 3
    if args.modulation_starts <= epoch <= args.modulation_ends:
        for name, parms in model3.named_parameters():
            if parms.grad is None:
                continue
            layer = str(name).split('.')[1]

            if 'audio' in layer and len(parms.grad.size()) == 4:
                if args.modulation == 'OGM_GE':
                    parms.grad = parms.grad * coeff_a + \
                                torch.zeros_like(parms.grad).normal_(0, parms.grad.std().item() + 1e-8)
                elif args.modulation == 'OGM':
                    parms.grad *= coeff_a

            if 'visual' in layer and len(parms.grad.size()) == 4:
                if args.modulation == 'OGM_GE':
                    parms.grad = parms.grad * coeff_v + \
                                torch.zeros_like(parms.grad).normal_(0, parms.grad.std().item() + 1e-8)
                elif args.modulation == 'OGM':
                    parms.grad *= coeff_v
    if mode == "OGM":
        for n, p in model3.module.named_parameters():
            if p.grad is not None:
                assert torch.equal(p.grad, after[n]), f"verbatim loop vs fused kernel: {n}"
    else:   # re-assigned gradients are ordinary tensors now; FusedSGD must honour them
        opt = mla_hip.FusedSGD(model3.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
        p = dict(model3.module.named_parameters())["audio_net.layer1.0.conv1.weight"]
        w0, g_new = p.detach().clone(), p.grad.clone()
        opt.step()
        assert_close(w0 - p.detach(), 1e-3 * (g_new + 1e-4 * w0), atol=1e-8, rtol=1e-4, name="step uses the modulated gradient")
