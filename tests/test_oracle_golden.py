"""CPU: the oracle (oracle/mla_oracle.py) against the golden vectors that tests/golden/make_golden.py
recorded from the REFERENCE's own modules (autograd + torch.optim.SGD + GSPlugin.before_update).
Runs without /root/reference.  Tolerances: 2e-4 abs on features/logits/losses/head grads; encoder-gradient
digests norm-wise 5e-3 (a flipped ReLU decision between two CPU runs with different thread counts moves them)."""
import os

import numpy as np
import pytest
import torch

from oracle import mla_oracle as O
from util import assert_close


def _run(tag, golden_dir):
    fx = np.load(os.path.join(golden_dir, f"mla_{tag}.npz"))
    B, sh, sw, T, ih, iw, steps, seed, ldl = [int(v) for v in fx["meta"]]
    gs_mode, legacy = str(fx["gs_mode"]), bool(int(fx["legacy"]))
    st = O.MLAState(O.make_resnet18_params("audio", seed), O.make_resnet18_params("visual", seed + 1),
                    O.make_head_params(512, 6, seed + 2))
    for s in range(steps):
        spec = O.portable_normal(seed + 100 + s, (B, sh, sw), stream=1, mean=-5.081, std=4.4849)
        image = O.portable_normal(seed + 100 + s, (B, 3, T, ih, iw), stream=2)
        label = O.portable_labels(seed + 100 + s, B, 6)
        out = O.mla_step(st, spec, image, label, s, ldl, gs_mode=gs_mode, legacy_zero_grad=legacy)
        for k in ("a", "v", "out_a", "out_v", "loss_a", "loss_v", "loss", "head_grad_a_raw", "head_grad_v_raw",
                  "head_grad_a", "head_grad_v"):
            assert_close(out[k], fx[f"s{s}.{k}"], atol=2e-4, name=f"{tag} s{s} {k}")
        assert_close(st.head["weight"], fx[f"s{s}.head.weight"], atol=2e-4, name="head weight")
        assert_close(st.head["bias"], fx[f"s{s}.head.bias"], atol=2e-4, name="head bias")
        for enc, params in (("audio_net", st.audio), ("visual_net", st.visual)):
            assert_close(params["bn1.running_mean"], fx[f"s{s}.{enc}.bn1.running_mean"], atol=1e-5, rtol=1e-5, name="rm")
            assert_close(params["bn1.running_var"], fx[f"s{s}.{enc}.bn1.running_var"], atol=1e-5, rtol=1e-5, name="rv")
        for key in fx.files:
            if key.startswith(f"s{s}.grad.") and key.endswith(".abssum"):
                _, _, enc, *rest = key.split(".")
                g = out["grads_" + enc][".".join(rest[:-1])]
                want = float(fx[key])
                assert abs(g.double().abs().sum().item() - want) <= 5e-3 * want + 1e-9, key
        assert_close(st.Pl[:8, :8], fx[f"s{s}.Pl.corner"], atol=1e-6, rtol=1e-4, name="Pl corner")
        assert abs(torch.trace(st.Pl).item() - float(fx[f"s{s}.Pl.trace"])) < 1e-4
    return st


@pytest.mark.parametrize("tag", ["small_intended", "small_published", "small_legacy"])
def test_oracle_step_vs_reference_golden(tag, golden_dir):
    st = _run(tag, golden_dir)
    if tag == "small_published":
        assert torch.equal(st.Pl, torch.eye(512))      # Q1: the published projection never fires


def test_oracle_full_size_b2(golden_dir):
    _run("full_b2", golden_dir)


@pytest.mark.parametrize("D", [512, 768])
def test_oracle_gs_known_answers(D, golden_dir):
    fx = np.load(os.path.join(golden_dir, f"gs_kat_d{D}.npz"))
    D_, C, B, calls, seed = [int(v) for v in fx["meta"]]
    Pl = torch.eye(D)
    for i in range(calls):
        X = O.portable_normal(seed + i, (B, D), stream=5, mean=0.3, std=0.7).abs()
        G = O.portable_normal(seed + i, (C, D), stream=6, std=0.05)
        Pl, Gp = O.gs_before_update(Pl, X, G, i % 7, 7, i, "as_intended")
        assert_close(Gp, fx[f"c{i}.G"], atol=1e-8, rtol=1e-4, name=f"G {i}")
        assert_close(Pl[::16, ::16], fx[f"c{i}.Pl.sub"], atol=1e-8, rtol=1e-4, name=f"Pl {i}")
        assert abs(torch.linalg.norm(Pl).item() - float(fx[f"c{i}.Pl.fro"])) < 1e-5     # Q2: ||Pl||_F == 1 after firing
    assert abs(float(fx["c0.Pl.fro"]) - D ** 0.5) < 1e-3                                 # first call is skipped (Q5)


def test_portable_prng_is_stable():
    """The fixtures hold outputs only; inputs are regenerated, so the generator must never change."""
    x = O.portable_normal(7, (5,), stream=3)
    want = torch.tensor([0.45811594, -1.3419596, -0.26062322, 0.2964195, -0.21338835])
    got_again = O.portable_normal(7, (5,), stream=3)
    assert torch.equal(x, got_again)
    assert O.portable_labels(3, 10, 6).tolist() == O.portable_labels(3, 10, 6).tolist()
    assert torch.isfinite(x).all() and x.dtype == torch.float32 and want.shape == x.shape


def test_explicit_backward_matches_autograd():
    """The oracle's hand-written backward vs torch autograd on the same functional forward (tiny)."""
    import torch.nn.functional as F
    p = O.make_resnet18_params("audio", 3)
    x = O.portable_normal(1, (2, 1, 64, 32), stream=1)
    y, cache = O.resnet18_fwd({k: v.clone() for k, v in p.items()}, x, "audio", update_running=False)
    dout = O.portable_normal(2, tuple(y.shape), stream=1)
    g = O.resnet18_bwd(p, cache, dout)
    # autograd reference with nn.functional (training-mode batch_norm)
    q = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in p.items()}

    def bn(t, n):
        return F.batch_norm(t, None, None, q[n + ".weight"], q[n + ".bias"], True, 0.1, 1e-5)

    t = F.max_pool2d(torch.relu(bn(F.conv2d(x, q["conv1.weight"], None, 2, 3), "bn1")), 3, 2, 1)
    inpl = 64
    for li, planes in enumerate([64, 128, 256, 512], start=1):
        for bi in range(2):
            pre, stride = f"layer{li}.{bi}", (2 if (li > 1 and bi == 0) else 1)
            o = torch.relu(bn(F.conv2d(t, q[pre + ".conv1.weight"], None, stride, 1), pre + ".bn1"))
            o = bn(F.conv2d(o, q[pre + ".conv2.weight"], None, 1, 1), pre + ".bn2")
            idn = t
            if bi == 0 and (stride != 1 or inpl != planes):
                idn = bn(F.conv2d(t, q[pre + ".downsample.0.weight"], None, stride, 0), pre + ".downsample.1")
            t = torch.relu(o + idn)
            inpl = planes
    (t * dout).sum().backward()
    for k, v in g.items():
        ref = q[k].grad
        err = (v - ref).norm().item() / max(ref.norm().item(), 1e-30)
        assert err < 2e-4, (k, err)


def test_oracle_m3ae_step_vs_reference_golden(golden_dir):
    """Row a8: the oracle's M3AE MLA step vs the outputs of the reference's own m3ae.py modules (2 steps, text + image)."""
    fx = np.load(os.path.join(golden_dir, "m3ae_small.npz"))
    B, depth, vocab, C, steps, seed = [int(v) for v in fx["meta"]]
    st = O.M3AEState(O.make_m3ae_params(seed, depth=depth, vocab=vocab), O.make_m3ae_params(seed + 1, depth=depth, vocab=vocab),
                     O.make_head_params(768, C, seed + 2))
    for s in range(steps):
        token = torch.from_numpy(np.minimum((O.portable_uniform(seed + 50 + s, B * 256, 7) * vocab).astype(np.int64), vocab - 1)).view(B, 1, 256)
        pm = torch.zeros(B, 1, 256)
        for b in range(B):
            pm[b, 0, 40 + 37 * b:] = 1.0
        image = O.portable_normal(seed + 50 + s, (B, 3, 256, 256), stream=3)
        label = O.portable_labels(seed + 50 + s, B, C)
        out = O.mla_step_m3ae(st, token, pm, image, label, s, 10)
        for k in ("feat_a", "feat_v", "out_a", "out_v", "loss_a", "loss_v", "head_grad_a_raw", "head_grad_v_raw"):
            assert_close(out[k], fx[f"s{s}.{k}"], atol=2e-4, name=f"m3ae s{s} {k}")
        # projected head gradient: north-star absolute tolerance (ill-conditioned reference arithmetic, see make_golden.py)
        assert_close(out["head_grad_v"], fx[f"s{s}.head_grad_v"], atol=1e-3, name=f"m3ae s{s} projected head grad")
        for key in fx.files:
            if key.startswith(f"s{s}.grad.") and key.endswith(".abssum"):
                _, _, nm, rest = key.split(".", 3)
                g = out["grads_" + nm][rest[:-len(".abssum")]]
                want = float(fx[key])
                assert abs(g.double().abs().sum().item() - want) <= 1e-3 * want + 1e-9, key
        assert_close(st.text["cls_token"], fx[f"s{s}.text.cls_token"], atol=1e-5, name="text cls_token")
        assert_close(st.image["cls_token"], fx[f"s{s}.image.cls_token"], atol=1e-5, name="image cls_token")


def test_oracle_eval_vs_reference_golden(golden_dir):
    """Row 8f-1: eval-mode ResNet forward + pooling + head vs the reference AVClassifier in eval(); fusion restatement self-check."""
    fx = np.load(os.path.join(golden_dir, "eval_small.npz"))
    B, sh, sw, T, ih, iw, seed = [int(v) for v in fx["meta"]]
    pa, pv = O.make_resnet18_params("audio", seed), O.make_resnet18_params("visual", seed + 1)
    hd = O.make_head_params(512, 6, seed + 2)
    for params, off in ((pa, 0), (pv, 500)):
        for si, k in enumerate(sorted(k for k in params if k.endswith("running_mean"))):
            params[k] = O.portable_normal(seed, tuple(params[k].shape), stream=4000 + off + si, std=0.3)
            kv = k.replace("running_mean", "running_var")
            params[kv] = O.portable_normal(seed, tuple(params[kv].shape), stream=4250 + off + si, std=0.2).abs() + 0.5
    spec = O.portable_normal(seed + 9, (B, sh, sw), stream=1, mean=-5.081, std=4.4849)
    image = O.portable_normal(seed + 9, (B, 3, T, ih, iw), stream=2)
    label = O.portable_labels(seed + 9, B, 6)
    a, v = O.av_pool_fwd(O.resnet18_eval_fwd(pa, spec.unsqueeze(1), "audio"), O.resnet18_eval_fwd(pv, image, "visual"), B)
    assert_close(a, fx["a"], atol=2e-4, rtol=1e-5, name="eval a")
    assert_close(v, fx["v"], atol=2e-4, rtol=1e-5, name="eval v")
    out_a, out_v = a @ hd["weight"].t() + hd["bias"], v @ hd["weight"].t() + hd["bias"]
    assert_close(out_a, fx["out_a"], atol=2e-4, rtol=1e-5, name="eval logits a")
    for tag, dyn in (("dynamic", True), ("fixed", False)):
        w, counts = O.valid_batch([torch.from_numpy(fx["out_a"]), torch.from_numpy(fx["out_v"])], label, 6, dyn, [0.5, 0.5])
        assert np.allclose(w, fx[f"{tag}.weights"], atol=1e-6) and (counts.numpy() == fx[f"{tag}.counts"]).all()
        assert abs(sum(w) - 1.0) < 1e-6 and int(counts[0].sum()) == B


def test_ogm_oracle_vs_reference_vectors(golden_dir):
    """OGM / OGM-GE: the oracle's restatement of main.py:314-337, 373-408 against the vectors make_golden.py
    produced around the reference's modules (coefficients for both dominance orders, three alphas, three modalities; OGM
    scaling of the reference's own conv gradients; the OGM-GE noise scale grad.std() + 1e-8)."""
    fx = np.load(os.path.join(golden_dir, "ogm_kat.npz"))
    label = torch.from_numpy(fx["label"])
    for cname in ("ref", "swapped", "sharp"):
        oa, ov = torch.from_numpy(fx[f"{cname}.out_a"]), torch.from_numpy(fx[f"{cname}.out_v"])
        for alpha in (0.1, 0.3, 0.8):
            want = fx[f"{cname}.alpha{alpha}"]
            cf, sc, ra = O.ogm_coefficients([oa, ov], label, alpha)
            got = [float(cf[0]), float(cf[1]), float(sc[0]), float(sc[1]), float(ra[0]), float(ra[1])]
            assert np.allclose(got, want, rtol=1e-6, atol=1e-7), (cname, alpha, got, want)
    oa, ov, ot = torch.from_numpy(fx["ref.out_a"]), torch.from_numpy(fx["ref.out_v"]), torch.from_numpy(fx["three.out_t"])
    for tag, outs in (("avt", [oa, ov, ot]), ("tva", [ot * 3, ov, oa])):
        cf, _sc, ra = O.ogm_coefficients(outs, label, 0.3)
        got = [float(c) for c in cf] + [float(r) for r in ra]
        assert np.allclose(got, fx[f"three.{tag}"], rtol=1e-6, atol=1e-7), (tag, got)
    coeff = {"audio_net": torch.tensor(float(fx["ref.alpha0.3"][0])), "visual_net": torch.tensor(float(fx["ref.alpha0.3"][1]))}
    for enc in ("audio_net", "visual_net"):
        grads = {k: torch.from_numpy(fx[f"grad.{enc}.{k}"]) for k in ("conv1.weight", "layer1.0.conv1.weight", "bn1.weight")}
        out = O.ogm_modulate(grads, coeff[enc], "OGM")
        for k in grads:
            assert torch.equal(out[k], torch.from_numpy(fx[f"ogm.{enc}.{k}"])), (enc, k)        # 4-D scaled, 1-D untouched
        g = torch.Generator().manual_seed(1)
        ge = O.ogm_modulate(grads, coeff[enc], "OGM_GE", generator=g)
        for k in ("conv1.weight", "layer1.0.conv1.weight"):
            noise = ge[k] - grads[k] * coeff[enc]
            sd = float(fx[f"ge_std.{enc}.{k}"])
            assert abs(noise.std().item() - sd) < 0.05 * sd and abs(noise.mean().item()) < 0.1 * sd
        assert torch.equal(ge["bn1.weight"], grads["bn1.weight"])
