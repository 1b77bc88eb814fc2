#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own modules.

Runs only in the build container (needs /root/reference, read-only).  It imports the
reference's `models.backbone`, `models.basic_model.AVClassifier`, `models.fusion_modules`
and `utils.utils.GSPlugin` unmodified (inert `sys.modules` stand-ins for the absent,
never-called `timm` / `ml_collections` / `torchvision` packages that `models/basic_model.py` imports at
module scope), drives them exactly as `main.py:419-476` does -- autograd `backward()`,
`torch.optim.SGD(momentum=0.9, weight_decay=1e-4)`, `GSPlugin.before_update` -- on inputs and
weights from the portable PRNG in `oracle/mla_oracle.py`, and

  1. asserts that every function of the oracle agrees with the reference on those runs,
  2. writes the reference's OUTPUTS as .npz fixtures (data only; no reference source text).

The fixtures are what `tests/test_oracle_golden.py` (CPU) and the GPU parity tests check
against when /root/reference is not present (the GPU box).

    python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import mla_oracle as O  # noqa: E402


def _install_stubs():
    """Inert stand-ins for packages imported (never called on this path) by models/basic_model.py."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:  # placeholder class objects so `from x import Y` succeeds
        def __init__(self, *a, **k):
            raise RuntimeError("stub: not available offline")

    class ConfigDict(dict):
        __getattr__ = dict.get
        __setattr__ = dict.__setitem__

        def copy_and_resolve_references(self):
            return ConfigDict(self)

    if "timm" not in sys.modules:
        mod("timm")
        mod("timm.models")
        mod("timm.models.layers", to_2tuple=lambda x: (x, x), trunc_normal_=None, DropPath=_Dummy)
        mod("timm.models.vision_transformer", Attention=_Dummy, Mlp=_Dummy, PatchEmbed=_Dummy, Block=_Dummy)
    if "torchvision" not in sys.modules:
        mod("torchvision", transforms=mod("torchvision.transforms"))
    if "ml_collections" not in sys.modules:
        mod("ml_collections", ConfigDict=ConfigDict)
        mod("ml_collections.config_dict", ConfigDict=ConfigDict, config_dict=types.SimpleNamespace(placeholder=lambda t: None))


import transformers  # noqa: E402,F401  real package; must be imported BEFORE the stubs (it probes __spec__)

_install_stubs()
from models.basic_model import AVClassifier  # noqa: E402  (reference)
from utils.utils import GSPlugin  # noqa: E402  (reference)


class _Args:
    fusion_method = "concat"
    dataset = "CREMAD"
    gs_flag = True
    modulation = "Normal"


class _Wrap(nn.Module):
    """Exposes fc_out as attribute `module`, so named_parameters() yields 'module.weight'
    and utils/utils.py:32-41 executes untouched ('as_intended')."""

    def __init__(self, m):
        super().__init__()
        self.module = m


def build_reference(seed):
    model = AVClassifier(_Args())
    pa, pv = O.make_resnet18_params("audio", seed), O.make_resnet18_params("visual", seed + 1)
    hd = O.make_head_params(512, 6, seed + 2)
    model.audio_net.load_state_dict(pa)
    model.visual_net.load_state_dict(pv)
    model.fusion_module.fc_out.load_state_dict(hd)
    return torch.nn.DataParallel(model), pa, pv, hd


def reference_step(model, optimizer, gs, spec, image, label, batch_step, len_dataloader,
                   gs_mode, legacy):
    """main.py:419-476 restated around the imported reference modules."""
    rec = {}
    model.train()
    optimizer.zero_grad()                                                   # main.py:164
    a, v = model(spec.unsqueeze(1).float(), image.float())                  # :431
    fc = model.module.fusion_module.fc_out
    target = _Wrap(fc) if gs_mode == "as_intended" else fc
    crit = nn.CrossEntropyLoss()
    rec["a"], rec["v"] = a.detach().clone(), v.detach().clone()
    for name, feat in (("a", a), ("v", v)):
        out = fc(feat)                                                      # :432 / :444
        loss = crit(out, label)
        loss.backward()                                                     # :435 / :447
        rec["out_" + name], rec["loss_" + name] = out.detach().clone(), loss.detach().clone()
        rec[f"head_grad_{name}_raw"] = fc.weight.grad.detach().clone()
        rec[f"head_bias_grad_{name}"] = fc.bias.grad.detach().clone()
        net = model.module.audio_net if name == "a" else model.module.visual_net
        rec["grads_" + ("audio" if name == "a" else "visual")] = {
            k: p.grad.detach().clone() for k, p in net.named_parameters()}
        gs.before_update(target, feat, batch_step, len_dataloader, gs.exp_count)   # :437 / :449
        rec[f"head_grad_{name}"] = fc.weight.grad.detach().clone()
        optimizer.step()                                                    # :439 / :451
        if legacy:
            optimizer.zero_grad(set_to_none=False)                          # torch 1.8.1 semantics (Q6)
        else:
            optimizer.zero_grad()
        gs.exp_count += 1
    for _n, p in model.named_parameters():                                  # :468-470
        if p.grad is not None:
            del p.grad
    rec["loss"] = (rec["loss_a"] * 0.55 + rec["loss_v"] * 0.45)             # :472 (Q8)
    return rec


def close(name, got, want, rtol=2e-4, atol=2e-6):
    got, want = torch.as_tensor(got).double(), torch.as_tensor(want).double()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    ok = err <= atol + rtol * ref
    print(f"  {'ok ' if ok else 'BAD'} {name:40s} max|d|={err:.3e} max|ref|={ref:.3e}")
    assert ok, name
    return err


def close_l2(name, got, want, tol=3e-3):
    """Norm-wise check for encoder gradients/weights: a single ReLU / max-pool decision that flips
    under 1e-7 weight differences moves isolated elements by a discrete amount, so element-wise
    max error is not meaningful there; the relative L2 error is."""
    got, want = torch.as_tensor(got).double(), torch.as_tensor(want).double()
    err = torch.linalg.norm((got - want).flatten()).item()
    ref = torch.linalg.norm(want.flatten()).item()
    ok = err <= tol * ref + 1e-12
    print(f"  {'ok ' if ok else 'BAD'} {name:40s} relL2={err / max(ref, 1e-30):.3e}")
    assert ok, name


def pl_digest(Pl):
    Pl = Pl.detach()
    return {"fro": torch.linalg.norm(Pl).item(), "trace": torch.trace(Pl).item(),
            "corner": Pl[:8, :8].clone().numpy(), "sub": Pl[::16, ::16].clone().numpy(),
            "rowsum": Pl.sum(1).numpy()}


def run_case(tag, B, spec_hw, T, img_hw, steps, gs_mode, legacy, seed, len_dataloader=10, keep_grads=()):
    print(f"== case {tag}: B={B} spec={spec_hw} T={T} img={img_hw} steps={steps} gs={gs_mode} legacy={legacy}")
    torch.manual_seed(0)
    model, pa, pv, hd = build_reference(seed)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)   # main.py:749
    gs = GSPlugin.__new__(GSPlugin)          # ctor needs a GPU (Q3); state set by hand
    gs.Pl = torch.eye(512)
    gs.exp_count = 0
    st = O.MLAState({k: v.clone() for k, v in pa.items()}, {k: v.clone() for k, v in pv.items()},
                    {k: v.clone() for k, v in hd.items()})
    fx = {}
    for s in range(steps):
        spec = O.portable_normal(seed + 100 + s, (B,) + spec_hw, stream=1, mean=-5.081, std=4.4849)
        image = O.portable_normal(seed + 100 + s, (B, 3, T) + img_hw, stream=2)
        label = O.portable_labels(seed + 100 + s, B, 6)
        ref = reference_step(model, opt, gs, spec, image, label, s, len_dataloader, gs_mode, legacy)
        orc = O.mla_step(st, spec, image, label, s, len_dataloader, gs_mode=gs_mode, legacy_zero_grad=legacy)
        for k in ("a", "v", "out_a", "out_v", "loss_a", "loss_v", "loss", "head_grad_a_raw",
                  "head_grad_v_raw", "head_grad_a", "head_grad_v"):
            close(f"s{s}.{k}", orc[k], ref[k])
            fx[f"s{s}.{k}"] = ref[k].numpy()
        for enc in ("audio", "visual"):
            for k, gref in ref["grads_" + enc].items():
                close_l2(f"s{s}.grad.{enc}.{k}", orc["grads_" + enc][k], gref)
            for k in keep_grads:
                gref = ref["grads_" + enc][k]
                fx[f"s{s}.grad.{enc}.{k}.sum"] = np.float64(gref.double().sum().item())
                fx[f"s{s}.grad.{enc}.{k}.abssum"] = np.float64(gref.double().abs().sum().item())
                fx[f"s{s}.grad.{enc}.{k}.head"] = gref.flatten()[:64].numpy()
        # post-step state
        sd = model.module.state_dict()
        for enc, params in (("audio_net", st.audio), ("visual_net", st.visual)):
            for k, vv in params.items():
                close_l2(f"s{s}.state.{enc}.{k}", vv, sd[f"{enc}.{k}"], tol=1e-4)
        close(f"s{s}.state.head.weight", st.head["weight"], sd["fusion_module.fc_out.weight"])
        close(f"s{s}.state.head.bias", st.head["bias"], sd["fusion_module.fc_out.bias"])
        close(f"s{s}.Pl", st.Pl, gs.Pl.detach(), rtol=1e-4, atol=1e-7)
        fx[f"s{s}.head.weight"] = sd["fusion_module.fc_out.weight"].numpy().copy()
        fx[f"s{s}.head.bias"] = sd["fusion_module.fc_out.bias"].numpy().copy()
        for enc in ("audio_net", "visual_net"):
            fx[f"s{s}.{enc}.bn1.running_mean"] = sd[f"{enc}.bn1.running_mean"].numpy().copy()
            fx[f"s{s}.{enc}.bn1.running_var"] = sd[f"{enc}.bn1.running_var"].numpy().copy()
            fx[f"s{s}.{enc}.conv1.weight"] = sd[f"{enc}.conv1.weight"].numpy().copy()
            w = sd[f"{enc}.layer4.1.conv2.weight"]
            fx[f"s{s}.{enc}.layer4.1.conv2.weight.sum"] = np.float64(w.double().sum().item())
            fx[f"s{s}.{enc}.layer4.1.conv2.weight.head"] = w.flatten()[:64].numpy().copy()
        for k, vv in pl_digest(gs.Pl).items():
            fx[f"s{s}.Pl.{k}"] = np.asarray(vv)
    fx["meta"] = np.array([B, spec_hw[0], spec_hw[1], T, img_hw[0], img_hw[1], steps, seed, len_dataloader], dtype=np.int64)
    fx["gs_mode"] = np.array(gs_mode)
    fx["legacy"] = np.array(int(legacy))
    path = os.path.join(HERE, f"mla_{tag}.npz")
    np.savez_compressed(path, **fx)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def run_gs_kat(D, C, B, calls, seed):
    """Known-answer trajectory of GSPlugin.before_update alone (utils/utils.py:24-41), D=512 and 768."""
    print(f"== gs KAT D={D} C={C} B={B} calls={calls}")
    fc = nn.Linear(D, C)
    gs = GSPlugin.__new__(GSPlugin)
    gs.Pl = torch.eye(D)
    gs.exp_count = 0
    Pl = torch.eye(D)
    fx = {}
    for i in range(calls):
        X = O.portable_normal(seed + i, (B, D), stream=5, mean=0.3, std=0.7).abs()   # pooled relu features are >= 0
        G = O.portable_normal(seed + i, (C, D), stream=6, std=0.05)
        fc.weight.grad = G.clone()
        gs.before_update(_Wrap(fc), X, i % 7, 7, gs.exp_count)
        gs.exp_count += 1
        Pl, Gp = O.gs_before_update(Pl, X, G, i % 7, 7, i, "as_intended")
        close(f"gs{i}.G", Gp, fc.weight.grad, rtol=1e-4, atol=1e-8)
        close(f"gs{i}.Pl", Pl, gs.Pl.detach(), rtol=1e-4, atol=1e-8)
        fx[f"c{i}.G"] = fc.weight.grad.detach().numpy().copy()
        for k, vv in pl_digest(gs.Pl).items():
            fx[f"c{i}.Pl.{k}"] = np.asarray(vv)
    fx["meta"] = np.array([D, C, B, calls, seed], dtype=np.int64)
    path = os.path.join(HERE, f"gs_kat_d{D}.npz")
    np.savez_compressed(path, **fx)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ----------------------------------------------------------------------------------------------
# M3AE: the reference's own Transformer / Block / Attention /
# TransformerMLP / embeddings (models/m3ae.py), DropPath == identity (Q10: the published DropPath returns
# None), forward_representation (m3ae.py:342-370) and M3AEClassifier.forward (basic_model.py:182-200)
# restated device-free around them (both hard-code cuda:0), main.py:419-476 restated as above.
# ----------------------------------------------------------------------------------------------
def run_m3ae_case(tag, B, depth, vocab, n_classes, steps, seed):
    import einops
    import models.m3ae as RM
    print(f"== m3ae case {tag}: B={B} depth={depth} vocab={vocab} classes={n_classes} steps={steps}")
    RM.DropPath.forward = lambda self, input, deterministic=False: input           # Q10
    cfg = dict(model_type=None, emb_dim=768, depth=depth, num_heads=12, mlp_ratio=4, dec_emb_dim=512, dec_depth=1, dec_num_heads=16)

    def build(pseed):
        m = RM.MaskedMultimodalAutoencoder(text_vocab_size=vocab, config_updates=cfg)
        params = O.make_m3ae_params(pseed, depth=depth, vocab=vocab)
        missing, unexpected = m.load_state_dict(params, strict=True), None
        return m, params

    def fwd_rep(m, image=None, text=None, text_padding_mask=None):                  # m3ae.py:342-370, device-free
        D = m.config.emb_dim
        bs = image.shape[0] if image is not None else text.shape[0]
        xs, pms = [m.cls_token.expand(bs, 1, D)], [torch.zeros((bs, 1))]
        if image is not None:
            xs.append(m.image_embedding(image) + torch.tensor(RM.get_2d_sincos_pos_embed(D, image.shape[1]))
                      + m.get_type_embedding('encoder_image_type_embedding'))
            pms.append(torch.zeros((bs, image.shape[1])))
        if text is not None:
            xs.append(m.text_embedding(text) + torch.tensor(RM.get_1d_sincos_pos_embed(D, text.shape[1]))
                      + m.get_type_embedding('encoder_text_type_embedding'))
            pms.append(text_padding_mask)
        return m.encoder(torch.cat(xs, dim=1), False, torch.cat(pms, dim=1))

    mae_a, pa = build(seed)
    mae_v, pv = build(seed + 1)
    hd = O.make_head_params(768, n_classes, seed + 2)
    fc = nn.Linear(768, n_classes)
    fc.load_state_dict(hd)
    opt = torch.optim.SGD(list(mae_a.parameters()) + list(mae_v.parameters()) + list(fc.parameters()),
                          lr=1e-3, momentum=0.9, weight_decay=1e-4)
    gs = GSPlugin.__new__(GSPlugin)
    gs.Pl = torch.eye(768)
    gs.exp_count = 0
    st = O.M3AEState({k: v.clone() for k, v in pa.items()}, {k: v.clone() for k, v in pv.items()}, {k: v.clone() for k, v in hd.items()})
    crit = nn.CrossEntropyLoss()
    fx = {}
    for s in range(steps):
        token = torch.from_numpy(np.minimum((O.portable_uniform(seed + 50 + s, B * 256, 7) * vocab).astype(np.int64), vocab - 1)).view(B, 1, 256)
        lens = [40 + 37 * b for b in range(B)]
        pm = torch.zeros(B, 1, 256)
        for b in range(B):
            pm[b, 0, lens[b]:] = 1.0
        image = O.portable_normal(seed + 50 + s, (B, 3, 256, 256), stream=3)
        label = O.portable_labels(seed + 50 + s, B, n_classes)
        # ---- reference (main.py:419-476, lorb == 'm3ae', non-modal3; basic_model.py:182-200)
        opt.zero_grad()
        visual = einops.rearrange(image, 'b c (h p1) (w p2) -> b (h w) (c p1 p2)', p1=16, p2=16)
        a = fwd_rep(mae_a, None, token.squeeze(1), pm.squeeze(1)).mean(dim=1)
        v = fwd_rep(mae_v, visual, None, None).mean(dim=1)
        rec = {"feat_a": a.detach().clone(), "feat_v": v.detach().clone()}
        for name, feat, net in (("a", a, mae_a), ("v", v, mae_v)):
            out = fc(feat)
            loss = crit(out, label)
            loss.backward()
            rec["out_" + name], rec["loss_" + name] = out.detach().clone(), loss.detach().clone()
            rec[f"head_grad_{name}_raw"] = fc.weight.grad.detach().clone()
            rec["grads_" + name] = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
            gs.before_update(_Wrap(fc), feat, s, 10, gs.exp_count)
            rec[f"head_grad_{name}"] = fc.weight.grad.detach().clone()
            opt.step()
            opt.zero_grad()
            gs.exp_count += 1
        # ---- oracle
        orc = O.mla_step_m3ae(st, token, pm, image, label, s, 10)
        for k in ("feat_a", "feat_v", "out_a", "out_v", "loss_a", "loss_v", "head_grad_a_raw", "head_grad_v_raw",
                  "head_grad_a", "head_grad_v"):
            # projected gradients: the element-wise denominator alpha + k_i r_j of utils/utils.py:36 (Q2) can be
            # close to 0, which amplifies fp32 re-association (measured up to 2.2e-4 absolute at D=768) -> they are
            # held to the north-star tolerance, 1e-3 absolute
            if k in ("head_grad_a", "head_grad_v"):
                close(f"s{s}.{k}", orc[k], rec[k], rtol=0.0, atol=1e-3)
            else:
                close(f"s{s}.{k}", orc[k], rec[k], rtol=2e-4, atol=2e-6)
            fx[f"s{s}.{k}"] = rec[k].numpy()
        for name in ("a", "v"):
            assert set(orc["grads_" + name]) == set(rec["grads_" + name]), "set of parameters that receive a gradient"
            for k, gref in rec["grads_" + name].items():
                close_l2(f"s{s}.grad.{name}.{k}", orc["grads_" + name][k], gref, tol=2e-4)
                fx[f"s{s}.grad.{name}.{k}.abssum"] = np.float64(gref.double().abs().sum().item())
                fx[f"s{s}.grad.{name}.{k}.head"] = gref.flatten()[:32].numpy().copy()
        for net, params, nm in ((mae_a, st.text, "text"), (mae_v, st.image, "image")):
            sd = net.state_dict()
            for k in params:
                close_l2(f"s{s}.state.{nm}.{k}", params[k], sd[k], tol=1e-5)
            fx[f"s{s}.{nm}.cls_token"] = sd["cls_token"].numpy().copy()
            fx[f"s{s}.{nm}.fc2w.head"] = sd[f"encoder.blocks.{depth - 1}.transformer_mlp.fc2.weight"].flatten()[:64].numpy().copy()
        # With mixed-sign transformer features alpha + k_i r_j (Q2) gets close to 0 and the renormalised projector
        # collapses onto a few huge entries (max |Pl| = 0.93 here): ill-conditioned in the reference itself.
        close(f"s{s}.Pl", st.Pl, gs.Pl.detach(), rtol=0.0, atol=5e-3)
        fx[f"s{s}.head.weight"] = fc.weight.detach().numpy().copy()
        for k, vv in pl_digest(gs.Pl).items():
            fx[f"s{s}.Pl.{k}"] = np.asarray(vv)
    fx["meta"] = np.array([B, depth, vocab, n_classes, steps, seed], dtype=np.int64)
    path = os.path.join(HERE, f"m3ae_{tag}.npz")
    np.savez_compressed(path, **fx)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def run_eval_case(B, spec_hw, T, img_hw, seed):
    """Evaluation path (main.py:486-679, gs_flag branch): the reference AVClassifier in eval() mode (BatchNorm on running
    statistics) + fc_out per modality; the fusion/accuracy arithmetic of main.py:65-106, 640-676 is restated in the oracle
    (main.py itself cannot be imported: tensorboard / torchvision / cuda:0)."""
    print(f"== eval case: B={B} spec={spec_hw} T={T} img={img_hw}")
    model, pa, pv, hd = build_reference(seed)
    for net, params, off in ((model.module.audio_net, pa, 0), (model.module.visual_net, pv, 500)):
        sd = net.state_dict()
        for si, k in enumerate(sorted(k for k in sd if k.endswith("running_mean"))):
            params[k] = O.portable_normal(seed, tuple(sd[k].shape), stream=4000 + off + si, std=0.3)
            kv = k.replace("running_mean", "running_var")
            params[kv] = O.portable_normal(seed, tuple(sd[kv].shape), stream=4250 + off + si, std=0.2).abs() + 0.5
        net.load_state_dict(params)
    model.eval()
    spec = O.portable_normal(seed + 9, (B,) + spec_hw, stream=1, mean=-5.081, std=4.4849)
    image = O.portable_normal(seed + 9, (B, 3, T) + img_hw, stream=2)
    label = O.portable_labels(seed + 9, B, 6)
    with torch.no_grad():
        a, v = model(spec.unsqueeze(1).float(), image.float())                      # main.py:633
        fc = model.module.fusion_module.fc_out
        out_a, out_v = fc(a), fc(v)                                                  # main.py:636-637
    fa = O.resnet18_eval_fwd(pa, spec.unsqueeze(1), "audio")
    fv = O.resnet18_eval_fwd(pv, image, "visual")
    oa, ov = O.av_pool_fwd(fa, fv, B)
    close("eval.a", oa, a)
    close("eval.v", ov, v)
    fx = {"a": a.numpy(), "v": v.numpy(), "out_a": out_a.numpy(), "out_v": out_v.numpy(),
          "meta": np.array([B, spec_hw[0], spec_hw[1], T, img_hw[0], img_hw[1], seed], dtype=np.int64)}
    for name, dyn in (("dynamic", True), ("fixed", False)):
        w, counts = O.valid_batch([out_a, out_v], label, 6, dyn, [0.5, 0.5])
        fx[f"{name}.weights"] = np.array(w, dtype=np.float64)
        fx[f"{name}.counts"] = counts.numpy()
    path = os.path.join(HERE, "eval_small.npz")
    np.savez_compressed(path, **fx)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")



def run_ogm_kat(seed=81, B=6, spec_hw=(96, 64), T=2, img_hw=(64, 64)):
    """OGM / OGM-GE (main.py:268-410, the non --gs_flag branch).  The modulation code is inline in
    `train_epoch` (main.py is not importable here), so this restates main.py:297-311 and 373-408 around the REFERENCE's
    own modules: `AVClassifier` with gs_flag=False (ConcatFusion(1024 -> 6), basic_model.py:31-34) gives a, v, out;
    out_a / out_v are the half-weight products of main.py:297-301; `loss.backward()` gives the encoder gradients; the
    coefficients and the scaled gradients use torch's nn.Softmax / nn.Tanh / nn.ReLU modules as main.py:131-133 does.
    Fixture: logits, labels, coefficients / scores / ratios for alpha in {0.1, 0.3, 0.8} and both orderings (audio- and
    visual-dominant), plus two conv gradients before / after OGM scaling and the noise scale `grad.std() + 1e-8` OGM_GE uses."""
    print("== OGM / OGM-GE known-answer vectors")

    class A(_Args):
        gs_flag = False
    torch.manual_seed(0)
    model = AVClassifier(A())
    pa, pv = O.make_resnet18_params("audio", seed), O.make_resnet18_params("visual", seed + 1)
    model.audio_net.load_state_dict(pa)
    model.visual_net.load_state_dict(pv)
    hd = O.make_head_params(1024, 6, seed + 2)
    model.fusion_module.fc_out.load_state_dict(hd)
    model = torch.nn.DataParallel(model)
    model.train()
    spec = O.portable_normal(seed, (B,) + spec_hw, stream=1, mean=-5.081, std=4.4849)
    image = O.portable_normal(seed, (B, 3, T) + img_hw, stream=2)
    label = O.portable_labels(seed, B, 6)
    softmax, relu, tanh = nn.Softmax(dim=1), nn.ReLU(inplace=True), nn.Tanh()                    # main.py:131-133
    a, v, out = model(spec.unsqueeze(1).float(), image.float())                                  # :273
    weight_size = model.module.fusion_module.fc_out.weight.size(1)                               # :297
    out_v = (torch.mm(v, torch.transpose(model.module.fusion_module.fc_out.weight[:, weight_size // 2:], 0, 1))
             + model.module.fusion_module.fc_out.bias / 2)                                       # :298-299
    out_a = (torch.mm(a, torch.transpose(model.module.fusion_module.fc_out.weight[:, :weight_size // 2], 0, 1))
             + model.module.fusion_module.fc_out.bias / 2)                                       # :301-302
    loss = nn.CrossEntropyLoss()(out, label)                                                     # :305
    loss.backward()                                                                              # :310
    fx = {"meta": np.array([B, 6, seed], dtype=np.int64), "label": label.numpy()}
    cases = {"ref": (out_a.detach(), out_v.detach()), "swapped": (out_v.detach(), out_a.detach()),
             "sharp": (out_a.detach() * 4.0, out_v.detach() * 0.25)}
    for cname, (oa, ov) in cases.items():
        fx[f"{cname}.out_a"], fx[f"{cname}.out_v"] = oa.numpy(), ov.numpy()
        for alpha in (0.1, 0.3, 0.8):
            score_v = sum([softmax(ov)[i][label[i]] for i in range(ov.size(0))])                 # :373
            score_a = sum([softmax(oa)[i][label[i]] for i in range(oa.size(0))])                 # :374
            ratio_v = score_v / score_a                                                          # :376
            ratio_a = 1 / ratio_v                                                                # :377
            if ratio_v > 1:                                                                      # :379-384
                coeff_v = 1 - tanh(alpha * relu(ratio_v))
                coeff_a = 1
            else:
                coeff_a = 1 - tanh(alpha * relu(ratio_a))
                coeff_v = 1
            cf, sc, ra = O.ogm_coefficients([oa, ov], label, alpha)
            close(f"ogm.{cname}.a{alpha}.coeff_a", cf[0], torch.as_tensor(coeff_a, dtype=torch.float32), rtol=1e-6, atol=1e-7)
            close(f"ogm.{cname}.a{alpha}.coeff_v", cf[1], torch.as_tensor(coeff_v, dtype=torch.float32), rtol=1e-6, atol=1e-7)
            fx[f"{cname}.alpha{alpha}"] = np.array([float(coeff_a), float(coeff_v), float(score_a), float(score_v), float(ratio_a),
                                                   float(ratio_v)], dtype=np.float64)
    # three modalities (main.py:314-337): a third set of logits stands in for out_t
    out_t = O.portable_normal(seed, (B, 6), stream=7)
    fx["three.out_t"] = out_t.numpy()
    for alpha in (0.3,):
        oa, ov = cases["ref"]
        for tag, (x0, x1, x2) in {"avt": (oa, ov, out_t), "tva": (out_t * 3, ov, oa)}.items():
            score_v = sum([softmax(x1)[i][label[i]] for i in range(B)])
            score_a = sum([softmax(x0)[i][label[i]] for i in range(B)])
            score_t = sum([softmax(x2)[i][label[i]] for i in range(B)])
            ratio_v = score_v / (score_a + score_t)
            ratio_a = score_a / (score_v + score_t)
            ratio_t = score_t / (score_v + score_a)
            if ratio_v > 1:
                c3 = [1, 1 - tanh(alpha * relu(ratio_v)), 1]
            elif ratio_t > 1:
                c3 = [1, 1, 1 - tanh(alpha * relu(ratio_t))]
            else:
                c3 = [1 - tanh(alpha * relu(ratio_a)), 1, 1]
            cf, _sc, _ra = O.ogm_coefficients([x0, x1, x2], label, alpha)
            for k in range(3):
                close(f"ogm.three.{tag}.coeff{k}", cf[k], torch.as_tensor(c3[k], dtype=torch.float32), rtol=1e-6, atol=1e-7)
            fx[f"three.{tag}"] = np.array([float(c) for c in c3] + [float(ratio_a), float(ratio_v), float(ratio_t)], dtype=np.float64)
    # gradient modulation on the reference's own gradients (main.py:394-408), alpha = 0.3 on the "ref" logits
    coeff_a, coeff_v = [torch.tensor(float(x), dtype=torch.float32) for x in fx["ref.alpha0.3"][:2]]
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    for name, parms in model.named_parameters():
        layer = str(name).split('.')[1]                                                          # :395
        if 'audio' in layer and len(parms.grad.size()) == 4:                                     # :397
            parms.grad *= coeff_a                                                                # :402 (OGM)
        if 'visual' in layer and len(parms.grad.size()) == 4:
            parms.grad *= coeff_v
    mod_a = O.ogm_modulate({k[len("module.audio_net."):]: g for k, g in grads.items() if k.startswith("module.audio_net.")}, coeff_a, "OGM")
    for k, gref in mod_a.items():
        close(f"ogm.scaled.audio.{k}", gref, dict(model.named_parameters())["module.audio_net." + k].grad, rtol=0, atol=0)
    for enc in ("audio_net", "visual_net"):
        for k in ("conv1.weight", "layer1.0.conv1.weight", "bn1.weight"):
            g0, g1 = grads[f"module.{enc}.{k}"], dict(model.named_parameters())[f"module.{enc}.{k}"].grad
            fx[f"grad.{enc}.{k}"], fx[f"ogm.{enc}.{k}"] = g0.numpy(), g1.numpy()
            if g0.dim() == 4:
                fx[f"ge_std.{enc}.{k}"] = np.float64(g0.std().item() + 1e-8)                     # :399 noise scale
    fx["spec_hw"], fx["img"] = np.array(spec_hw), np.array((T,) + img_hw)
    path = os.path.join(HERE, "ogm_kat.npz")
    np.savez_compressed(path, **fx)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


if __name__ == "__main__":
    torch.set_num_threads(8)
    keep = ("conv1.weight", "bn1.weight", "bn1.bias", "layer1.0.conv1.weight", "layer2.0.downsample.0.weight",
            "layer2.0.conv1.weight", "layer4.1.conv2.weight", "layer4.1.bn2.weight")
    run_case("small_intended", 4, (128, 64), 2, (96, 96), 2, "as_intended", False, seed=7, keep_grads=keep)
    run_case("small_published", 4, (128, 64), 2, (96, 96), 2, "as_published", False, seed=7, keep_grads=keep)
    run_case("small_legacy", 4, (128, 64), 2, (96, 96), 2, "as_intended", True, seed=7, keep_grads=keep)
    run_case("full_b2", 2, (1024, 128), 3, (224, 224), 1, "as_intended", False, seed=11, keep_grads=keep)
    run_gs_kat(512, 6, 64, 5, seed=21)
    run_gs_kat(768, 4, 32, 3, seed=23)
    run_m3ae_case("small", 3, 2, 1000, 11, 2, seed=61)
    run_eval_case(16, (128, 64), 2, (96, 96), seed=71)
    run_ogm_kat()
    print("all oracle-vs-reference checks passed")
