"""Shared comparison helpers for the parity tests (tolerances are stated where they are used)."""
import torch


def max_abs_err(got, want):
    return (torch.as_tensor(got).double().cpu() - torch.as_tensor(want).double().cpu()).abs().max().item()


def assert_close(got, want, atol, rtol=0.0, name=""):
    """|got-want| <= atol + rtol*max|want| element-wise (north-star outputs: logits, loss, features,
    head gradients -- continuous functions of the inputs)."""
    got, want = torch.as_tensor(got).double().cpu(), torch.as_tensor(want).double().cpu()
    assert got.shape == want.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    err = (got - want).abs().max().item()
    bound = atol + rtol * want.abs().max().item()
    assert err <= bound, f"{name}: max|d|={err:.3e} > {bound:.3e} (max|ref|={want.abs().max().item():.3e})"
    return err


def assert_close_robust(got, want, rel_l2, elem_tol, frac=0.97, name=""):
    """For encoder gradients / updated encoder weights.  fp32 re-association can flip a single ReLU or
    max-pool decision (probability ~ #elements * 1e-7), which moves one channel's gradient by a
    discrete amount; that is not an arithmetic error.  So: relative L2 error <= rel_l2 AND at least
    `frac` of the elements within elem_tol * max|want|."""
    got, want = torch.as_tensor(got).double().cpu().flatten(), torch.as_tensor(want).double().cpu().flatten()
    assert got.shape == want.shape, f"{name}: shape"
    ref = torch.linalg.norm(want).item()
    err = torch.linalg.norm(got - want).item()
    assert err <= rel_l2 * ref + 1e-12, f"{name}: relL2={err / max(ref, 1e-30):.3e} > {rel_l2:.1e}"
    within = ((got - want).abs() <= elem_tol * want.abs().max().item() + 1e-12).double().mean().item()
    assert within >= frac, f"{name}: only {within:.4f} of elements within {elem_tol:.1e}*max"
    return err / max(ref, 1e-30)


def oracle_cache_from_hip(enc, params_before=None):
    """Rebuild the oracle's backward cache (NCHW, oracle/mla_oracle.py:resnet18_fwd) from the HIP encoder's
    saved forward state.  Feeding it to O.resnet18_bwd gives a backward that uses exactly the same ReLU /
    max-pool decisions as the HIP path, so gradients can be compared element-wise (flip-immune).
    params_before: the encoder's reference-layout parameters at the time of that forward (the stem's ReLU output is never
    stored by the HIP path; it is rebuilt from the saved conv output, which needs bn1's affine parameters of that moment)."""
    ws = enc._ws
    to = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu()
    st = lambda bn: tuple(v.cpu() for v in ws["stats"][bn])
    cache = {"modality": enc.modality, "x0": to(ws["x0"]), "stem_relu": to(enc.stem_relu(*((params_before["bn1.weight"], params_before["bn1.bias"]) if params_before else ())))}
    cache["bn1"] = (to(ws["y_stem"]),) + st("bn1")
    N, H, W, C = ws["y_stem"].shape
    code = to(ws["pool_idx"]).long()                              # (N,C,OH,OW), kh*3+kw
    OH, OW = code.shape[2:]
    oy = torch.arange(OH).view(1, 1, OH, 1)
    ox = torch.arange(OW).view(1, 1, 1, OW)
    cache["pool_idx"] = (oy * 2 - 1 + code // 3) * W + (ox * 2 - 1 + code % 3)
    for blk in ws["blocks"]:
        pre = blk["pre"]
        a1 = enc.block_a1(blk, *((params_before[pre + ".bn1.weight"], params_before[pre + ".bn1.bias"]) if params_before else ()))
        cache[pre + ".in"], cache[pre + ".out"], cache[pre + ".a1"] = to(blk["xin"]), to(blk["out"]), to(a1)
        cache[pre + ".bn1"] = (to(blk["y1"]),) + st(pre + ".bn1")
        cache[pre + ".bn2"] = (to(blk["y2"]),) + st(pre + ".bn2")
        if blk["ds"]:
            cache[pre + ".downsample.1"] = (to(blk["yd"]),) + st(pre + ".downsample.1")
    return cache


def sync_oracle_state_from_hip(st, model, tr):
    """Copy parameters, BN buffers, SGD momentum, Pl and exp_count from the HIP trainer into the oracle's
    MLAState, so that the next step is compared from an identical starting point (a flipped ReLU decision in
    one step otherwise makes the two trajectories drift apart chaotically)."""
    import math
    for name, net, params in (("audio", model.audio_net, st.audio), ("visual", model.visual_net, st.visual)):
        for k, v in net.state_dict().items():
            params[k] = v.cpu().clone()
        buf = tr.optimizer.buf[name]
        for k, (o, shape) in net.layout.items():
            t = buf[o:o + math.prod(shape)].view(shape)
            st.mom[name][k] = (t.permute(3, 2, 0, 1) if t.dim() == 4 else t).contiguous().cpu().clone()
    head = model.fusion_module.fc_out
    st.head["weight"], st.head["bias"] = head.weight.cpu().clone(), head.bias.cpu().clone()
    hb = tr.optimizer.buf["head"]
    n = head.weight.numel()
    st.mom["head"] = {"weight": hb[:n].view_as(head.weight).cpu().clone(), "bias": hb[n:].cpu().clone()}
    st.Pl = tr.gs_plugin.Pl.cpu().clone()
    st.exp_count = tr.gs_plugin.exp_count
