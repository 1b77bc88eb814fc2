"""CPU oracle for the MLA alternating-unimodal training step (TEST INFRASTRUCTURE ONLY).

This file is the *checker*, never the product.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The shipped path
(``mla_hip``) never imports anything from ``oracle/`` and fails loudly when the HIP
library is missing.

It restates, in reference layout (NCHW activations, OIHW weights), with *explicit*
forward and backward formulas (no autograd), the arithmetic that the reference
triggers on the ``--gs_flag`` branch of ``main.py:419-476``:

  * ResNet-18 trunk            models/backbone.py:36-52 (BasicBlock), 142-160 (ResNet.forward)
  * pooling + flatten          models/basic_model.py:52-77 (AVClassifier.forward)
  * shared head + CE           models/fusion_modules.py:16-19 (fc_out), main.py:130, 432-435
  * head-gradient projection   utils/utils.py:24-41 (GSPlugin.before_update)
  * SGD momentum + wd          main.py:749 (torch.optim.SGD), 439-440, 451-452
  * the step orchestration     main.py:419-476

Parity pin: the reference holds no golden vectors or tests, so this
oracle is pinned against outputs of the reference's own modules run in the build
container: ``tests/golden/make_golden.py`` imports ``/root/reference`` (models.backbone,
models.basic_model.AVClassifier, utils.utils.GSPlugin.before_update), drives them with
autograd + torch.optim.SGD exactly as main.py does, checks every function below against
them and writes the fixtures under ``tests/golden/``.  ``tests/test_oracle_golden.py``
re-checks this oracle against those fixtures without the reference present.

Conv contractions use torch's CPU convolution primitives (F.conv2d and
torch.nn.grad.conv2d_input/weight): these are the same ATen kernels the reference's
CPU path executes, so the oracle is also the fairest CPU baseline ("port").
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# --------------------------------------------------------------------------------------
# portable counter-based PRNG (splitmix64 + Box-Muller).  Same numbers on every machine,
# independent of numpy/torch generator versions, so fixtures only need to hold OUTPUTS.
# --------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def portable_uniform(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n float64 uniforms in (0,1), a pure function of (seed, stream, index)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed], dtype=np.uint64) * np.uint64(0x632BE59BD9B4E019)
                           + np.uint64(stream) * np.uint64(0xD1342543DE82EF95))
        idx = np.arange(n, dtype=np.uint64)
        bits = _splitmix64(idx * np.uint64(0x2545F4914F6CDD1D) + base)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def portable_normal(seed: int, shape, stream: int = 0, mean: float = 0.0, std: float = 1.0) -> torch.Tensor:
    n = int(np.prod(shape))
    m = (n + 1) // 2
    u1 = portable_uniform(seed, m, stream * 2 + 1)
    u2 = portable_uniform(seed, m, stream * 2 + 2)
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])[:n]
    return torch.from_numpy((z * std + mean).astype(np.float32)).reshape(tuple(shape))


def portable_labels(seed: int, n: int, n_classes: int, stream: int = 0) -> torch.Tensor:
    u = portable_uniform(seed, n, 1000 + stream)
    return torch.from_numpy(np.minimum((u * n_classes).astype(np.int64), n_classes - 1))


# --------------------------------------------------------------------------------------
# ResNet-18 parameter structure (models/backbone.py:55-140, 211-213)
# --------------------------------------------------------------------------------------
def resnet18_conv_specs(modality: str) -> List[Tuple[str, int, int, int, int, int]]:
    """(state_dict prefix, cin, cout, k, stride, pad) in module order (backbone.py:78-95,118-140)."""
    cin0 = 1 if modality == "audio" else 3
    specs = [("conv1", cin0, 64, 7, 2, 3)]
    inpl = 64
    for li, planes in enumerate([64, 128, 256, 512], start=1):
        for bi in range(2):
            stride = 2 if (li > 1 and bi == 0) else 1
            specs.append((f"layer{li}.{bi}.conv1", inpl, planes, 3, stride, 1))
            specs.append((f"layer{li}.{bi}.conv2", planes, planes, 3, 1, 1))
            if bi == 0 and (stride != 1 or inpl != planes):
                specs.append((f"layer{li}.{bi}.downsample.0", inpl, planes, 1, stride, 0))
            inpl = planes
    return specs


def bn_name_for_conv(conv_name: str) -> str:
    if conv_name == "conv1":
        return "bn1"
    if conv_name.endswith("downsample.0"):
        return conv_name[:-1] + "1"
    return conv_name.replace("conv", "bn")


def make_resnet18_params(modality: str, seed: int) -> Dict[str, torch.Tensor]:
    """Portable random ResNet-18 state (reference state_dict keys, OIHW).

    Distributions follow utils/utils.py:106-114 (weight_init: kaiming-normal fan_out conv,
    BN weight 1 / bias 0), but BN affine params are jittered so that parity tests exercise
    gamma/beta (a constant 1/0 would hide scale/shift bugs)."""
    p: Dict[str, torch.Tensor] = {}
    for si, (name, cin, cout, k, _s, _p) in enumerate(resnet18_conv_specs(modality)):
        std = math.sqrt(2.0 / (cout * k * k))
        p[name + ".weight"] = portable_normal(seed, (cout, cin, k, k), stream=10 + 4 * si, std=std)
        bn = bn_name_for_conv(name)
        p[bn + ".weight"] = portable_normal(seed, (cout,), stream=11 + 4 * si, mean=1.0, std=0.05)
        p[bn + ".bias"] = portable_normal(seed, (cout,), stream=12 + 4 * si, mean=0.0, std=0.05)
        p[bn + ".running_mean"] = torch.zeros(cout)
        p[bn + ".running_var"] = torch.ones(cout)
        p[bn + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return p


def make_head_params(d: int, c: int, seed: int) -> Dict[str, torch.Tensor]:
    std = math.sqrt(2.0 / (d + c))  # xavier normal, utils/utils.py:107-109
    return {"weight": portable_normal(seed, (c, d), stream=900, std=std),
            "bias": portable_normal(seed, (c,), stream=901, std=0.01)}


# --------------------------------------------------------------------------------------
# primitive ops, explicit forward/backward
# --------------------------------------------------------------------------------------
def conv2d_fwd(x, w, stride, pad):
    return F.conv2d(x, w, None, stride, pad)


def conv2d_dgrad(dy, w, x_shape, stride, pad):
    return torch.nn.grad.conv2d_input(list(x_shape), w, dy, stride=stride, padding=pad)


def conv2d_wgrad(x, dy, w_shape, stride, pad):
    return torch.nn.grad.conv2d_weight(x, list(w_shape), dy, stride=stride, padding=pad)


def bn_train_fwd(x, gamma, beta, running_mean=None, running_var=None,
                 momentum=BN_MOMENTUM, eps=BN_EPS):
    """nn.BatchNorm2d training forward (backbone.py:29,32,86,128): biased batch variance for
    normalisation, unbiased for the running estimate.  Returns y, mean, invstd."""
    n = x.numel() // x.shape[1]
    xd = x.double()
    mean = xd.mean(dim=(0, 2, 3))
    var = ((xd - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
    invstd = (1.0 / torch.sqrt(var + eps))
    y = ((x - mean.float()[None, :, None, None]) * invstd.float()[None, :, None, None]
         * gamma[None, :, None, None] + beta[None, :, None, None])
    if running_mean is not None:
        running_mean.mul_(1 - momentum).add_(momentum * mean.float())
        running_var.mul_(1 - momentum).add_(momentum * (var * n / max(n - 1, 1)).float())
    return y, mean.float(), invstd.float()


def bn_train_bwd(dy, x, gamma, mean, invstd):
    n = x.numel() // x.shape[1]
    xhat = (x - mean[None, :, None, None]) * invstd[None, :, None, None]
    dbeta = dy.double().sum(dim=(0, 2, 3))
    dgamma = (dy.double() * xhat.double()).sum(dim=(0, 2, 3))
    dx = (gamma * invstd)[None, :, None, None] * (
        dy - (dbeta / n).float()[None, :, None, None] - xhat * (dgamma / n).float()[None, :, None, None])
    return dx, dgamma.float(), dbeta.float()


def maxpool3x3s2_fwd(x):
    return F.max_pool2d(x, 3, 2, 1, return_indices=True)


def maxpool3x3s2_bwd(dy, idx, x_shape):
    n, c, h, w = x_shape
    dx = torch.zeros(n, c, h * w, dtype=dy.dtype)
    dx.scatter_add_(2, idx.reshape(n, c, -1), dy.reshape(n, c, -1))
    return dx.reshape(n, c, h, w)


# --------------------------------------------------------------------------------------
# ResNet-18 trunk forward / backward  (backbone.py:142-160, 36-52)
# --------------------------------------------------------------------------------------
def resnet18_fwd(p: Dict[str, torch.Tensor], x: torch.Tensor, modality: str,
                 update_running: bool = True, taps: Optional[dict] = None):
    """Returns (feature map (N,512,h,w), cache).  `x` is (B,1,H,W) audio or (B,3,T,H,W) visual."""
    cache: dict = {"modality": modality}
    if modality == "visual":
        B, C, T, H, W = x.shape
        x = x.permute(0, 2, 1, 3, 4).contiguous().view(B * T, C, H, W)  # backbone.py:144-147
    cache["x0"] = x

    def bn(name, t):
        rm = p[name + ".running_mean"] if update_running else None
        rv = p[name + ".running_var"] if update_running else None
        y, mean, invstd = bn_train_fwd(t, p[name + ".weight"], p[name + ".bias"], rm, rv)
        if update_running:
            p[name + ".num_batches_tracked"] += 1
        cache[name] = (t, mean, invstd)
        return y

    y = conv2d_fwd(x, p["conv1.weight"], 2, 3)
    y = torch.relu(bn("bn1", y))
    cache["stem_relu"] = y
    y, idx = maxpool3x3s2_fwd(y)
    cache["pool_idx"] = idx
    if taps is not None:
        taps["stem"] = y
    inpl = 64
    for li, planes in enumerate([64, 128, 256, 512], start=1):
        for bi in range(2):
            pre = f"layer{li}.{bi}"
            stride = 2 if (li > 1 and bi == 0) else 1
            xin = y
            o = conv2d_fwd(xin, p[pre + ".conv1.weight"], stride, 1)
            o = torch.relu(bn(pre + ".bn1", o))
            cache[pre + ".a1"] = o
            o = conv2d_fwd(o, p[pre + ".conv2.weight"], 1, 1)
            o = bn(pre + ".bn2", o)
            if bi == 0 and (stride != 1 or inpl != planes):
                idn = conv2d_fwd(xin, p[pre + ".downsample.0.weight"], stride, 0)
                idn = bn(pre + ".downsample.1", idn)
            else:
                idn = xin
            y = torch.relu(o + idn)
            cache[pre + ".in"] = xin
            cache[pre + ".out"] = y
            inpl = planes
            if taps is not None:
                taps[pre] = y
    return y, cache


def resnet18_bwd(p: Dict[str, torch.Tensor], cache: dict, dout: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Explicit backward of resnet18_fwd; returns grads keyed like the state_dict."""
    g: Dict[str, torch.Tensor] = {}

    def bn_b(name, dy):
        t, mean, invstd = cache[name]
        dx, dgamma, dbeta = bn_train_bwd(dy, t, p[name + ".weight"], mean, invstd)
        g[name + ".weight"] = dgamma
        g[name + ".bias"] = dbeta
        return dx

    d = dout
    inpls = {1: 64, 2: 64, 3: 128, 4: 256}
    for li, planes in reversed(list(enumerate([64, 128, 256, 512], start=1))):
        for bi in (1, 0):
            pre = f"layer{li}.{bi}"
            stride = 2 if (li > 1 and bi == 0) else 1
            xin = cache[pre + ".in"]
            out = cache[pre + ".out"]
            a1 = cache[pre + ".a1"]
            d = d * (out > 0)                                   # relu after the residual add
            d_idn = d
            dy2 = bn_b(pre + ".bn2", d)
            w2 = p[pre + ".conv2.weight"]
            g[pre + ".conv2.weight"] = conv2d_wgrad(a1, dy2, w2.shape, 1, 1)
            da1 = conv2d_dgrad(dy2, w2, a1.shape, 1, 1) * (a1 > 0)
            dy1 = bn_b(pre + ".bn1", da1)
            w1 = p[pre + ".conv1.weight"]
            g[pre + ".conv1.weight"] = conv2d_wgrad(xin, dy1, w1.shape, stride, 1)
            dx = conv2d_dgrad(dy1, w1, xin.shape, stride, 1)
            if bi == 0 and (stride != 1 or inpls[li] != planes):
                dyd = bn_b(pre + ".downsample.1", d_idn)
                wd = p[pre + ".downsample.0.weight"]
                g[pre + ".downsample.0.weight"] = conv2d_wgrad(xin, dyd, wd.shape, stride, 0)
                dx = dx + conv2d_dgrad(dyd, wd, xin.shape, stride, 0)
            else:
                dx = dx + d_idn
            d = dx
    d = maxpool3x3s2_bwd(d, cache["pool_idx"], cache["stem_relu"].shape)
    d = d * (cache["stem_relu"] > 0)
    dy = bn_b("bn1", d)
    g["conv1.weight"] = conv2d_wgrad(cache["x0"], dy, p["conv1.weight"].shape, 2, 3)
    return g


# --------------------------------------------------------------------------------------
# AVClassifier pooling (basic_model.py:56-65)
# --------------------------------------------------------------------------------------
def av_pool_fwd(fa: torch.Tensor, fv: torch.Tensor, batch: int):
    a = fa.mean(dim=(2, 3))                                      # adaptive_avg_pool2d(a,1)+flatten
    _, C, H, W = fv.shape
    v = fv.view(batch, -1, C, H, W).permute(0, 2, 1, 3, 4)        # (B,C,T,H,W)
    v = v.mean(dim=(2, 3, 4))                                     # adaptive_avg_pool3d(v,1)+flatten
    return a, v


def audio_pool_bwd(da: torch.Tensor, fa_shape):
    n, c, h, w = fa_shape
    return (da / (h * w))[:, :, None, None].expand(n, c, h, w).contiguous()


def visual_pool_bwd(dv: torch.Tensor, fv_shape, batch: int):
    nt, c, h, w = fv_shape
    t = nt // batch
    return (dv / (t * h * w))[:, None, :, None, None].expand(batch, t, c, h, w).reshape(nt, c, h, w).contiguous()


# --------------------------------------------------------------------------------------
# shared head + cross entropy (fusion_modules.py:16-19; main.py:130,432-435)
# --------------------------------------------------------------------------------------
def head_ce_fwd_bwd(X: torch.Tensor, W: torch.Tensor, b: torch.Tensor, labels: torch.Tensor):
    """logits = X W^T + b; mean CE; returns logits, loss, dW, db, dX."""
    B = X.shape[0]
    logits = X @ W.t() + b
    m = logits.max(dim=1, keepdim=True).values
    e = torch.exp(logits - m)
    s = e.sum(dim=1, keepdim=True)
    logp = logits - m - torch.log(s)
    loss = -logp[torch.arange(B), labels].mean()
    dl = e / s
    dl[torch.arange(B), labels] -= 1.0
    dl = dl / B
    return logits, loss, dl.t() @ X, dl.sum(dim=0), dl @ W


# --------------------------------------------------------------------------------------
# GSPlugin.before_update (utils/utils.py:24-41), literal, incl. quirks Q1/Q2/Q5 of SURVEY.md
# --------------------------------------------------------------------------------------
def gs_alpha(batch_index: int, len_dataloader: int) -> float:
    lamda = batch_index / len_dataloader + 1        # utils/utils.py:26
    return 1.0 * 0.1 ** lamda                       # utils/utils.py:27


def gs_before_update(Pl: torch.Tensor, X: torch.Tensor, G: torch.Tensor, batch_index: int,
                     len_dataloader: int, train_exp_counter: int, mode: str = "as_intended"):
    """Returns (Pl_new, G_new).  mode 'as_published' reproduces the name-mismatch no-op (Q1);
    'as_intended' executes utils/utils.py:34-41 (element-wise DxD denominator, Frobenius renorm)."""
    if mode == "as_published" or train_exp_counter == 0:
        return Pl, G
    alpha = gs_alpha(batch_index, len_dataloader)
    r = X.mean(dim=0, keepdim=True)                              # (1,D)          :34
    k = Pl @ r.t()                                               # (D,1)          :35
    Pl = Pl - (k @ k.t()) / (alpha + k @ r)                      # element-wise   :36
    Pl = Pl / torch.linalg.norm(Pl)                              # Frobenius      :38-40
    return Pl, G @ Pl.t()                                        #                :41


# --------------------------------------------------------------------------------------
# SGD with momentum + weight decay (main.py:749)
# --------------------------------------------------------------------------------------
def sgd_step(p: torch.Tensor, g: Optional[torch.Tensor], buf: Optional[torch.Tensor],
             lr: float, momentum: float = 0.9, wd: float = 1e-4):
    """torch.optim.SGD semantics (dampening 0, no nesterov).  g None -> zero gradient
    (legacy torch-1.8.1 zero_grad behaviour).  Returns (p_new, buf_new)."""
    g = torch.zeros_like(p) if g is None else g
    d = g + wd * p
    buf = d.clone() if buf is None else momentum * buf + d
    return p - lr * buf, buf


# --------------------------------------------------------------------------------------
# the MLA alternating step (main.py:419-476) over a functional state
# --------------------------------------------------------------------------------------
class MLAState:
    """Everything the step mutates: encoder params + BN buffers, head, momentum, Pl, exp_count."""

    def __init__(self, audio: Dict[str, torch.Tensor], visual: Dict[str, torch.Tensor],
                 head: Dict[str, torch.Tensor], d: int = 512):
        self.audio, self.visual, self.head = audio, visual, head
        self.mom: Dict[str, Dict[str, Optional[torch.Tensor]]] = {"audio": {}, "visual": {}, "head": {}}
        self.Pl = torch.eye(d)
        self.exp_count = 0

    def clone(self) -> "MLAState":
        s = MLAState({k: v.clone() for k, v in self.audio.items()},
                     {k: v.clone() for k, v in self.visual.items()},
                     {k: v.clone() for k, v in self.head.items()}, self.Pl.shape[0])
        s.mom = {m: {k: (None if v is None else v.clone()) for k, v in d.items()} for m, d in self.mom.items()}
        s.Pl = self.Pl.clone()
        s.exp_count = self.exp_count
        return s


def _is_param(k: str) -> bool:
    return not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))


def _sgd_group(params: Dict[str, torch.Tensor], grads: Optional[Dict[str, torch.Tensor]],
               mom: Dict[str, Optional[torch.Tensor]], lr, momentum, wd):
    for k in list(params.keys()):
        if not _is_param(k):
            continue
        g = None if grads is None else grads[k]
        params[k], mom[k] = sgd_step(params[k], g, mom.get(k), lr, momentum, wd)


def mla_step(st: MLAState, spec: torch.Tensor, image: torch.Tensor, label: torch.Tensor,
             batch_index: int, len_dataloader: int, lr: float = 1e-3, momentum: float = 0.9,
             wd: float = 1e-4, gs_mode: str = "as_intended", legacy_zero_grad: bool = False,
             av_alpha: float = 0.55) -> dict:
    """One pass of main.py:419-476 (ResNet/CREMA-D branch).  spec (B,H,W) or (B,1,H,W); image (B,3,T,H,W)."""
    out: dict = {}
    B = spec.shape[0]
    if spec.dim() == 3:
        spec = spec.unsqueeze(1)                                  # main.py:431
    fa, ca = resnet18_fwd(st.audio, spec.float(), "audio")
    fv, cv = resnet18_fwd(st.visual, image.float(), "visual")
    a, v = av_pool_fwd(fa, fv, B)
    out["a"], out["v"] = a, v

    # ---- audio phase (main.py:432-442)
    W, b = st.head["weight"], st.head["bias"]
    logits, loss, dW, db, dX = head_ce_fwd_bwd(a, W, b, label)
    out["out_a"], out["loss_a"], out["head_grad_a_raw"] = logits, loss, dW.clone()
    ga = resnet18_bwd(st.audio, ca, audio_pool_bwd(dX, fa.shape))
    st.Pl, dW = gs_before_update(st.Pl, a, dW, batch_index, len_dataloader, st.exp_count, gs_mode)
    out["head_grad_a"], out["grads_audio"] = dW, ga
    _sgd_group(st.audio, ga, st.mom["audio"], lr, momentum, wd)
    _sgd_group(st.head, {"weight": dW, "bias": db}, st.mom["head"], lr, momentum, wd)
    st.exp_count += 1

    # ---- visual phase (main.py:444-454); uses the head already updated by the audio step (Q7)
    W, b = st.head["weight"], st.head["bias"]
    logits, loss_v, dW, db, dX = head_ce_fwd_bwd(v, W, b, label)
    out["out_v"], out["loss_v"], out["head_grad_v_raw"] = logits, loss_v, dW.clone()
    gv = resnet18_bwd(st.visual, cv, visual_pool_bwd(dX, fv.shape, B))
    st.Pl, dW = gs_before_update(st.Pl, v, dW, batch_index, len_dataloader, st.exp_count, gs_mode)
    out["head_grad_v"], out["grads_visual"] = dW, gv
    if legacy_zero_grad:                                          # torch 1.8.1: zeroed, not None (Q6)
        _sgd_group(st.audio, None, st.mom["audio"], lr, momentum, wd)
    _sgd_group(st.visual, gv, st.mom["visual"], lr, momentum, wd)
    _sgd_group(st.head, {"weight": dW, "bias": db}, st.mom["head"], lr, momentum, wd)
    st.exp_count += 1

    out["loss"] = loss * av_alpha + loss_v * (1 - av_alpha)      # main.py:472 (Q8)
    return out


# ======================================================================================
# M3AE modality encoder: models/m3ae.py:48-179 (blocks), 181-223 (sin-cos),
# 300-370 (MaskedMultimodalAutoencoder.forward_representation); models/basic_model.py:182-200
# (M3AEClassifier.forward).  Functional restatement on torch-CPU ops; gradients by autograd (the reference
# uses autograd too, and there are no ReLU/max-pool decisions to flip here).  DropPath == identity (Q10).
# Parameter dict uses the reference's state_dict keys and layouts (nn.Linear.weight is (out, in)).
# ======================================================================================
def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:            # m3ae.py:181-194
    omega = np.arange(embed_dim // 2, dtype=np.float32)
    omega /= embed_dim / 2.
    omega = 1. / 10000 ** omega
    out = np.einsum('m,d->md', pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_pos_embed_1d(embed_dim: int, length: int) -> torch.Tensor:    # m3ae.py:197-203
    return torch.from_numpy(sincos_1d(embed_dim, np.arange(length, dtype=np.float32)).astype(np.float32))


def sincos_pos_embed_2d(embed_dim: int, length: int) -> torch.Tensor:    # m3ae.py:206-223 ("w goes first")
    gs = int(length ** 0.5)
    assert gs * gs == length
    grid = np.stack(np.meshgrid(np.arange(gs, dtype=np.float32), np.arange(gs, dtype=np.float32)), axis=0)
    grid = grid.reshape([2, 1, gs, gs])
    emb = np.concatenate([sincos_1d(embed_dim // 2, grid[0]), sincos_1d(embed_dim // 2, grid[1])], axis=1)
    return torch.from_numpy(emb.astype(np.float32))


def m3ae_param_names(depth: int) -> List[str]:
    names = ["text_embedding.weight", "image_embedding.weight", "image_embedding.bias",
             "encoder_image_type_embedding", "encoder_text_type_embedding", "cls_token"]
    for i in range(depth):
        b = f"encoder.blocks.{i}."
        names += [b + "layer_norm1.weight", b + "layer_norm1.bias", b + "attention.qkv_linear.weight",
                  b + "attention.qkv_linear.bias", b + "attention.fc.weight", b + "attention.fc.bias",
                  b + "layer_norm2.weight", b + "layer_norm2.bias", b + "transformer_mlp.fc1.weight",
                  b + "transformer_mlp.fc1.bias", b + "transformer_mlp.fc2.weight", b + "transformer_mlp.fc2.bias"]
    return names + ["encoder.layer_norm.weight", "encoder.layer_norm.bias"]


def make_m3ae_params(seed: int, depth: int = 12, emb: int = 768, vocab: int = 30522, patch_dim: int = 768) -> Dict[str, torch.Tensor]:
    shapes = {"text_embedding.weight": (vocab, emb), "image_embedding.weight": (emb, patch_dim), "image_embedding.bias": (emb,),
              "encoder_image_type_embedding": (1, 1, emb), "encoder_text_type_embedding": (1, 1, emb), "cls_token": (1, 1, emb)}
    p: Dict[str, torch.Tensor] = {}
    for si, name in enumerate(m3ae_param_names(depth)):
        if name in shapes:
            shp = shapes[name]
        elif name.endswith("qkv_linear.weight"):
            shp = (3 * emb, emb)
        elif name.endswith("qkv_linear.bias"):
            shp = (3 * emb,)
        elif name.endswith("fc1.weight"):
            shp = (4 * emb, emb)
        elif name.endswith("fc1.bias"):
            shp = (4 * emb,)
        elif name.endswith("fc2.weight"):
            shp = (emb, 4 * emb)
        elif name.endswith("weight") and "layer_norm" not in name:
            shp = (emb, emb)
        else:
            shp = (emb,)
        if "layer_norm" in name and name.endswith("weight"):
            p[name] = portable_normal(seed, shp, stream=2000 + si, mean=1.0, std=0.05)
        elif name == "text_embedding.weight":
            p[name] = portable_normal(seed, shp, stream=2000 + si, std=1.0)              # m3ae.py:307 normal_(0,1)
        elif len(shp) == 2:
            p[name] = portable_normal(seed, shp, stream=2000 + si, std=(2.0 / (shp[0] + shp[1])) ** 0.5)
        else:
            p[name] = portable_normal(seed, shp, stream=2000 + si, std=0.02)
    return p


def patchify(image: torch.Tensor, p: int = 16) -> torch.Tensor:
    """einops 'b c (h p1) (w p2) -> b (h w) (c p1 p2)' (basic_model.py:184-186)."""
    B, C, H, W = image.shape
    x = image.reshape(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, (H // p) * (W // p), C * p * p)


def m3ae_block(p, pre: str, x: torch.Tensor, padding_mask: Optional[torch.Tensor], heads: int) -> torch.Tensor:
    B, n, D = x.shape
    h = F.layer_norm(x, (D,), p[pre + "layer_norm1.weight"], p[pre + "layer_norm1.bias"])          # m3ae.py:146
    qkv = F.linear(h, p[pre + "attention.qkv_linear.weight"], p[pre + "attention.qkv_linear.bias"])
    qkv = qkv.view(B, n, 3, heads, D // heads).permute(2, 0, 3, 1, 4)                                # :104-106
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = torch.matmul(q, k.transpose(-2, -1)) * ((D // heads) ** -0.5)                             # :109
    if padding_mask is not None:
        pm = padding_mask[:, None, None, :].expand(att.shape)
        att = torch.where(pm > 0, torch.tensor(-1e7), att)                                          # :111-117
    att = F.softmax(att, dim=-1)
    o = torch.matmul(att, v).permute(0, 2, 1, 3).reshape(B, n, D)                                   # :121-122
    x = x + F.linear(o, p[pre + "attention.fc.weight"], p[pre + "attention.fc.bias"])               # :123, 149
    h = F.layer_norm(x, (D,), p[pre + "layer_norm2.weight"], p[pre + "layer_norm2.bias"])
    h = F.gelu(F.linear(h, p[pre + "transformer_mlp.fc1.weight"], p[pre + "transformer_mlp.fc1.bias"]))   # :76-77
    return x + F.linear(h, p[pre + "transformer_mlp.fc2.weight"], p[pre + "transformer_mlp.fc2.bias"])     # :79, 154


def m3ae_forward_representation(p, image_patches=None, text=None, text_padding_mask=None, heads: int = 12, depth: Optional[int] = None):
    """m3ae.py:342-370, device-free.  Exactly one of image_patches (B,L,768) / text (B,L) int64 is given
    (MLA runs each modality through its own M3AE).  Returns the (B, 1+L, D) representation."""
    D = p["cls_token"].shape[-1]
    if depth is None:
        depth = sum(1 for k in p if k.endswith("layer_norm1.weight"))
    B = image_patches.shape[0] if image_patches is not None else text.shape[0]
    xs = [p["cls_token"].expand(B, 1, D)]
    pms = [torch.zeros((B, 1))]
    if image_patches is not None:
        L = image_patches.shape[1]
        xs.append(F.linear(image_patches, p["image_embedding.weight"], p["image_embedding.bias"])
                  + sincos_pos_embed_2d(D, L)[None] + p["encoder_image_type_embedding"])
        pms.append(torch.zeros((B, L)))
    if text is not None:
        L = text.shape[1]
        xs.append(F.embedding(text, p["text_embedding.weight"]) + sincos_pos_embed_1d(D, L)[None]
                  + p["encoder_text_type_embedding"])
        pms.append(text_padding_mask)
    x = torch.cat(xs, dim=1)
    pm = torch.cat(pms, dim=1)
    for i in range(depth):
        x = m3ae_block(p, f"encoder.blocks.{i}.", x, pm, heads)
    return F.layer_norm(x, (D,), p["encoder.layer_norm.weight"], p["encoder.layer_norm.bias"])      # m3ae.py:176


def m3ae_feature(p, image=None, token=None, padding_mask=None, heads: int = 12):
    """M3AEClassifier.forward for one modality (basic_model.py:182-200): token mean over ALL 1+L positions."""
    if image is not None:
        return m3ae_forward_representation(p, image_patches=patchify(image), heads=heads).mean(dim=1)
    return m3ae_forward_representation(p, text=token.squeeze(1), text_padding_mask=padding_mask.squeeze(1), heads=heads).mean(dim=1)


class M3AEState:
    def __init__(self, text: Dict[str, torch.Tensor], image: Dict[str, torch.Tensor], head: Dict[str, torch.Tensor]):
        self.text, self.image, self.head = text, image, head
        self.mom: Dict[str, Dict[str, torch.Tensor]] = {"text": {}, "image": {}, "head": {}}
        self.Pl = torch.eye(head["weight"].shape[1])
        self.exp_count = 0


def mla_step_m3ae(st: M3AEState, token, padding_mask, image, label, batch_index: int, len_dataloader: int,
                  lr: float = 1e-3, momentum: float = 0.9, wd: float = 1e-4, gs_mode: str = "as_intended",
                  heads: int = 12) -> dict:
    """main.py:419-476 with args.lorb == 'm3ae' (a = text modality via mae_a, v = image via mae_v; main.py:426)."""
    out: dict = {}

    def phase(name, params, feat_fn):
        leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        W = st.head["weight"].clone().requires_grad_(True)
        b = st.head["bias"].clone().requires_grad_(True)
        feat = feat_fn(leaves)
        logits = feat @ W.t() + b
        loss = F.cross_entropy(logits, label)
        keys = list(leaves)
        grads = torch.autograd.grad(loss, [leaves[k] for k in keys] + [W, b], allow_unused=True)
        g = {k: gv for k, gv in zip(keys, grads[:-2]) if gv is not None}     # unused params: grad None -> SGD skips them
        dW, db = grads[-2], grads[-1]
        out["feat_" + name], out["out_" + name], out["loss_" + name] = feat.detach(), logits.detach(), loss.detach()
        out[f"head_grad_{name}_raw"] = dW.clone()
        st.Pl, dWp = gs_before_update(st.Pl, feat.detach(), dW, batch_index, len_dataloader, st.exp_count, gs_mode)
        out[f"head_grad_{name}"], out["grads_" + name] = dWp, g
        mom = st.mom["text" if name == "a" else "image"]
        for k, gv in g.items():
            params[k], mom[k] = sgd_step(params[k], gv, mom.get(k), lr, momentum, wd)
        for k, gv in (("weight", dWp), ("bias", db)):
            st.head[k], st.mom["head"][k] = sgd_step(st.head[k], gv, st.mom["head"].get(k), lr, momentum, wd)
        st.exp_count += 1

    phase("a", st.text, lambda p: m3ae_feature(p, token=token, padding_mask=padding_mask, heads=heads))
    phase("v", st.image, lambda p: m3ae_feature(p, image=image, heads=heads))
    out["loss"] = out["loss_a"] * 0.55 + out["loss_v"] * 0.45
    return out


# ======================================================================================
# CAV-MAE audio branch: models/cav_mae.py:69-84 (PatchEmbed), 86-113 (Block with
# modality-specific norms), 116-186 (CAVMAEFT.__init__), 337-351 (forward_feat(.., 'a')); models/basic_model.py
# 252-275 (Modal3Classifier.forward).  PARITY UNPINNED: the Attention / Mlp arithmetic lives in timm==0.4.5
# (requirements.txt:55; cav_mae.py:16, 93-94, 101), which is neither vendored nor installed, and the reference has
# no test or fixture for this path.  Restated from timm 0.4.5's published definitions: Attention = qkv Linear
# (bias) -> (B,N,3,H,hd) -> softmax(q k^T * hd^-0.5) v -> proj Linear; Mlp = fc1 -> GELU(erf) -> fc2.
# ======================================================================================
def cavmae_audio_param_names(depth: int = 12) -> List[str]:
    names = ["patch_embed_a.proj.weight", "patch_embed_a.proj.bias", "modality_a", "pos_embed_a"]
    for i in range(depth):
        shared = i >= depth - 1
        pre = f"blocks_u.{i - (depth - 1)}." if shared else f"blocks_a.{i}."
        n1, n2 = ("norm1_a", "norm2_a") if shared else ("norm1", "norm2")
        names += [pre + n1 + ".weight", pre + n1 + ".bias", pre + "attn.qkv.weight", pre + "attn.qkv.bias", pre + "attn.proj.weight",
                  pre + "attn.proj.bias", pre + n2 + ".weight", pre + n2 + ".bias", pre + "mlp.fc1.weight", pre + "mlp.fc1.bias",
                  pre + "mlp.fc2.weight", pre + "mlp.fc2.bias"]
    return names + ["norm_a.weight", "norm_a.bias"]


def make_cavmae_audio_params(seed: int, depth: int = 12, emb: int = 768, tokens: int = 512) -> Dict[str, torch.Tensor]:
    p: Dict[str, torch.Tensor] = {}
    for si, name in enumerate(cavmae_audio_param_names(depth)):
        if name == "patch_embed_a.proj.weight":
            shp, std = (emb, 1, 16, 16), (2.0 / (emb + 256)) ** 0.5
        elif name == "pos_embed_a":
            shp, std = (1, tokens, emb), 0.5
        elif name == "modality_a":
            shp, std = (1, 1, emb), 0.02
        elif name.endswith("qkv.weight"):
            shp, std = (3 * emb, emb), (2.0 / (4 * emb)) ** 0.5
        elif name.endswith("qkv.bias"):
            shp, std = (3 * emb,), 0.02
        elif name.endswith("fc1.weight"):
            shp, std = (4 * emb, emb), (2.0 / (5 * emb)) ** 0.5
        elif name.endswith("fc1.bias"):
            shp, std = (4 * emb,), 0.02
        elif name.endswith("fc2.weight"):
            shp, std = (emb, 4 * emb), (2.0 / (5 * emb)) ** 0.5
        elif name.endswith("proj.weight"):
            shp, std = (emb, emb), (1.0 / emb) ** 0.5
        else:
            shp, std = (emb,), 0.02
        mean = 1.0 if ("norm" in name and name.endswith("weight")) else 0.0
        p[name] = portable_normal(seed, shp, stream=3000 + si, mean=mean, std=0.05 if mean else std)
    return p


def cavmae_audio_feature(p, audio: torch.Tensor, heads: int = 12) -> torch.Tensor:
    """CAVMAEFT.forward_feat(audio, None, 'a') (cav_mae.py:337-351) + .mean(dim=1) (basic_model.py:267)."""
    a = audio.unsqueeze(1).transpose(2, 3)                                              # (B,1,128,1024)   :339-340
    a = F.conv2d(a, p["patch_embed_a.proj.weight"], p["patch_embed_a.proj.bias"], stride=16).flatten(2).transpose(1, 2)   # :82
    x = a + p["pos_embed_a"] + p["modality_a"]                                          # :342-343
    B, n, D = x.shape
    depth = sum(1 for k in p if k.endswith("attn.qkv.weight"))
    for i in range(depth):
        shared = i >= depth - 1
        pre = f"blocks_u.{i - (depth - 1)}." if shared else f"blocks_a.{i}."
        n1, n2 = ("norm1_a", "norm2_a") if shared else ("norm1", "norm2")              # cav_mae.py:103-108, 345-348
        h = F.layer_norm(x, (D,), p[pre + n1 + ".weight"], p[pre + n1 + ".bias"])
        qkv = F.linear(h, p[pre + "attn.qkv.weight"], p[pre + "attn.qkv.bias"]).reshape(B, n, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
        att = F.softmax((qkv[0] @ qkv[1].transpose(-2, -1)) * (D // heads) ** -0.5, dim=-1)
        o = (att @ qkv[2]).transpose(1, 2).reshape(B, n, D)
        x = x + F.linear(o, p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"])
        h = F.layer_norm(x, (D,), p[pre + n2 + ".weight"], p[pre + n2 + ".bias"])
        h = F.gelu(F.linear(h, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"]))
        x = x + F.linear(h, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])
    return F.layer_norm(x, (D,), p["norm_a.weight"], p["norm_a.bias"]).mean(dim=1)       # :349; basic_model.py:267


def mla_step_modal3(audio_p, image_p, text_p, head, Pl, exp_count, token, padding_mask, image, spec, label,
                    batch_index: int, len_dataloader: int, lr: float = 1e-3, gs_mode: str = "as_intended"):
    """The FIRST pass of main.py:419-476 with --modal3 (a -> v -> t): returns per-modality features / logits / losses /
    raw + projected head gradients / parameter gradients and the updated head, Pl, exp_count.  Encoders take their first
    SGD step (momentum buffer = gradient); the shared head is stepped three times and its momentum carries over."""
    out = {}
    hmom: Dict[str, Optional[torch.Tensor]] = {"weight": None, "bias": None}
    feats = [("a", audio_p, lambda q: cavmae_audio_feature(q, spec)),
             ("v", image_p, lambda q: m3ae_feature(q, image=image)),
             ("t", text_p, lambda q: m3ae_feature(q, token=token, padding_mask=padding_mask))]
    # joint forward happens before any update (Q7); the encoders are independent, so per-phase forward is identical
    for name, params, fn in feats:
        leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        W, b = head["weight"].clone().requires_grad_(True), head["bias"].clone().requires_grad_(True)
        feat = fn(leaves)
        logits = feat @ W.t() + b
        loss = F.cross_entropy(logits, label)
        keys = list(leaves)
        grads = torch.autograd.grad(loss, [leaves[k] for k in keys] + [W, b], allow_unused=True)
        g = {k: gv for k, gv in zip(keys, grads[:-2]) if gv is not None}
        dW, db = grads[-2], grads[-1]
        out["feat_" + name], out["out_" + name], out["loss_" + name] = feat.detach(), logits.detach(), loss.detach()
        out[f"head_grad_{name}_raw"] = dW.clone()
        Pl, dWp = gs_before_update(Pl, feat.detach(), dW, batch_index, len_dataloader, exp_count, gs_mode)
        out[f"head_grad_{name}"], out["grads_" + name] = dWp, g
        for k, gv in g.items():
            params[k] = sgd_step(params[k], gv, None, lr)[0]
        head = dict(head)
        for k, gv in (("weight", dWp), ("bias", db)):
            head[k], hmom[k] = sgd_step(head[k], gv, hmom[k], lr)
        exp_count += 1
    out["head"], out["Pl"], out["exp_count"] = head, Pl, exp_count
    return out


# ======================================================================================
# Evaluation path: main.py:486-679 `valid`, gs_flag branch 622-651; entropy gating 65-106.
# ======================================================================================
def resnet18_eval_fwd(p: Dict[str, torch.Tensor], x: torch.Tensor, modality: str) -> torch.Tensor:
    """ResNet.forward (backbone.py:142-160) with the BatchNorm layers in eval mode (running statistics)."""
    if modality == "visual":
        B, C, T, H, W = x.shape
        x = x.permute(0, 2, 1, 3, 4).contiguous().view(B * T, C, H, W)

    def bn(name, t):
        return F.batch_norm(t, p[name + ".running_mean"], p[name + ".running_var"], p[name + ".weight"], p[name + ".bias"],
                            False, BN_MOMENTUM, BN_EPS)

    y = F.max_pool2d(torch.relu(bn("bn1", conv2d_fwd(x, p["conv1.weight"], 2, 3))), 3, 2, 1)
    inpl = 64
    for li, planes in enumerate([64, 128, 256, 512], start=1):
        for bi in range(2):
            pre, stride = f"layer{li}.{bi}", (2 if (li > 1 and bi == 0) else 1)
            o = torch.relu(bn(pre + ".bn1", conv2d_fwd(y, p[pre + ".conv1.weight"], stride, 1)))
            o = bn(pre + ".bn2", conv2d_fwd(o, p[pre + ".conv2.weight"], 1, 1))
            idn = y
            if bi == 0 and (stride != 1 or inpl != planes):
                idn = bn(pre + ".downsample.1", conv2d_fwd(y, p[pre + ".downsample.0.weight"], stride, 0))
            y = torch.relu(o + idn)
            inpl = planes
    return y


def calculate_entropy(output: torch.Tensor) -> torch.Tensor:                 # main.py:65-70 (softmax over dim 0 = the batch, Q9)
    prob = F.softmax(output, dim=0)
    return -torch.sum(prob * torch.log(prob))


def gating_weights(outs: List[torch.Tensor]) -> List[torch.Tensor]:         # main.py:72-106
    ent = [calculate_entropy(o) for o in outs]
    mx = max(ent)
    w = [torch.exp(mx - e) for e in ent]
    s = sum(w)
    return [x / s for x in w]


def valid_batch(outs: List[torch.Tensor], label: torch.Tensor, n_classes: int, dynamic: bool, alphas: List[float]):
    """main.py:640-676 for one batch: returns (weights, counts) with counts[k][c]: k = 0 num, 1 fused, 2.. per modality."""
    w = gating_weights(outs) if dynamic else [torch.tensor(a) for a in alphas]
    fused = sum(wi * o for wi, o in zip(w, outs))
    counts = torch.zeros(2 + len(outs), n_classes, dtype=torch.int64)
    preds = [F.softmax(fused, dim=1).argmax(1)] + [F.softmax(o, dim=1).argmax(1) for o in outs]
    for i in range(label.shape[0]):
        counts[0, label[i]] += 1
        for k, pr in enumerate(preds):
            if pr[i] == label[i]:
                counts[1 + k, label[i]] += 1
    return [float(x) for x in w], counts


# ---------------------------------------------------------------------------------------------------------------------
# OGM / OGM-GE gradient modulation.  main.py:312-410, non --gs_flag branch.
# ---------------------------------------------------------------------------------------------------------------------
def ogm_coefficients(outs: List[torch.Tensor], label: torch.Tensor, alpha: float):
    """main.py:373-384 (two modalities: outs = [out_a, out_v]) / main.py:314-337 (three: [out_a, out_v, out_t]).
    Returns (coeffs, scores, ratios) as python floats / 0-dim tensors exactly as the reference forms them: softmax over
    dim 1 (main.py:131), python sum() over the rows in order, tanh / relu modules on 0-dim tensors."""
    softmax, relu, tanh = torch.nn.Softmax(dim=1), torch.nn.ReLU(inplace=False), torch.nn.Tanh()
    scores = [sum([softmax(o)[i][label[i]] for i in range(o.size(0))]) for o in outs]
    one = torch.tensor(1.0)
    if len(outs) == 2:
        score_a, score_v = scores
        ratio_v = score_v / score_a                                   # :376
        ratio_a = 1 / ratio_v                                         # :377
        if ratio_v > 1:                                               # :379-384
            coeffs = [one, 1 - tanh(alpha * relu(ratio_v))]
        else:
            coeffs = [1 - tanh(alpha * relu(ratio_a)), one]
        ratios = [ratio_a, ratio_v]
    else:
        score_a, score_v, score_t = scores
        ratio_v = score_v / (score_a + score_t)                       # :319-321
        ratio_a = score_a / (score_v + score_t)
        ratio_t = score_t / (score_v + score_a)
        if ratio_v > 1:                                               # :323-337
            coeffs = [one, 1 - tanh(alpha * relu(ratio_v)), one]
        elif ratio_t > 1:
            coeffs = [one, one, 1 - tanh(alpha * relu(ratio_t))]
        else:
            coeffs = [1 - tanh(alpha * relu(ratio_a)), one, one]
        ratios = [ratio_a, ratio_v, ratio_t]
    return [torch.as_tensor(c, dtype=torch.float32) for c in coeffs], scores, ratios


def ogm_modulate(grads: Dict[str, torch.Tensor], coeff: torch.Tensor, mode: str, generator: Optional[torch.Generator] = None):
    """main.py:394-408 for one encoder: every 4-D gradient is scaled (OGM) or scaled and perturbed with N(0, std + 1e-8)
    of the unscaled gradient (OGM_GE); other gradients are untouched.  Returns the new dict."""
    out = {}
    for k, g in grads.items():
        if g.dim() != 4:
            out[k] = g
        elif mode == "OGM_GE":
            out[k] = g * coeff + torch.zeros_like(g).normal_(0, g.std().item() + 1e-8, generator=generator)   # :397-400
        else:
            out[k] = g * coeff                                                                                 # :401-402
    return out
