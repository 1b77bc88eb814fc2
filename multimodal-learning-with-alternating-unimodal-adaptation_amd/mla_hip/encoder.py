"""ResNet-18 modality encoder on the HIP kernels (reference: models/backbone.py:15-52, 55-160, 211-213).

MI355X-first host design, not a module-by-module port:
  * all parameters of an encoder live in ONE flat fp32 buffer (conv weights HWIO, then BN
    gamma/beta), gradients and SGD momentum in two more of the same shape, so the optimiser is a
    single launch and data-parallel gradient exchange is a single RCCL all-reduce over 44.7 MB;
  * activations are pixel-major NHWC so the implicit-GEMM gathers whole channel rows and the
    BatchNorm passes stream float4 with channels on the fast axis;
  * forward and backward are explicit launch plans over a per-shape workspace that is allocated
    once (288 GB of HBM: nothing is recomputed, nothing is allocated in steady state);
  * BatchNorm statistics come out of the conv epilogue; ReLU masks and residual adds ride in the
    dgrad / BN-apply epilogues, so no standalone element-wise kernels exist;
  * conv_math selects the arithmetic of the forward / input-gradient / weight-gradient contractions of the 64..512-channel
    convs (default "split", $MLA_CONV_MATH overrides):
    "f32" = exact fp32 MFMA (v_mfma_f32_32x32x2_f32), "split" = fp32 operands split exactly into three bf16 terms,
    six products on v_mfma_f32_32x32x16_bf16 (fp32 in / out / accumulate, same error against fp64; the conv weights
    are re-split once per forward).  The stem (1 / 3 input channels) follows: persistent split-arithmetic kernels (stem_split.hip)
    under "split", the fp32 MFMA under "f32".
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from ._lib import MLAHipError
from .module import BatchNorm2dHolder, Conv2dHolder, FlatModule


# Shipped default arithmetic of the conv / Linear contractions: "split" = every fp32 operand split exactly into three
# bf16 terms, six bf16 MFMAs per fp32 product, fp32 accumulate -- fp32 in / out, error against fp64 no larger than the fp32
# MFMA's (tests/test_ops_gpu.py::test_conv_split_is_not_reduced_precision), 1.2-1.3x the throughput.  "f32" = v_mfma_f32_32x32x2_f32.
DEFAULT_CONV_MATH = "split"
# measurement switch (same-box A/B): 0 = every BatchNorm backward runs its own reduction pass (the round-1 / early round-2 flow)
FUSE_BN_REDUCE = os.environ.get("MLA_FUSE_BN_REDUCE", "1") != "0"
# measurement switch: 0 = the downsample input gradient is its own four-launch pass over dx (round 2)
DS_DGRAD_FOLD = os.environ.get("MLA_DS_DGRAD_FOLD", "1") != "0"
# conv1 -> bn1 -> relu -> conv2 of a BasicBlock (backbone.py:38-46) without materialising relu(bn1(.)): the three consumers of that
# activation re-form it from conv1's output (bit-identical: ops.conv2d_*_bnin / _bnmask), where the kernels support it (the 64-channel
# blocks of layer1 on the split arithmetic).  MLA_BN_FOLD=0 restores the bn_apply pass.
BN_FOLD = os.environ.get("MLA_BN_FOLD", "1") != "0"


def conv_specs(modality: str) -> List[Tuple[str, int, int, int, int, int]]:
    """(state_dict prefix, cin, cout, k, stride, pad) in reference module order (backbone.py:78-95, 118-140)."""
    if modality not in ("audio", "visual"):
        # same error behaviour as backbone.py:84-85
        raise NotImplementedError("Incorrect modality, should be audio or visual but got {}".format(modality))
    specs = [("conv1", 1 if modality == "audio" else 3, 64, 7, 2, 3)]
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


class ResNet18Encoder(FlatModule):
    """ResNet-18 trunk without avgpool/fc (backbone.py:96-99), training-mode BatchNorm.

    nn.Module face (module.py): `named_parameters()` / `state_dict()` yield the reference's names in the reference's
    registration order (conv1, bn1, layer1.0.conv1, ... backbone.py:78-95, 27-33) with OIHW conv weights that are
    strided views of the flat HWIO buffer; leaves are nn.Conv2d / nn.BatchNorm2d instances for `weight_init`."""

    def __init__(self, modality: str, device="cuda", seed: Optional[int] = None, conv_math: Optional[str] = None):
        super().__init__()
        self.modality = modality
        self.conv_math = conv_math or os.environ.get("MLA_CONV_MATH", DEFAULT_CONV_MATH)
        if self.conv_math not in ("f32", "split"):
            raise MLAHipError(f"conv_math must be 'f32' or 'split', got {self.conv_math!r}")
        self.device = torch.device(device)
        self.specs = conv_specs(modality)
        # the stem follows conv_math too (round 3): "split" -> stem_split.hip's persistent kernels, "f32" -> the exact fp32 MFMA
        # ($MLA_STEM_SPLIT=0: same-box A/B switch back to the fp32 stem under conv_math="split")
        self.stem_split = (self.conv_math == "split" and os.environ.get("MLA_STEM_SPLIT", "1") != "0"
                           and ops.conv2d_stem_supported(self.specs[0][1], self.specs[0][2], self.specs[0][3], self.specs[0][3], self.specs[0][4], self.specs[0][5]))
        # ---- flat layout: name -> (offset, shape); conv weights HWIO
        self.layout: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for name, cin, cout, k, _s, _p in self.specs:
            self.layout[name + ".weight"] = (off, (k, k, cin, cout))
            off += k * k * cin * cout
        for name, _cin, cout, _k, _s, _p in self.specs:
            bn = bn_name_for_conv(name)
            self.layout[bn + ".weight"] = (off, (cout,))
            off += cout
            self.layout[bn + ".bias"] = (off, (cout,))
            off += cout
        self.numel = off
        self.flat = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.grad = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.p = {k: self.flat[o:o + math.prod(s)].view(s) for k, (o, s) in self.layout.items()}
        self.g = {k: self.grad[o:o + math.prod(s)].view(s) for k, (o, s) in self.layout.items()}
        # ---- BN buffers (one flat buffer: running_mean | running_var per BN)
        self.bn_names = [bn_name_for_conv(n) for n, *_ in self.specs]
        self.bn_ch = {bn_name_for_conv(n): cout for n, _ci, cout, *_ in self.specs}
        tot = sum(self.bn_ch.values())
        self.running = torch.zeros(2 * tot, device=self.device, dtype=torch.float32)
        self.rm, self.rv = {}, {}
        o = 0
        for bn in self.bn_names:
            c = self.bn_ch[bn]
            self.rm[bn] = self.running[o:o + c]
            self.rv[bn] = self.running[tot + o:tot + o + c]
            o += c
        self.running[tot:].fill_(1.0)
        self.num_batches_tracked = {bn: 0 for bn in self.bn_names}
        self.training = True
        self._rinv_flat = torch.zeros(tot, device=self.device, dtype=torch.float32)    # 1/sqrt(running_var+eps), eval mode
        self._tot_bn = tot
        self.rinv = {}
        o = 0
        for bn in self.bn_names:
            c = self.bn_ch[bn]
            self.rinv[bn] = self._rinv_flat[o:o + c]
            o += c
        self.grad_ready = False
        self._plan_key = None
        self._ws: dict = {}
        # split-bf16 images of the conv weights (conv_math == "split"): per conv a transposed image for the forward
        # GEMM and a straight one for the input-gradient GEMM, 3 bf16 planes each, in one int16 buffer
        self.wsp: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._wsplit_dirty = True
        if self.conv_math == "split":
            tot16 = sum(2 * 3 * k * k * cin * cout for _n, cin, cout, k, _s, _p in self.specs if cin % 64 == 0)
            self._wsplit_flat = torch.empty(tot16, device=self.device, dtype=torch.int16)
            o16, rows, blocks = 0, [], 0
            for name, cin, cout, k, _s, _p in self.specs:
                if cin % 64 != 0:
                    continue
                n16 = 3 * k * k * cin * cout
                self.wsp[name] = (self._wsplit_flat[o16:o16 + n16], self._wsplit_flat[o16 + n16:o16 + 2 * n16])
                nb = k * k * ((cin + 31) // 32) * ((cout + 31) // 32)
                for transposed, off in ((1, o16), (0, o16 + n16)):     # descriptor rows of mla_conv2d_wsplit_batch
                    rows.append([self.layout[name + ".weight"][0], off, k * k, cin, cout, transposed, blocks, 0])
                    blocks += nb
                o16 += 2 * n16
            self._wsplit_desc = torch.tensor(rows, dtype=torch.int32, device=self.device)
            self._wsplit_blocks = blocks
        # Optional second HIP stream for the weight-gradient GEMMs: they are off the dgrad -> BN-backward critical
        # chain, so (with one dy buffer per conv: 288 GB of HBM) they run beside it and fill its kernel tails.
        self.wgrad_stream: Optional[torch.cuda.Stream] = None
        # Stream that carries this encoder's training chain (set by MLATrainer): anything that touches the encoder from
        # another stream first waits for it (stream-ordered semantics for forward / state_dict / eval without a device sync).
        self.tail_stream: Optional[torch.cuda.Stream] = None
        # ---- reference-named parameter / buffer tree (views; nothing is copied)
        for name, _cin, _cout, _k, s_, p_ in self.specs:
            w = self._param_view(self.p[name + ".weight"], self.g[name + ".weight"], lambda t: t.permute(3, 2, 0, 1), name + ".weight")
            parent, leaf = self._descend(self, name)
            parent.add_module(leaf, Conv2dHolder(w, s_, p_))
            bn = bn_name_for_conv(name)
            bw = self._param_view(self.p[bn + ".weight"], self.g[bn + ".weight"], lambda t: t, bn + ".weight")
            bb = self._param_view(self.p[bn + ".bias"], self.g[bn + ".bias"], lambda t: t, bn + ".bias")
            parent, leaf = self._descend(self, bn)
            parent.add_module(leaf, BatchNorm2dHolder(bw, bb, self.rm[bn], self.rv[bn]))
        self.register_state_dict_pre_hook(ResNet18Encoder._before_state_dict)
        self.register_load_state_dict_post_hook(ResNet18Encoder._after_load_state_dict)
        self.reset_parameters(seed)

    def _await_tail(self) -> None:
        ts = self.tail_stream
        if ts is not None:
            cur = torch.cuda.current_stream()
            if cur != ts:
                cur.wait_stream(ts)

    # ------------------------------------------------------------------------------------------
    # parameters / state_dict (reference keys and OIHW layout at the boundary)
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def reset_parameters(self, seed: Optional[int] = None) -> None:
        """utils/utils.py:106-114 weight_init (overrides backbone.py:101-106): kaiming-normal fan_out, BN 1/0."""
        gen = torch.Generator(device="cpu")
        if seed is not None:
            gen.manual_seed(seed)
        else:
            gen.seed()
        for name, cin, cout, k, _s, _p in self.specs:
            std = math.sqrt(2.0 / (cout * k * k))
            w = torch.randn((k, k, cin, cout), generator=gen) * std
            self.p[name + ".weight"].copy_(w)
            bn = bn_name_for_conv(name)
            self.p[bn + ".weight"].fill_(1.0)
            self.p[bn + ".bias"].zero_()

    def _bn_module(self, bn: str):
        m = self
        for part in bn.split("."):
            m = m._modules[part]
        return m

    @staticmethod
    def _before_state_dict(self, prefix, keep_vars) -> None:
        """state_dict() hook: stream-order after the encoder's training chain; materialise the BN call counters."""
        self._await_tail()
        for bn in self.bn_names:
            self._bn_module(bn).num_batches_tracked.fill_(self.num_batches_tracked[bn])

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        self._await_tail()                                   # the copies below must not race the training chain
        self._wsplit_dirty = True
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    @staticmethod
    def _after_load_state_dict(self, incompatible_keys) -> None:
        for bn in self.bn_names:
            self.num_batches_tracked[bn] = int(self._bn_module(bn).num_batches_tracked)

    def grads_as_reference(self) -> Dict[str, torch.Tensor]:
        """Gradients keyed/laid out like the reference's named_parameters() (OIHW)."""
        self._await_tail()
        out = {}
        for k in self.layout:
            t = self.g[k]
            out[k] = t.permute(3, 2, 0, 1).contiguous() if t.dim() == 4 else t.clone()
        return out

    # ------------------------------------------------------------------------------------------
    # workspace plan
    # ------------------------------------------------------------------------------------------
    def _plan(self, N: int, H: int, W: int) -> dict:
        key = (N, H, W)
        if self._plan_key == key:
            return self._ws
        dev = self.device
        ws: dict = {"shapes": {}}
        f32 = dict(device=dev, dtype=torch.float32)
        cin0 = self.specs[0][1]
        # forward activation shapes
        h1, w1 = ops.conv_out(H, 7, 2, 3), ops.conv_out(W, 7, 2, 3)
        hp, wp = ops.conv_out(h1, 3, 2, 1), ops.conv_out(w1, 3, 2, 1)
        ws["x0_shape"] = (N, H, W, cin0)
        ws["y_stem"] = torch.empty((N, h1, w1, 64), **f32)
        ws["p0"] = torch.empty((N, hp, wp, 64), **f32)
        ws["pool_idx"] = torch.empty((N, hp, wp, 64), device=dev, dtype=torch.uint8)
        max_act = N * hp * wp * 64          # the stem's ReLU output / its gradient are never materialised (fused kernels)
        max_partial = max(ops.conv2d_fwd_partial_elems(N, H, W, cin0, 64, 7, 7, 2, 3), ops.conv2d_stem_fwd_partial_elems())
        max_wgrad = max(ops.conv2d_wgrad_ws_bytes(N, H, W, cin0, 64, 7, 7, 2, 3), ops.conv2d_stem_wgrad_split_ws_bytes(cin0))
        max_bnws = ops.bn_bwd_ws_elems(N * h1 * w1, 64)
        max_w = 0
        ch, cw, inpl = hp, wp, 64
        blocks = []
        for li, planes in enumerate([64, 128, 256, 512], start=1):
            for bi in range(2):
                pre = f"layer{li}.{bi}"
                stride = 2 if (li > 1 and bi == 0) else 1
                oh, ow = ops.conv_out(ch, 3, stride, 1), ops.conv_out(cw, 3, stride, 1)
                has_ds = bi == 0 and (stride != 1 or inpl != planes)
                blk = {"pre": pre, "stride": stride, "cin": inpl, "cout": planes, "in_hw": (ch, cw), "out_hw": (oh, ow),
                       "ds": has_ds}
                for nm in ("y1", "a1", "y2", "out"):
                    blk[nm] = torch.empty((N, oh, ow, planes), **f32)
                blk["fold"] = BN_FOLD and self.conv_math == "split" and ops.conv2d_bnfold_supported(N, oh, ow, planes, planes, 3, 3, 1, 1)
                if has_ds:
                    blk["yd"] = torch.empty((N, oh, ow, planes), **f32)
                blocks.append(blk)
                max_act = max(max_act, N * oh * ow * planes, N * ch * cw * inpl)
                for (ci, co, k, s, p, hh, ww) in ((inpl, planes, 3, stride, 1, ch, cw), (planes, planes, 3, 1, 1, oh, ow)) + \
                        (((inpl, planes, 1, stride, 0, ch, cw),) if has_ds else ()):
                    max_partial = max(max_partial, ops.conv2d_fwd_partial_elems(N, hh, ww, ci, co, k, k, s, p))
                    max_wgrad = max(max_wgrad, ops.conv2d_wgrad_ws_bytes(N, hh, ww, ci, co, k, k, s, p))
                    if self.conv_math == "split":
                        max_wgrad = max(max_wgrad, ops.conv2d_wgrad_split_ws_bytes(N, hh, ww, ci, co, k, k, s, p))
                    max_w = max(max_w, ci * co * k * k)
                max_bnws = max(max_bnws, ops.bn_bwd_ws_elems(N * oh * ow, planes))
                ch, cw, inpl = oh, ow, planes
        ws["blocks"] = blocks
        ws["feat_hw"] = (ch, cw)
        ws["idn"] = torch.empty(max_act, **f32)                       # bn(downsample) temp
        ws["partial"] = torch.empty(max(max_partial, 1), **f32)
        ws["stats"] = {bn: (torch.empty(self.bn_ch[bn], **f32), torch.empty(self.bn_ch[bn], **f32)) for bn in self.bn_names}
        # backward scratch (allocated lazily on first backward)
        ws["bwd_sizes"] = (max_act, max_wgrad, max_w, max_bnws)
        self._ws, self._plan_key = ws, key
        return ws

    def _bwd_ws(self, ws: dict) -> None:
        if "G" in ws:
            return
        max_act, max_wgrad, max_w, max_bnws = ws["bwd_sizes"]
        f32 = dict(device=self.device, dtype=torch.float32)
        ws["G"] = [torch.empty(max_act, **f32) for _ in range(4)]
        # one dy buffer per conv output (never reused inside a backward, so the side-stream wgrads need no extra fences)
        ws["DY"] = {"conv1": torch.empty_like(ws["y_stem"])}
        for blk in ws["blocks"]:
            ws["DY"][blk["pre"] + ".conv1"] = torch.empty_like(blk["y1"])
            ws["DY"][blk["pre"] + ".conv2"] = torch.empty_like(blk["y2"])
            if blk["ds"]:
                ws["DY"][blk["pre"] + ".downsample.0"] = torch.empty_like(blk["yd"])
        ws["wgrad_ws"] = torch.empty((max_wgrad + 3) // 4, **f32)
        ws["wt_ws"] = torch.empty(max_w, **f32)
        ws["bn_ws"] = torch.empty(max_bnws, **f32)
        # partial sums of the BatchNorm-backward reductions formed in input-gradient epilogues: block-below bn2, its downsample, bn1
        n_bnp = max(ops.conv2d_dgrad_bn_partial_elems(b["out"].shape[0], b["out"].shape[1], b["out"].shape[2], b["cout"])
                    for b in ws["blocks"])
        ws["bnp"] = [torch.empty(n_bnp, **f32) for _ in range(3)]

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def _refresh_wsplit(self, st) -> None:
        """Re-split the conv weights (they change with every optimizer step; in eval mode only when marked dirty)."""
        ops.conv2d_wsplit_batch(self.flat, self._wsplit_flat, self._wsplit_desc, self._wsplit_blocks, stream=st)
        self._wsplit_dirty = False

    def train(self, mode: bool = True):
        """nn.Module.train/eval semantics for the BatchNorm layers: eval uses the running statistics."""
        super().train(mode)
        self._await_tail()
        self._wsplit_dirty = True
        if not self.training:
            ops.bn_invstd(self.running[self._tot_bn:], self._rinv_flat)       # one launch for all 20 BN layers
        return self

    def _fold(self, blk) -> bool:
        """True when relu(bn1(y1)) of this block is never materialised (training mode; see BN_FOLD)."""
        return bool(self.training and blk.get("fold") and FUSE_BN_REDUCE)      # (the fold rides on the fused reduction's read of y1)

    def _bn_in(self, ws, bn):
        mean, invstd = ws["stats"][bn]
        return (mean, invstd, self.p[bn + ".weight"], self.p[bn + ".bias"])

    def _conv_bn(self, ws, st, x, conv_name, stride, pad, y, out, relu, residual=None, bn_in=None):
        """y = conv(x); BN statistics fused in the conv epilogue; out = [relu](bn(y) [+ residual]).  out = None: the consumers of the
        activation re-form it from y (BN_FOLD); bn_in: x is itself such a BatchNorm input and the convolution runs over relu(bn_in(x))."""
        w = self.p[conv_name + ".weight"]
        bn = bn_name_for_conv(conv_name)
        wsp = self.wsp.get(conv_name)
        if not self.training:                                                  # eval: running statistics, nothing saved
            if wsp is not None:
                ops.conv2d_fwd_split(x, wsp[0], w.shape, stride, pad, y=y, stream=st)
            else:
                ops.conv2d_fwd(x, w, stride, pad, y=y, stream=st)
            C = w.shape[3]
            ops.bn_apply(y, self.rm[bn], self.rinv[bn], self.p[bn + ".weight"], self.p[bn + ".bias"], out, y.numel() // C, C,
                         relu, residual=residual, stream=st)
            return
        if bn_in is not None:
            _, tiles = ops.conv2d_fwd_split_bnin(x, wsp[0], w.shape, stride, pad, bn_in, y=y, bn_partial=ws["partial"], stream=st)
        elif wsp is not None:
            _, tiles = ops.conv2d_fwd_split(x, wsp[0], w.shape, stride, pad, y=y, bn_partial=ws["partial"], stream=st)
        else:
            _, tiles = ops.conv2d_fwd(x, w, stride, pad, y=y, bn_partial=ws["partial"], stream=st)
        C = w.shape[3]
        M = y.numel() // C
        mean, invstd = ws["stats"][bn]
        ops.bn_finalize(ws["partial"], tiles, M, C, mean, invstd, self.rm[bn], self.rv[bn], stream=st)
        self.num_batches_tracked[bn] += 1
        if out is None:
            return
        ops.bn_apply(y, mean, invstd, self.p[bn + ".weight"], self.p[bn + ".bias"], out, M, C, relu, residual=residual,
                     stream=st)

    def _stem(self, ws, st) -> None:
        """p0 = maxpool(relu(bn1(conv1(x0)))): BN + ReLU are applied inside the max-pool, the (N,112,112,64)-sized ReLU
        output is never written (backbone.py:149-152)."""
        w = self.p["conv1.weight"]
        x, y = ws["x0"], ws["y_stem"]
        ga, be = self.p["bn1.weight"], self.p["bn1.bias"]
        stem_split = self.stem_split          # persistent split-arithmetic patch-loader kernel (stem_split.hip); else the fp32 MFMA
        if not self.training:
            if stem_split:
                ops.conv2d_stem_fwd_split(x, w, y=y, stream=st)
            else:
                ops.conv2d_fwd(x, w, 2, 3, y=y, stream=st)
            ops.bn_relu_maxpool_fwd(y, self.rm["bn1"], self.rinv["bn1"], ga, be, ws["p0"], ws["pool_idx"], stream=st)
            return
        if stem_split:
            _, tiles = ops.conv2d_stem_fwd_split(x, w, y=y, bn_partial=ws["partial"], stream=st)
        else:
            _, tiles = ops.conv2d_fwd(x, w, 2, 3, y=y, bn_partial=ws["partial"], stream=st)
        mean, invstd = ws["stats"]["bn1"]
        ops.bn_finalize(ws["partial"], tiles, y.numel() // 64, 64, mean, invstd, self.rm["bn1"], self.rv["bn1"], stream=st)
        self.num_batches_tracked["bn1"] += 1
        ops.bn_relu_maxpool_fwd(y, mean, invstd, ga, be, ws["p0"], ws["pool_idx"], stream=st)

    def stem_relu(self, gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None) -> torch.Tensor:
        """relu(bn1(conv1 output)) of the live forward, rebuilt on demand (tests / inspection; the step never forms it).
        gamma / beta: bn1's affine parameters AT THE TIME OF THAT FORWARD if an optimizer step has changed them since."""
        ws = self._ws
        y = ws["y_stem"]
        mean, invstd = ws["stats"]["bn1"] if self.training else (self.rm["bn1"], self.rinv["bn1"])
        ga = self.p["bn1.weight"] if gamma is None else gamma.to(y.device, torch.float32).contiguous()
        be = self.p["bn1.bias"] if beta is None else beta.to(y.device, torch.float32).contiguous()
        out = torch.empty_like(y)
        ops.bn_apply(y, mean, invstd, ga, be, out, y.numel() // 64, 64, True)
        return out

    def block_a1(self, blk: dict, gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None) -> torch.Tensor:
        """relu(bn1(conv1 output)) of a BasicBlock of the live forward (tests / inspection).  Blocks whose consumers re-form it from
        conv1's output (BN_FOLD) never store it: rebuilt here by bn_apply -- the same expression, bit for bit.  gamma / beta: bn1's
        affine parameters AT THE TIME OF THAT FORWARD if an optimizer step has changed them since."""
        if not self._fold(blk):
            return blk["a1"]
        bn = blk["pre"] + ".bn1"
        y = blk["y1"]
        C = y.shape[3]
        mean, invstd = self._ws["stats"][bn]
        ga = self.p[bn + ".weight"] if gamma is None else gamma.to(y.device, torch.float32).contiguous()
        be = self.p[bn + ".bias"] if beta is None else beta.to(y.device, torch.float32).contiguous()
        out = torch.empty_like(y)
        ops.bn_apply(y, mean, invstd, ga, be, out, y.numel() // C, C, True)
        return out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: (B,1,H,W) audio or (B,3,T,H,W) visual, fp32, reference layout.  Returns the NHWC feature
        map (N,h,w,512) of layer4 (backbone.py:142-160); activations are kept for backward()."""
        self._await_tail()
        st = ops.cur_stream()
        if self.modality == "visual":
            if x.dim() != 5:
                raise MLAHipError("visual encoder expects (B,C,T,H,W)")
            B, C, T, H, W = x.shape
            N = B * T
        else:
            if x.dim() != 4 or x.shape[1] != 1:
                raise MLAHipError("audio encoder expects (B,1,H,W)")
            N, C, H, W = x.shape
        if C != self.specs[0][1]:
            raise MLAHipError(f"{self.modality} encoder expects {self.specs[0][1]} input channels, got {C}")
        ws = self._plan(N, H, W)
        if self.wsp and (self.training or self._wsplit_dirty):
            self._refresh_wsplit(st)
        x = x.contiguous()
        if self.modality == "visual":
            if "x0" not in ws:
                ws["x0"] = torch.empty((N, H, W, C), device=self.device, dtype=torch.float32)
            ops.video_to_nhwc(x, ws["x0"], stream=st)                      # backbone.py:144-147
        else:
            # C == 1: NCHW == NHWC, but the stem weight gradient reads x0 long after forward() has returned (on the encoder's
            # own streams), so the caller's buffer is copied: it may be a DeviceFeeder slot that is refilled meanwhile
            if "x0" not in ws:
                ws["x0"] = torch.empty((N, H, W, 1), device=self.device, dtype=torch.float32)
            ws["x0"].copy_(x.view(N, H, W, 1))
        self._stem(ws, st)                                                                       # :149-152
        cur = ws["p0"]
        for blk in ws["blocks"]:                                                                 # :154-157
            pre = blk["pre"]
            fold = self._fold(blk)
            self._conv_bn(ws, st, cur, pre + ".conv1", blk["stride"], 1, blk["y1"], None if fold else blk["a1"], relu=True)
            if blk["ds"]:
                idn = ws["idn"][:blk["yd"].numel()].view(blk["yd"].shape)
                self._conv_bn(ws, st, cur, pre + ".downsample.0", blk["stride"], 0, blk["yd"], idn, relu=False)
            else:
                idn = cur
            # out = relu(bn2(conv2(a1)) + identity)   (backbone.py:43-50)
            self._conv_bn(ws, st, blk["y1"] if fold else blk["a1"], pre + ".conv2", 1, 1, blk["y2"], blk["out"], relu=True, residual=idn,
                          bn_in=self._bn_in(ws, pre + ".bn1") if fold else None)
            blk["xin"] = cur
            cur = blk["out"]
        self.grad_ready = False
        return cur

    # ------------------------------------------------------------------------------------------
    # backward (explicit; fills self.grad completely)
    # ------------------------------------------------------------------------------------------
    def _bn_bwd(self, ws, st, bn, dout, x, dx, reduced=None):
        """reduced = (partial sums, tiles) when the producer of dout has already formed the reduction pass (_dgrad bn_next)."""
        C = self.bn_ch[bn]
        M = x.numel() // C
        mean, invstd = ws["stats"][bn]
        if reduced is not None:
            ops.bn_bwd_from_partial(dout, x, mean, invstd, self.p[bn + ".weight"], dx, self.g[bn + ".weight"],
                                    self.g[bn + ".bias"], reduced[0], reduced[1], M, C, stream=st)
            return
        ops.bn_bwd(dout, x, mean, invstd, self.p[bn + ".weight"], dx, self.g[bn + ".weight"], self.g[bn + ".bias"],
                   ws["bn_ws"], M, C, stream=st)

    def backward_from_pooled(self, dfeat: torch.Tensor, P: int) -> None:
        """dfeat: (NB, 512) gradient of the global-average-pooled feature (NB groups of P pixels)."""
        ws = self._ws
        st = ops.cur_stream()
        self._bwd_ws(ws)
        G = ws["G"]
        last = ws["blocks"][-1]
        out = last["out"]
        NB = dfeat.shape[0]
        d = G[0][:out.numel()].view(out.shape)
        ops.avgpool_bwd(dfeat.contiguous(), d, NB, P, out.shape[3], relu_src=out, stream=st)
        self._backward_trunk(ws, st)

    def _wgrad(self, ws, x, dy, name, stride, pad, bn_in=None) -> None:
        """Weight gradient of conv `name`; on the side stream when one is attached (after dy has been produced).  bn_in: x is a BatchNorm
        input and the operand is relu(bn_in(x)) (BN_FOLD)."""
        side = self.wgrad_stream
        wgrad = ops.conv2d_wgrad_split if name in self.wsp else ops.conv2d_wgrad
        if name == "conv1" and self.stem_split:
            wgrad = ops.conv2d_stem_wgrad_split
        if bn_in is not None:
            def wgrad(x_, dy_, dw_, stride_, pad_, ws_):                      # noqa: F811
                return ops.conv2d_wgrad_split_bnin(x_, dy_, dw_, stride_, pad_, ws_, bn_in)
        if side is None:
            wgrad(x, dy, self.g[name + ".weight"], stride, pad, ws["wgrad_ws"])
            return
        ev = torch.cuda.Event()
        ev.record()                                   # dy is complete on the main stream at this point
        side.wait_event(ev)
        with torch.cuda.stream(side):
            wgrad(x, dy, self.g[name + ".weight"], stride, pad, ws["wgrad_ws"])

    def _dgrad(self, ws, st, dy, name, x_shape, stride, pad, dx, residual=None, relu_src=None, bn_next=(), class_mask=0xF,
               residual_mask=0xF):
        """Input gradient of conv `name`.  bn_next: [(bn name, its input y, partial buffer)] -- the BatchNorm layers whose
        backward consumes dx: the epilogue forms their reduction pass; returns {bn name: (partial, tiles)}."""
        wsp = self.wsp.get(name)
        w = self.p[name + ".weight"]
        if not FUSE_BN_REDUCE:
            bn_next = ()
        reqs = [(y,) + tuple(ws["stats"][bn]) + (part,) for bn, y, part in bn_next]
        if wsp is not None:
            r = ops.conv2d_dgrad_split(dy, wsp[1], w.shape, x_shape, stride, pad, dx=dx, residual=residual, relu_src=relu_src,
                                       stream=st, bn_reqs=reqs, class_mask=class_mask, residual_mask=residual_mask)
        else:
            r = ops.conv2d_dgrad(dy, w, x_shape, stride, pad, ws["wt_ws"], dx=dx, residual=residual, relu_src=relu_src, stream=st,
                                 bn_reqs=reqs, class_mask=class_mask, residual_mask=residual_mask)
        return {bn: (part, r[1]) for bn, _, part in bn_next} if bn_next else {}

    def _backward_trunk(self, ws, st) -> None:
        """Expects G[0] = gradient w.r.t. the last block's output, already masked by (out > 0)."""
        G, DY = ws["G"], ws["DY"]
        if self.wgrad_stream is not None:
            self.wgrad_stream.wait_stream(torch.cuda.current_stream())      # wgrad_ws / grads of the previous step are consumed
        # The reduction pass of a BatchNorm backward (sum g, sum g * xhat) is formed by the epilogue of the input-gradient kernel
        # that writes g, for every BatchNorm but the last block's bn2 / downsample (g comes from the average pool) and the stem's:
        # `have` = {bn name: (partial sums, tiles)} handed from the producing launch to the consuming BatchNorm backward.
        blocks = ws["blocks"]
        bnp = ws["bnp"]
        have: dict = {}
        for bi_, blk in reversed(list(enumerate(blocks))):
            pre = blk["pre"]
            oshape = blk["out"].shape
            n_out = blk["out"].numel()
            xin = blk["xin"]
            d = G[0][:n_out].view(oshape)
            dy2 = DY[pre + ".conv2"]
            self._bn_bwd(ws, st, pre + ".bn2", d, blk["y2"], dy2, have.pop(pre + ".bn2", None))
            da1 = G[2][:n_out].view(oshape)
            if self._fold(blk):
                bn_in = self._bn_in(ws, pre + ".bn1")
                self._wgrad(ws, blk["y1"], dy2, pre + ".conv2", 1, 1, bn_in=bn_in)
                _, rt = ops.conv2d_dgrad_split_bnmask(dy2, self.wsp[pre + ".conv2"][1], self.p[pre + ".conv2.weight"].shape, oshape, 1, 1, da1,
                                                      (blk["y1"],) + tuple(ws["stats"][pre + ".bn1"]) + (bnp[2],), bn_in[2], bn_in[3], stream=st)
                got = {pre + ".bn1": (bnp[2], rt)}
            else:
                self._wgrad(ws, blk["a1"], dy2, pre + ".conv2", 1, 1)
                got = self._dgrad(ws, st, dy2, pre + ".conv2", oshape, 1, 1, da1, relu_src=blk["a1"],
                                  bn_next=[(pre + ".bn1", blk["y1"], bnp[2])])
            dy1 = DY[pre + ".conv1"]
            self._bn_bwd(ws, st, pre + ".bn1", da1, blk["y1"], dy1, got.get(pre + ".bn1"))
            self._wgrad(ws, xin, dy1, pre + ".conv1", blk["stride"], 1)
            dx = G[3][:xin.numel()].view(xin.shape)
            mask = xin if bi_ > 0 else None          # block input is a ReLU output except after the max-pool
            nxt = []                                 # BatchNorms of the block below that consume dx
            if bi_ > 0:
                prev = blocks[bi_ - 1]
                nxt.append((prev["pre"] + ".bn2", prev["y2"], bnp[0]))
                if prev["ds"]:
                    nxt.append((prev["pre"] + ".downsample.1", prev["yd"], bnp[1]))
            if blk["ds"]:
                dyd = DY[pre + ".downsample.0"]
                self._bn_bwd(ws, st, pre + ".downsample.1", d, blk["yd"], dyd, have.pop(pre + ".downsample.1", None))
                self._wgrad(ws, xin, dyd, pre + ".downsample.0", blk["stride"], 0)
                if blk["stride"] == 2 and DS_DGRAD_FOLD:
                    # the 1x1 / stride-2 downsample reaches the (even, even) pixels only: it writes that parity class (one launch),
                    # conv1's four class launches then add onto it there, apply the block-input ReLU mask and form the BatchNorm
                    # reductions of the block below -- instead of a second full read-modify-write pass over dx in four launches
                    self._dgrad(ws, st, dyd, pre + ".downsample.0", xin.shape, 2, 0, dx, class_mask=0x1)
                    have = self._dgrad(ws, st, dy1, pre + ".conv1", xin.shape, 2, 1, dx, residual=dx, relu_src=mask, bn_next=nxt,
                                       residual_mask=0x1)
                else:
                    self._dgrad(ws, st, dy1, pre + ".conv1", xin.shape, blk["stride"], 1, dx)
                    have = self._dgrad(ws, st, dyd, pre + ".downsample.0", xin.shape, blk["stride"], 0, dx, residual=dx, relu_src=mask,
                                       bn_next=nxt)
            else:
                have = self._dgrad(ws, st, dy1, pre + ".conv1", xin.shape, blk["stride"], 1, dx, residual=d, relu_src=mask,
                                   bn_next=nxt)
            G[0], G[3] = G[3], G[0]
        # stem: maxpool -> relu -> bn1 -> conv1
        dpool = G[0][:ws["p0"].numel()].view(ws["p0"].shape)
        mean, invstd = ws["stats"]["bn1"]
        ops.bn_bwd_pooled(dpool, ws["pool_idx"], ws["y_stem"], mean, invstd, self.p["bn1.weight"], self.p["bn1.bias"],
                          DY["conv1"], self.g["bn1.weight"], self.g["bn1.bias"], ws["bn_ws"], stream=st)
        self._wgrad(ws, ws["x0"], DY["conv1"], "conv1", 2, 3)
        if self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)      # all gradients complete before SGD / all-reduce
        self.grad_ready = True
