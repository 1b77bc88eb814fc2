"""Drop-in model objects for the --gs_flag path of the reference.

  AVClassifier   models/basic_model.py:14-77   (attribute paths audio_net / visual_net / fusion_module.fc_out)
  ConcatFusion   models/fusion_modules.py:16-24 (only fc_out is used by MLA; main.py:432, 444)
  SharedHead     the nn.Linear(D, C) behind fc_out

state_dict()/load_state_dict() speak the reference's keys and layouts (OIHW conv weights,
`audio_net.conv1.weight`, `fusion_module.fc_out.weight`, optional `module.` prefix,
main.py:724-727, 921); internally everything is flat HWIO buffers (see encoder.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from .streams import distinct_streams
import torch.nn as nn

from . import ops
from ._lib import MLAHipError
from .autograd import EncoderFeature, HeadLinear, make_anchor
from .encoder import ResNet18Encoder
from .module import FlatModule

N_CLASSES = {"CREMAD": 6, "MVSA": 3, "Food101": 101, "IEMOCAP": 4}   # main.py:491-507


class SharedHead(FlatModule, nn.Linear):
    """fc_out = nn.Linear(in_features, out_features): weight (C,D) and bias (C) in one flat buffer.

    Protocol face: `fc_out(x)` is a differentiable Linear on the HIP kernels (autograd.HeadLinear), `.weight` / `.bias`
    are nn.Parameters viewing the flat buffer, after `loss.backward()` `.weight.grad` / `.bias.grad` alias the flat
    gradient buffer, `named_parameters()` yields 'weight', 'bias' (what utils/utils.py:30 iterates)."""

    def __init__(self, in_features: int, out_features: int, device="cuda", seed: Optional[int] = None):
        FlatModule.__init__(self)
        self.in_features, self.out_features = in_features, out_features
        self.device = torch.device(device)
        n = out_features * in_features
        self.numel = n + out_features
        self.flat = torch.zeros(self.numel, device=self.device, dtype=torch.float32)
        self.grad = torch.zeros(self.numel, device=self.device, dtype=torch.float32)
        self.weight_grad = self.grad[:n].view(out_features, in_features)
        self.bias_grad = self.grad[n:]
        self.weight = self._param_view(self.flat[:n].view(out_features, in_features), self.weight_grad, lambda t: t, "weight")
        self.bias = self._param_view(self.flat[n:], self.bias_grad, lambda t: t, "bias")
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed) if seed is not None else gen.seed()
        std = math.sqrt(2.0 / (in_features + out_features))           # xavier_normal_, utils/utils.py:107-109
        with torch.no_grad():
            self.weight.copy_(torch.randn((out_features, in_features), generator=gen) * std)
        self._ws: dict = {}
        self._anchor = make_anchor(self.device)

    def _bufs(self, B: int, slot: str = "") -> dict:
        if (B, slot) not in self._ws:
            f32 = dict(device=self.device, dtype=torch.float32)
            self._ws[(B, slot)] = {"logits": torch.empty((B, self.out_features), **f32), "loss": torch.empty(1, **f32),
                           "dX": torch.empty((B, self.in_features), **f32),
                           "ws": torch.empty(ops.head_ws_elems(B, self.out_features), **f32)}
        return self._ws[(B, slot)]

    def forward_backward(self, X: torch.Tensor, labels: torch.Tensor, inv_batch: Optional[float] = None, slot: str = ""):
        """Fused trainer path: logits, CE loss and all gradients (main.py:432-435).  Gradients land in
        self.weight_grad / self.bias_grad; returns (logits, loss[1], dX).  inv_batch = 1/global batch."""
        B = X.shape[0]
        buf = self._bufs(B, slot)
        ops.head_ce_fwd_bwd(X, self.weight.detach(), self.bias.detach(), labels, buf["logits"], buf["loss"], self.weight_grad,
                            self.bias_grad, buf["dX"], buf["ws"], (1.0 / B) if inv_batch is None else inv_batch)
        return buf["logits"], buf["loss"], buf["dX"]

    def logits(self, X: torch.Tensor, slot: str = "eval") -> torch.Tensor:
        """out = fc_out(x) without loss/gradients into a reused buffer (Evaluator, main.py:636-639)."""
        buf = self._bufs(X.shape[0], slot)
        ops.head_logits(X, self.weight.detach(), self.bias.detach(), buf["logits"])
        return buf["logits"]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """out = fc_out(x)  (main.py:432, 444, 456, 636-639): differentiable when x carries history."""
        if x.dim() != 2 or x.shape[1] != self.in_features:
            raise MLAHipError(f"fc_out expects (B, {self.in_features}), got {tuple(x.shape)}")
        return HeadLinear.apply(self._anchor, self, x)


class ConcatFusion(nn.Module):
    """models/fusion_modules.py:16-24; under --gs_flag only `fc_out` is touched."""

    def __init__(self, input_dim: int = 512, output_dim: int = 100, device="cuda", seed: Optional[int] = None):
        super().__init__()
        self.fc_out = SharedHead(input_dim, output_dim, device, seed)

    def forward(self, x, y):
        raise NotImplementedError("mla_hip implements the --gs_flag (MLA) path only: fc_out is applied per modality")


class _Classifier(nn.Module):
    """Shared protocol behaviour of AVClassifier / M3AEClassifier / Modal3Classifier."""
    side_streams = True

    @property
    def module(self):
        """`model.module.fusion_module.fc_out` (main.py:432) also works on the bare model (no DataParallel wrapper)."""
        return self

    def _side_stream(self) -> torch.cuda.Stream:
        """One weight-gradient side stream for all encoders of this model on the protocol path, on a hardware queue of its own
        (streams.py): the encoders' backwards run one after the other there, so one stream serves them all."""
        if getattr(self, "_wgrad_side", None) is None:
            self._wgrad_side = distinct_streams(1, self.device)[0]
        return self._wgrad_side

    def _feature(self, enc, run, B: int, D: int) -> torch.Tensor:
        if torch.is_grad_enabled() and self.training:
            if not hasattr(enc, "_anchor"):
                enc._anchor = make_anchor(self.device)
                # protocol path: the weight-gradient GEMMs of a backward run on a side stream beside the dgrad -> BN-backward
                # chain (joined before the gradients are published), like in MLATrainer's pipeline
                if self.side_streams and getattr(enc, "wgrad_stream", 0) is None and self.device.type == "cuda":
                    enc.wgrad_stream = self._side_stream()
            return EncoderFeature.apply(enc._anchor, enc, run, B, D)
        out = torch.empty((B, D), device=self.device, dtype=torch.float32)
        run(out)
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Accepts DataParallel-style `module.`-prefixed keys as well (main.py:724 strips them by hand)."""
        if any(k.startswith("module.") for k in state_dict):
            state_dict = {k[len("module."):] if k.startswith("module.") else k: v for k, v in state_dict.items()}
        return super().load_state_dict(state_dict, strict=strict, **kw)


class AVClassifier(_Classifier):
    """Two ResNet-18 encoders + shared head (models/basic_model.py:14-77), --gs_flag configuration."""

    def __init__(self, args, device="cuda", seed: Optional[int] = None, conv_math: Optional[str] = None):
        """conv_math: "f32" (exact fp32 MFMA, default) or "split" (exact bf16 operand split, fp32-equivalent; see
        encoder.py); default from $MLA_CONV_MATH."""
        super().__init__()
        fusion = getattr(args, "fusion_method", "concat")
        dataset = getattr(args, "dataset", "CREMAD")
        if dataset != "CREMAD":                                             # basic_model.py:19-26
            raise NotImplementedError("Incorrect dataset name {}".format(dataset))
        n_classes = N_CLASSES[dataset]
        if fusion != "concat":                                              # basic_model.py:28-40
            raise NotImplementedError("Incorrect fusion method: {}!".format(fusion))
        if not getattr(args, "gs_flag", False):
            raise NotImplementedError("mla_hip implements the --gs_flag (MLA) path only")
        self.args = args
        self.device = torch.device(device)
        s = (lambda k: None if seed is None else seed + k)
        self.fusion_module = ConcatFusion(512, n_classes, device, s(2))    # basic_model.py:31-32
        self.audio_net = ResNet18Encoder("audio", device, s(0), conv_math)     # basic_model.py:42
        self.visual_net = ResNet18Encoder("visual", device, s(1), conv_math)   # basic_model.py:43
        self._feat: Dict[int, dict] = {}

    def mla_encoders(self):
        """(phase tag, optimiser group name, encoder) in the order main.py:432-454 alternates over them."""
        return [("a", "audio", self.audio_net), ("v", "visual", self.visual_net)]

    def _feat_buffers(self, B: int) -> dict:
        if B not in self._feat:
            f32 = dict(device=self.device, dtype=torch.float32)
            self._feat[B] = {"a": torch.empty((B, 512), **f32), "v": torch.empty((B, 512), **f32)}
        return self._feat[B]

    def forward_audio(self, audio: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        B = audio.shape[0]
        fa = self.audio_net.forward(audio)
        if out is None:
            out = self._feat_buffers(B)["a"]
        n, h, w, c = fa.shape
        self.audio_net._pa = h * w                                           # pooled pixels per sample (for the backward)
        ops.avgpool_fwd(fa, out, B, h * w, c)                                # adaptive_avg_pool2d + flatten (basic_model.py:61,64)
        return out

    def forward_visual(self, visual: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        B = visual.shape[0]
        fv = self.visual_net.forward(visual)
        if out is None:
            out = self._feat_buffers(B)["v"]
        nt, hv, wv, cv = fv.shape
        self.visual_net._pa = (nt // B) * hv * wv
        ops.avgpool_fwd(fv, out, B, self.visual_net._pa, cv)                 # regroup T + adaptive_avg_pool3d + flatten (:56-65)
        return out

    def forward_raw(self, audio: torch.Tensor, visual: torch.Tensor):
        """Kernel-level joint forward into the reused feature buffers (MLATrainer / Evaluator; no autograd)."""
        if visual.shape[0] != audio.shape[0]:
            raise MLAHipError("audio/visual batch mismatch")
        return self.forward_audio(audio), self.forward_visual(visual)

    def forward(self, audio: torch.Tensor, visual: torch.Tensor):
        """a, v = model(spec.unsqueeze(1).float(), image.float())  (main.py:431; basic_model.py:52-77): fresh (B,512)
        tensors that carry autograd history to their encoder when grad mode is on and the model is training."""
        if visual.shape[0] != audio.shape[0]:
            raise MLAHipError("audio/visual batch mismatch")
        B = audio.shape[0]
        a = self._feature(self.audio_net, lambda out: self.forward_audio(audio, out), B, 512)
        v = self._feature(self.visual_net, lambda out: self.forward_visual(visual, out), B, 512)
        return a, v

    def forward_split(self, audio: torch.Tensor, visual: torch.Tensor):
        """Per-encoder forward closures in alternation order, so the trainer may run later encoders' forwards on a
        side stream: no encoder forward depends on the head or on another encoder."""
        if visual.shape[0] != audio.shape[0]:
            raise MLAHipError("audio/visual batch mismatch")
        return [lambda: self.forward_audio(audio), lambda: self.forward_visual(visual)]
