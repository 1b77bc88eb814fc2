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

from . import ops
from ._lib import MLAHipError
from .encoder import ResNet18Encoder

N_CLASSES = {"CREMAD": 6, "MVSA": 3, "Food101": 101, "IEMOCAP": 4}   # main.py:491-507


class SharedHead:
    """fc_out = nn.Linear(in_features, out_features): weight (C,D) and bias (C) in one flat buffer."""

    def __init__(self, in_features: int, out_features: int, device="cuda", seed: Optional[int] = None):
        self.in_features, self.out_features = in_features, out_features
        self.device = torch.device(device)
        n = out_features * in_features
        self.numel = n + out_features
        self.flat = torch.zeros(self.numel, device=self.device, dtype=torch.float32)
        self.grad = torch.zeros(self.numel, device=self.device, dtype=torch.float32)
        self.weight = self.flat[:n].view(out_features, in_features)
        self.bias = self.flat[n:]
        self.weight_grad = self.grad[:n].view(out_features, in_features)
        self.bias_grad = self.grad[n:]
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed) if seed is not None else gen.seed()
        std = math.sqrt(2.0 / (in_features + out_features))           # xavier_normal_, utils/utils.py:107-109
        self.weight.copy_(torch.randn((out_features, in_features), generator=gen) * std)
        self._ws: dict = {}

    def _buffers(self, B: int, slot: str = "") -> dict:
        if (B, slot) not in self._ws:
            f32 = dict(device=self.device, dtype=torch.float32)
            self._ws[(B, slot)] = {"logits": torch.empty((B, self.out_features), **f32), "loss": torch.empty(1, **f32),
                           "dX": torch.empty((B, self.in_features), **f32),
                           "ws": torch.empty(ops.head_ws_elems(B, self.out_features), **f32)}
        return self._ws[(B, slot)]

    def forward_backward(self, X: torch.Tensor, labels: torch.Tensor, inv_batch: Optional[float] = None, slot: str = ""):
        """logits, CE loss and all gradients (main.py:432-435).  Gradients land in
        self.weight_grad / self.bias_grad; returns (logits, loss[1], dX).  inv_batch = 1/global batch."""
        B = X.shape[0]
        buf = self._buffers(B, slot)
        ops.head_ce_fwd_bwd(X, self.weight, self.bias, labels, buf["logits"], buf["loss"], self.weight_grad,
                            self.bias_grad, buf["dX"], buf["ws"], (1.0 / B) if inv_batch is None else inv_batch)
        return buf["logits"], buf["loss"], buf["dX"]

    def logits(self, X: torch.Tensor, slot: str = "eval") -> torch.Tensor:
        """out = fc_out(x) without loss/gradients (evaluation, main.py:636-639)."""
        buf = self._buffers(X.shape[0], slot)
        ops.head_logits(X, self.weight, self.bias, buf["logits"])
        return buf["logits"]

    __call__ = logits

    def state_dict(self, prefix: str = "") -> Dict[str, torch.Tensor]:
        return {prefix + "weight": self.weight.clone(), prefix + "bias": self.bias.clone()}

    def load_state_dict(self, sd, prefix: str = "") -> None:
        self.weight.copy_(sd[prefix + "weight"].to(self.device, torch.float32))
        self.bias.copy_(sd[prefix + "bias"].to(self.device, torch.float32))


class ConcatFusion:
    """models/fusion_modules.py:16-24; under --gs_flag only `fc_out` is touched."""

    def __init__(self, input_dim: int = 512, output_dim: int = 100, device="cuda", seed: Optional[int] = None):
        self.fc_out = SharedHead(input_dim, output_dim, device, seed)


class AVClassifier:
    """Two ResNet-18 encoders + shared head (models/basic_model.py:14-77), --gs_flag configuration."""

    def __init__(self, args, device="cuda", seed: Optional[int] = None, conv_math: Optional[str] = None):
        """conv_math: "f32" (exact fp32 MFMA, default) or "split" (exact bf16 operand split, fp32-equivalent; see
        encoder.py); default from $MLA_CONV_MATH."""
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
        self.module = self                                                  # `model.module.` paths (DataParallel, main.py:432)
        self.training = True
        self._feat: Dict[int, dict] = {}

    def mla_encoders(self):
        """(phase tag, optimiser group name, encoder) in the order main.py:432-454 alternates over them."""
        return [("a", "audio", self.audio_net), ("v", "visual", self.visual_net)]

    def train(self, mode: bool = True):
        self.training = bool(mode)
        self.audio_net.train(mode)
        self.visual_net.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def _feat_buffers(self, B: int) -> dict:
        if B not in self._feat:
            f32 = dict(device=self.device, dtype=torch.float32)
            self._feat[B] = {"a": torch.empty((B, 512), **f32), "v": torch.empty((B, 512), **f32)}
        return self._feat[B]

    def forward_audio(self, audio: torch.Tensor) -> torch.Tensor:
        B = audio.shape[0]
        fa = self.audio_net.forward(audio)
        buf = self._feat_buffers(B)
        n, h, w, c = fa.shape
        self.audio_net._pa = h * w                                           # pooled pixels per sample (for the backward)
        ops.avgpool_fwd(fa, buf["a"], B, h * w, c)                           # adaptive_avg_pool2d + flatten (basic_model.py:61,64)
        return buf["a"]

    def forward_visual(self, visual: torch.Tensor) -> torch.Tensor:
        B = visual.shape[0]
        fv = self.visual_net.forward(visual)
        buf = self._feat_buffers(B)
        nt, hv, wv, cv = fv.shape
        self.visual_net._pa = (nt // B) * hv * wv
        ops.avgpool_fwd(fv, buf["v"], B, self.visual_net._pa, cv)            # regroup T + adaptive_avg_pool3d + flatten (:56-65)
        return buf["v"]

    def forward(self, audio: torch.Tensor, visual: torch.Tensor):
        """a, v = model(spec.unsqueeze(1).float(), image.float())  (main.py:431; basic_model.py:52-77)."""
        if visual.shape[0] != audio.shape[0]:
            raise MLAHipError("audio/visual batch mismatch")
        return self.forward_audio(audio), self.forward_visual(visual)

    def forward_split(self, audio: torch.Tensor, visual: torch.Tensor):
        """Per-encoder forward closures in alternation order, so the trainer may run later encoders' forwards on a
        side stream: no encoder forward depends on the head or on another encoder."""
        if visual.shape[0] != audio.shape[0]:
            raise MLAHipError("audio/visual batch mismatch")
        return [lambda: self.forward_audio(audio), lambda: self.forward_visual(visual)]

    __call__ = forward

    # ---- reference-compatible (de)serialisation ------------------------------------------------
    def state_dict(self, prefix: str = "") -> Dict[str, torch.Tensor]:
        sd = {}
        sd.update(self.fusion_module.fc_out.state_dict(prefix + "fusion_module.fc_out."))
        sd.update(self.audio_net.state_dict(prefix + "audio_net."))
        sd.update(self.visual_net.state_dict(prefix + "visual_net."))
        return sd

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        if any(k.startswith("module.") for k in sd):                         # main.py:724-727 strips it too
            sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
        self.audio_net.load_state_dict(sd, "audio_net.", strict)
        self.visual_net.load_state_dict(sd, "visual_net.", strict)
        if "fusion_module.fc_out.weight" in sd:
            self.fusion_module.fc_out.load_state_dict(sd, "fusion_module.fc_out.")
        elif strict:
            raise KeyError("missing key fusion_module.fc_out.weight")
