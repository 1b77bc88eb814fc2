"""Transformer modality encoders on the HIP kernels: M3AE text / image (reference: models/m3ae.py:48-179,
300-370; models/basic_model.py:127-200 M3AEClassifier) and the CAV-MAE audio branch (models/cav_mae.py:69-113,
116-186, 337-351; models/basic_model.py:202-275 Modal3Classifier).

One `M3AEEncoder` per modality, like the reference's `mae_a` (text) / `mae_v` (image): pre-LN ViT-B
(emb 768, 12 heads, mlp x4) over [cls] + 256 tokens, token-mean feature.  DropPath == identity (SURVEY Q10:
the published DropPath returns None).  Same host design as the ResNet encoder: ONE flat fp32 buffer for the
parameters that receive gradients (Linear weights stored [in][out] = the GEMM B operand), one for gradients,
explicit forward/backward launch plans over a workspace allocated once; attention probabilities are kept
(not recomputed) for the backward.  Parameters the modality never touches (the image embedding of the text
encoder and vice versa) live outside the flat buffer: their gradient is None in the reference, so SGD skips them.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import MLAHipError
from .encoder import DEFAULT_CONV_MATH
from .model import ConcatFusion, N_CLASSES, SharedHead, _Classifier
from .module import FlatModule, Holder


def _sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:                      # m3ae.py:181-194
    omega = np.arange(embed_dim // 2, dtype=np.float32)
    omega /= embed_dim / 2.
    omega = 1. / 10000 ** omega
    out = np.einsum('m,d->md', pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_pos_embed(embed_dim: int, length: int, two_d: bool) -> torch.Tensor:
    """get_1d_sincos_pos_embed / get_2d_sincos_pos_embed (m3ae.py:197-223), without the leading 1."""
    if not two_d:
        emb = _sincos_1d(embed_dim, np.arange(length, dtype=np.float32))
    else:
        gs = int(length ** 0.5)
        if gs * gs != length:
            raise MLAHipError("2-D position embedding needs a square patch grid")
        grid = np.stack(np.meshgrid(np.arange(gs, dtype=np.float32), np.arange(gs, dtype=np.float32)), axis=0)   # w first
        grid = grid.reshape([2, 1, gs, gs])
        emb = np.concatenate([_sincos_1d(embed_dim // 2, grid[0]), _sincos_1d(embed_dim // 2, grid[1])], axis=1)
    return torch.from_numpy(emb.astype(np.float32))


# measurement switch (same-box A/B): 0 = bias gradients by separate column-reduction launches
FUSE_BIAS_GRAD = os.environ.get("MLA_FUSE_BIAS_GRAD", "1") != "0"

class M3AEEncoder(FlatModule):
    """nn.Module face (module.py): parameters under the reference's names, order and layouts -- nn.Linear weights
    (out, in) as transposed views of the [in][out] GEMM operands, type embeddings / cls as (1,1,D) -- registered as
    MaskedMultimodalAutoencoder does (direct parameters, text_embedding, image_embedding, encoder: m3ae.py:305-331)
    resp. the audio branch of CAVMAEFT (cav_mae.py:116-145)."""
    BLOCK_PARAMS = [("layer_norm1.weight", "D"), ("layer_norm1.bias", "D"), ("attention.qkv_linear.weight", "D,3D"),
                    ("attention.qkv_linear.bias", "3D"), ("attention.fc.weight", "D,D"), ("attention.fc.bias", "D"),
                    ("layer_norm2.weight", "D"), ("layer_norm2.bias", "D"), ("transformer_mlp.fc1.weight", "D,4D"),
                    ("transformer_mlp.fc1.bias", "4D"), ("transformer_mlp.fc2.weight", "4D,D"), ("transformer_mlp.fc2.bias", "D")]

    def __init__(self, kind: str, device="cuda", depth: int = 12, emb_dim: int = 768, num_heads: int = 12,
                 text_vocab_size: int = 30522, patch_dim: int = 768, seed: Optional[int] = None, conv_math: Optional[str] = None):
        if kind not in ("text", "image", "audio"):
            raise ValueError("kind must be 'text', 'image' or 'audio'")
        super().__init__()
        self.conv_math = conv_math or os.environ.get("MLA_CONV_MATH", DEFAULT_CONV_MATH)     # arithmetic of the Linear GEMMs (encoder.py)
        if self.conv_math not in ("f32", "split"):
            raise MLAHipError(f"conv_math must be 'f32' or 'split', got {self.conv_math!r}")
        self.split = self.conv_math == "split"
        # "fused": one flash-style kernel per direction, scores never materialised (csrc/attention.hip); "materialized": the
        # round-1 form (QK^T -> masked softmax -> PV as strided batched GEMMs, probabilities kept in HBM), kept as a
        # cross-check of the fused kernels at full size and for A/B measurements
        self.attention = os.environ.get("MLA_ATTENTION", "fused")
        if self.attention not in ("fused", "materialized"):
            raise MLAHipError(f"MLA_ATTENTION must be 'fused' or 'materialized', got {self.attention!r}")
        self.kind, self.device = kind, torch.device(device)
        if kind == "audio":
            patch_dim = 256                  # conv 16x16 over 1 channel (cav_mae.py:127)
        self.has_cls = kind != "audio"       # CAV-MAE has no [cls] token
        self.audio_tokens = 512              # audio_length * 128 / 256 (cav_mae.py:131)
        self.depth, self.D, self.H, self.V, self.PD = depth, emb_dim, num_heads, text_vocab_size, patch_dim
        D = emb_dim
        dims = {"D": (D,), "3D": (3 * D,), "4D": (4 * D,), "D,3D": (D, 3 * D), "D,D": (D, D), "D,4D": (D, 4 * D), "4D,D": (4 * D, D)}
        # ---- flat layout of the parameters this modality trains (Linear weights as [in][out])
        lay: List[Tuple[str, Tuple[int, ...]]] = []
        if kind == "text":
            lay += [("text_embedding.weight", (text_vocab_size, D)), ("encoder_text_type_embedding", (D,)), ("cls_token", (D,))]
        elif kind == "image":
            lay += [("image_embedding.weight", (patch_dim, D)), ("image_embedding.bias", (D,)), ("encoder_image_type_embedding", (D,)),
                    ("cls_token", (D,))]
        else:   # CAV-MAE audio: conv patch embed (as [256][D]), modality embedding, LEARNED position embedding (tr_pos=True)
            lay += [("patch_embed_a.proj.weight", (patch_dim, D)), ("patch_embed_a.proj.bias", (D,)), ("modality_a", (D,)),
                    ("pos_embed_a", (self.audio_tokens, D))]
        for i in range(depth):
            lay += [(f"encoder.blocks.{i}.{n}", dims[s]) for n, s in self.BLOCK_PARAMS]
        lay += [("encoder.layer_norm.weight", (D,)), ("encoder.layer_norm.bias", (D,))]
        self.layout: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for name, shp in lay:
            self.layout[name] = (off, shp)
            off += math.prod(shp)
        self.numel = off
        f32 = dict(device=self.device, dtype=torch.float32)
        self.flat = torch.zeros(off, **f32)
        self.grad = torch.zeros(off, **f32)
        self.p = {k: self.flat[o:o + math.prod(s)].view(s) for k, (o, s) in self.layout.items()}
        self.g = {k: self.grad[o:o + math.prod(s)].view(s) for k, (o, s) in self.layout.items()}
        # ---- parameters of the other modality's input path: never used, never receive a gradient (grad stays None, so
        # SGD skips them, like the reference); registered below with their own storage, reference shapes
        if kind == "text":
            unused_shapes = {"image_embedding.weight": (D, patch_dim), "image_embedding.bias": (D,),
                             "encoder_image_type_embedding": (1, 1, D)}
        elif kind == "image":
            unused_shapes = {"text_embedding.weight": (text_vocab_size, D), "encoder_text_type_embedding": (1, 1, D)}
        else:
            unused_shapes = {}   # CAVMAEFT's visual branch / unused norms are not materialised (never touched by forward_feat(.,'a'))
        self.unused: Dict[str, nn.Parameter] = {}
        self._register_reference_tree(unused_shapes)
        self._pos: Dict[int, torch.Tensor] = {}
        self._ws: dict = {}
        self._key = None
        # split-bf16 images of every Linear weight ([K][N] = a 1-tap conv weight): forward (transposed) and input-gradient
        self.wsp: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.tail_stream: Optional[torch.cuda.Stream] = None     # see ResNet18Encoder.tail_stream
        self.training, self._wsplit_dirty = True, True
        if self.split:
            lin = [k for k, (_o, shp) in self.layout.items() if len(shp) == 2 and k.endswith(".weight") and k != "text_embedding.weight"]
            tot16 = sum(2 * 3 * math.prod(self.layout[k][1]) for k in lin)
            self._wsplit_flat = torch.empty(tot16, device=self.device, dtype=torch.int16)
            o16, rows, blocks = 0, [], 0
            for k in lin:
                off, (K, N) = self.layout[k]
                n16 = 3 * K * N
                self.wsp[k] = (self._wsplit_flat[o16:o16 + n16], self._wsplit_flat[o16 + n16:o16 + 2 * n16])
                nb = ((K + 31) // 32) * ((N + 31) // 32)
                for transposed, o in ((1, o16), (0, o16 + n16)):
                    rows.append([off, o, 1, K, N, transposed, blocks, 0])
                    blocks += nb
                o16 += 2 * n16
            self._wsplit_desc = torch.tensor(rows, dtype=torch.int32, device=self.device)
            self._wsplit_blocks = blocks
        self.reset_parameters(seed)

    def _await_tail(self) -> None:
        ts = self.tail_stream
        if ts is not None:
            cur = torch.cuda.current_stream()
            if cur != ts:
                cur.wait_stream(ts)

    def _w(self, name: str, which: int):
        """split image of Linear `name` (0: forward, 1: input gradient) or None in f32 mode"""
        e = self.wsp.get(name)
        return None if e is None else e[which]

    def train(self, mode: bool = True):
        super().train(mode)
        self._wsplit_dirty = True
        return self

    # ---- reference-named parameter tree -----------------------------------------------------------
    def _to_ref(self, name: str):
        """internal flat view -> reference-shaped VIEW (no copy)."""
        D = self.D
        if name == "patch_embed_a.proj.weight":
            return lambda t: t.t().view(D, 1, 16, 16)                                    # Conv2d(1, D, 16, 16) weight (cav_mae.py:80)
        if name in ("cls_token", "encoder_text_type_embedding", "encoder_image_type_embedding", "modality_a"):
            return lambda t: t.view(1, 1, -1)
        if name == "pos_embed_a":
            return lambda t: t.view(1, *t.shape)
        if len(self.layout[name][1]) == 2 and name != "text_embedding.weight":
            return lambda t: t.t()                                                       # nn.Linear.weight is (out, in)
        return lambda t: t

    def _register_reference_tree(self, unused_shapes: Dict[str, Tuple[int, ...]]) -> None:
        if self.kind == "audio":
            order = ["modality_a", "pos_embed_a", "patch_embed_a.proj.weight", "patch_embed_a.proj.bias"]
        else:
            order = ["encoder_image_type_embedding", "encoder_text_type_embedding", "cls_token", "text_embedding.weight",
                     "image_embedding.weight", "image_embedding.bias"]
        order += [k for k in self.layout if k.startswith("encoder.")]
        for name in order:
            if name in self.layout:
                p = self._param_view(self.p[name], self.g[name], self._to_ref(name), self._ref_name(name))
            else:
                p = nn.Parameter(torch.zeros(unused_shapes[name], device=self.device, dtype=torch.float32))
                p._mla_owner = self
                self.unused[name] = p
            parent, leaf = self._descend(self, self._ref_name(name))
            parent.register_parameter(leaf, p)
        self.register_state_dict_pre_hook(lambda m, prefix, keep_vars: m._await_tail())

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        self._await_tail()
        self._wsplit_dirty = True
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def reset_parameters(self, seed: Optional[int] = None) -> None:
        """Module-default initialisation of the reference (m3ae.py:306-324; nn.Linear / nn.LayerNorm defaults)."""
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed) if seed is not None else gen.seed()
        for name, (_o, shp) in self.layout.items():
            t = self.p[name]
            if name == "text_embedding.weight":
                t.copy_(torch.randn(shp, generator=gen))                                   # normal_(0, 1)
            elif name in ("cls_token", "encoder_text_type_embedding", "encoder_image_type_embedding"):
                t.copy_(torch.randn(shp, generator=gen) + 0.02)                            # torch.empty().normal_(0.02): mean .02, std 1
            elif name == "modality_a":
                t.copy_(torch.randn(shp, generator=gen) * 0.02)                            # cav_mae.py:171
            elif name == "pos_embed_a":
                t.copy_(self._cav_pos_embed())                                             # cav_mae.py:160-161 (sin-cos init, then trained)
            elif "layer_norm" in name:
                t.fill_(1.0 if name.endswith("weight") else 0.0)
            elif len(shp) == 2:
                bound = math.sqrt(6.0 / (shp[0] + shp[1])) if name == "image_embedding.weight" else 1.0 / math.sqrt(shp[0])
                t.copy_((torch.rand(shp, generator=gen) * 2 - 1) * bound)
            else:   # nn.Linear bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                fan_in = 4 * self.D if name.endswith("fc2.bias") else (self.PD if name == "image_embedding.bias" else self.D)
                t.copy_((torch.rand(shp, generator=gen) * 2 - 1) / math.sqrt(fan_in))

    def _cav_pos_embed(self) -> torch.Tensor:
        """get_2d_sincos_pos_embed(D, 8, L/8) of cav_mae.py:47-66 (grid 8 x L/8, 'w goes first')."""
        D, L = self.D, self.audio_tokens
        gh, gw = 8, L // 8
        grid = np.stack(np.meshgrid(np.arange(gw, dtype=np.float32), np.arange(gh, dtype=np.float32)), axis=0).reshape([2, 1, gw, gh])
        emb = np.concatenate([_sincos_1d(D // 2, grid[0]), _sincos_1d(D // 2, grid[1])], axis=1)
        return torch.from_numpy(emb.astype(np.float32))

    def _ref_name(self, name: str) -> str:
        """internal (M3AE-style) parameter name -> reference state_dict key."""
        if self.kind != "audio" or not name.startswith("encoder."):
            return name
        if name.startswith("encoder.layer_norm."):
            return name.replace("encoder.layer_norm.", "norm_a.")                          # cav_mae.py:145, 349
        _e, _b, i, rest = name.split(".", 3)
        i = int(i)
        shared = i >= self.depth - 1                                                       # 11 blocks_a + 1 blocks_u (cav_mae.py:141-143)
        pre = f"blocks_u.{i - (self.depth - 1)}." if shared else f"blocks_a.{i}."
        sub = {"layer_norm1": "norm1_a" if shared else "norm1", "layer_norm2": "norm2_a" if shared else "norm2",
               "attention.qkv_linear": "attn.qkv", "attention.fc": "attn.proj", "transformer_mlp.fc1": "mlp.fc1",
               "transformer_mlp.fc2": "mlp.fc2"}
        for k, v in sub.items():
            if rest.startswith(k + "."):
                return pre + v + rest[len(k):]
        raise KeyError(name)

    def grads_as_reference(self) -> Dict[str, torch.Tensor]:
        out = {}
        for name in self.layout:
            t = self.g[name]
            if self.kind == "audio":
                key = self._ref_name(name)
                if name == "patch_embed_a.proj.weight":
                    out[key] = t.t().contiguous().view(self.D, 1, 16, 16)
                elif name == "modality_a":
                    out[key] = t.clone().view(1, 1, -1)
                elif name == "pos_embed_a":
                    out[key] = t.clone().view(1, *t.shape)
                else:
                    out[key] = t.t().contiguous() if t.dim() == 2 else t.clone()
                continue
            if name in ("cls_token", "encoder_text_type_embedding", "encoder_image_type_embedding"):
                out[name] = t.clone().view(1, 1, -1)
            elif t.dim() == 2 and name != "text_embedding.weight":
                out[name] = t.t().contiguous()
            else:
                out[name] = t.clone()
        return out

    # ------------------------------------------------------------------------------------------
    def _plan(self, B: int, L: int) -> dict:
        if self._key == (B, L):
            return self._ws
        D, H, n = self.D, self.H, L + (1 if self.has_cls else 0)
        M = B * n
        f32 = dict(device=self.device, dtype=torch.float32)
        ws: dict = {"B": B, "L": L, "n": n, "M": M}
        ws["x0"] = torch.empty((M, D), **f32)
        ws["blocks"] = []
        for _ in range(self.depth):
            ws["blocks"].append({"h1": torch.empty((M, D), **f32), "qkv": torch.empty((M, 3 * D), **f32),
                                 "P": torch.empty((B, H, n, n), **f32), "o": torch.empty((M, D), **f32),
                                 "xmid": torch.empty((M, D), **f32), "h2": torch.empty((M, D), **f32),
                                 "u": torch.empty((M, 4 * D), **f32), "gl": torch.empty((M, 4 * D), **f32),
                                 "xout": torch.empty((M, D), **f32),
                                 "st": torch.empty((4, M), **f32)})          # mean1, rstd1, mean2, rstd2
        ws["y"] = torch.empty((M, D), **f32)
        ws["stf"] = torch.empty((2, M), **f32)
        ws["feat"] = torch.empty((B, D), **f32)
        if self.kind != "text":
            ws["patches"] = torch.empty((B * L, self.PD), **f32)
        if self.kind != "audio":
            ws["pos"] = sincos_pos_embed(D, L, two_d=(self.kind == "image")).to(self.device)
        if self.attention == "fused":
            for bk in ws["blocks"]:
                del bk["P"]
                bk["lse"] = torch.empty((B, H, n), **f32)
        self._ws, self._key = ws, (B, L)
        return ws

    def _bwd_ws(self, ws: dict) -> None:
        if "dA" in ws:
            return
        D, H, n, M, B = self.D, self.H, ws["n"], ws["M"], ws["B"]
        f32 = dict(device=self.device, dtype=torch.float32)
        ws["dA"], ws["dB"], ws["dC"] = (torch.empty((M, D), **f32) for _ in range(3))
        ws["du"] = torch.empty((M, 4 * D), **f32)
        ws["dqkv"] = torch.empty((M, 3 * D), **f32)
        if self.attention == "fused":
            ws["dvec"] = torch.empty((B, H, n), **f32)
        else:
            ws["dP"] = torch.empty((B, H, n, n), **f32)
        ws["wt_ws"] = torch.empty(4 * D * D, **f32)
        wb = max(ops.linear_wgrad_ws_bytes(m_, k_, n_, sp) for sp in ((False, True) if self.split else (False,))
                 for (m_, k_, n_) in ((M, D, 3 * D), (M, D, 4 * D), (M, 4 * D, D), (M, D, D), (B * ws["L"], self.PD, D)))
        ws["wgrad_ws"] = torch.empty((wb + 3) // 4, **f32)
        ws["red_ws"] = torch.empty(ops.colreduce_ws_elems(M, 4 * D), **f32)
        ws["colsum"] = torch.empty(D, **f32)

    # ------------------------------------------------------------------------------------------
    def forward(self, inp: torch.Tensor, padding_mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """text: inp = token ids (B,1,L) or (B,L) int64, padding_mask (B,1,L)/(B,L) float (1 = padded);
        image: inp = (B,3,256,256) fp32.  Returns the (B, D) token-mean feature (basic_model.py:182-200), written into
        `out` if given (else into the reused workspace buffer)."""
        self._await_tail()
        st = ops.cur_stream()
        D, H = self.D, self.H
        hd = D // H
        if self.split and (self.training or self._wsplit_dirty):
            ops.conv2d_wsplit_batch(self.flat, self._wsplit_flat, self._wsplit_desc, self._wsplit_blocks, stream=st)
            self._wsplit_dirty = False
        if self.kind == "text":
            ids = inp.reshape(inp.shape[0], -1).contiguous()                                 # token.squeeze(1)
            B, L = ids.shape
            ws = self._plan(B, L)
            pm = padding_mask.reshape(B, -1).to(torch.float32)
            if "ids" not in ws:
                ws["ids"] = torch.empty_like(ids)
            ws["ids"].copy_(ids)          # owned copy: tokens_assemble_bwd reads it after forward() returned (feeder slots are reused)
            ids = ws["ids"]
            ws["pm"] = torch.cat([torch.zeros((B, 1), device=self.device), pm], dim=1).contiguous()   # cls is never masked (m3ae.py:347)
            ops.tokens_assemble(ws["x0"], self.p["text_embedding.weight"], ids, ws["pos"], self.p["encoder_text_type_embedding"],
                                self.p["cls_token"], B, L, D, stream=st)
        elif self.kind == "audio":
            # cav_mae.py:337-343: (B, time, freq) -> unsqueeze, transpose -> conv16x16/16 -> (B, 8*64, D) + pos + modality
            B, T_, F_ = inp.shape
            L = (F_ // 16) * (T_ // 16)
            if L != self.audio_tokens:
                raise MLAHipError(f"audio encoder expects {self.audio_tokens} patches, got {L}")
            ws = self._plan(B, L)
            ws["pm"] = None
            ops.patchify(inp.contiguous().float(), ws["patches"], 16, transposed_hw=(F_, T_), stream=st)
            ops.linear_fwd(ws["patches"], self.p["patch_embed_a.proj.weight"], self.p["patch_embed_a.proj.bias"], ws["x0"], 1, B * L,
                           self.PD, D, stream=st, wsplit=self._w("patch_embed_a.proj.weight", 0))
            ops.tokens_assemble(ws["x0"], None, None, self.p["pos_embed_a"], self.p["modality_a"], None, B, L, D, stream=st)
        else:
            B = inp.shape[0]
            L = (inp.shape[2] // 16) * (inp.shape[3] // 16)
            ws = self._plan(B, L)
            ws["pm"] = None
            ops.patchify(inp.contiguous(), ws["patches"], 16, stream=st)                     # basic_model.py:184-186
            ops.linear_fwd(ws["patches"], self.p["image_embedding.weight"], self.p["image_embedding.bias"], ws["x0"], B, L,
                           self.PD, D, y_group_rows=L + 1, y_off=1, stream=st, wsplit=self._w("image_embedding.weight", 0))               # m3ae.py:353
            ops.tokens_assemble(ws["x0"], None, None, ws["pos"], self.p["encoder_image_type_embedding"], self.p["cls_token"],
                                B, L, D, stream=st)
        n, M = ws["n"], ws["M"]
        scale = hd ** -0.5
        x = ws["x0"]
        for i, bk in enumerate(ws["blocks"]):
            P_ = lambda nm: self.p[f"encoder.blocks.{i}.{nm}"]
            bk["x"] = x
            ops.layernorm_fwd(x, P_("layer_norm1.weight"), P_("layer_norm1.bias"), bk["h1"], bk["st"][0], bk["st"][1], M, D, stream=st)
            ops.linear_fwd(bk["h1"], P_("attention.qkv_linear.weight"), P_("attention.qkv_linear.bias"), bk["qkv"], 1, M, D, 3 * D, stream=st, wsplit=self._w(f"encoder.blocks.{i}.attention.qkv_linear.weight", 0))
            if self.attention == "fused":
                ops.attention_fwd(bk["qkv"], ws["pm"], bk["o"], bk["lse"], B, H, n, hd, stream=st)                                   # m3ae.py:109-122
            else:
                qs, ss, os_ = (n * 3 * D, hd, 3 * D, 1), (H * n * n, n * n, n, 1), (n * D, hd, D, 1)
                ops.bgemm(bk["qkv"], bk["qkv"], bk["P"], B, H, n, n, hd, qs, (n * 3 * D, hd, 1, 3 * D), ss, scale, b_off=D, stream=st)   # m3ae.py:109
                ops.softmax_fwd(bk["P"], ws["pm"], B, H, n, stream=st)                                                                    # :111-118
                ops.bgemm(bk["P"], bk["qkv"], bk["o"], B, H, n, hd, n, ss, (n * 3 * D, hd, 3 * D, 1), os_, 1.0, b_off=2 * D, stream=st)   # :121-122
            ops.linear_fwd(bk["o"], P_("attention.fc.weight"), P_("attention.fc.bias"), bk["xmid"], 1, M, D, D, residual=x, stream=st, wsplit=self._w(f"encoder.blocks.{i}.attention.fc.weight", 0))  # :123,149
            ops.layernorm_fwd(bk["xmid"], P_("layer_norm2.weight"), P_("layer_norm2.bias"), bk["h2"], bk["st"][2], bk["st"][3], M, D, stream=st)
            ops.linear_fwd(bk["h2"], P_("transformer_mlp.fc1.weight"), P_("transformer_mlp.fc1.bias"), bk["u"], 1, M, D, 4 * D,
                           y_gelu=bk["gl"], stream=st, wsplit=self._w(f"encoder.blocks.{i}.transformer_mlp.fc1.weight", 0))                                                                               # :76-77
            ops.linear_fwd(bk["gl"], P_("transformer_mlp.fc2.weight"), P_("transformer_mlp.fc2.bias"), bk["xout"], 1, M, 4 * D, D,
                           residual=bk["xmid"], stream=st, wsplit=self._w(f"encoder.blocks.{i}.transformer_mlp.fc2.weight", 0))                                                                            # :79,154
            x = bk["xout"]
        ws["xlast"] = x
        ops.layernorm_fwd(x, self.p["encoder.layer_norm.weight"], self.p["encoder.layer_norm.bias"], ws["y"], ws["stf"][0], ws["stf"][1],
                          M, D, stream=st)                                                                                            # m3ae.py:176
        feat = ws["feat"] if out is None else out
        ops.avgpool_fwd(ws["y"], feat, B, n, D, stream=st)                                    # .mean(dim=1) over all 1+L tokens
        self._pa = n
        return feat

    # ------------------------------------------------------------------------------------------
    def _wgrad_bias(self, x, dy, dw, db, wgw, red, M: int, K: int, N: int, st) -> None:
        """Weight and bias gradient of one Linear (dw = x^T dy, db = column sums of dy).  On the split arithmetic the bias
        gradient comes out of the weight-gradient kernel's own pass over dy (no separate column-reduction launches)."""
        if self.split and FUSE_BIAS_GRAD:
            ops.linear_wgrad(x, dy, dw, wgw, 1, M, K, N, stream=st, split=True, dbias=db)
        else:
            ops.colsum_rows(dy, db, red, M, N, stream=st)
            ops.linear_wgrad(x, dy, dw, wgw, 1, M, K, N, stream=st, split=self.split)

    def backward_from_pooled(self, dfeat: torch.Tensor, P: Optional[int] = None) -> None:
        ws = self._ws
        self._bwd_ws(ws)
        st = ops.cur_stream()
        D, H, n, M, B, L = self.D, self.H, ws["n"], ws["M"], ws["B"], ws["L"]
        hd = D // H
        scale = hd ** -0.5
        dA, dB_, dC = ws["dA"], ws["dB"], ws["dC"]
        red, wtw, wgw = ws["red_ws"], ws["wt_ws"], ws["wgrad_ws"]
        ops.avgpool_bwd(dfeat.contiguous(), dA, B, n, D, stream=st)                                             # d y
        ops.layernorm_bwd(dA, ws["xlast"], self.p["encoder.layer_norm.weight"], ws["stf"][0], ws["stf"][1], dA,
                          self.g["encoder.layer_norm.weight"], self.g["encoder.layer_norm.bias"], red, M, D, stream=st)
        dx = dA                                                                                                 # grad wrt block output
        for i in reversed(range(self.depth)):
            bk = ws["blocks"][i]
            P_ = lambda nm: self.p[f"encoder.blocks.{i}.{nm}"]
            G_ = lambda nm: self.g[f"encoder.blocks.{i}.{nm}"]
            # ---- MLP: xout = xmid + fc2(gelu(fc1(LN2(xmid))))
            self._wgrad_bias(bk["gl"], dx, G_("transformer_mlp.fc2.weight"), G_("transformer_mlp.fc2.bias"), wgw, red, M, 4 * D, D, st)
            ops.linear_dgrad(dx, P_("transformer_mlp.fc2.weight"), ws["du"], wtw, 1, M, 4 * D, D, gelu_src=bk["u"], stream=st, wsplit=self._w(f"encoder.blocks.{i}.transformer_mlp.fc2.weight", 1))
            self._wgrad_bias(bk["h2"], ws["du"], G_("transformer_mlp.fc1.weight"), G_("transformer_mlp.fc1.bias"), wgw, red, M, D, 4 * D, st)
            ops.linear_dgrad(ws["du"], P_("transformer_mlp.fc1.weight"), dB_, wtw, 1, M, D, 4 * D, stream=st, wsplit=self._w(f"encoder.blocks.{i}.transformer_mlp.fc1.weight", 1))   # d h2
            ops.layernorm_bwd(dB_, bk["xmid"], P_("layer_norm2.weight"), bk["st"][2], bk["st"][3], dB_, G_("layer_norm2.weight"),
                              G_("layer_norm2.bias"), red, M, D, add=dx, stream=st)                             # d xmid -> dB_
            # ---- attention: xmid = x + fc(PV)
            self._wgrad_bias(bk["o"], dB_, G_("attention.fc.weight"), G_("attention.fc.bias"), wgw, red, M, D, D, st)
            ops.linear_dgrad(dB_, P_("attention.fc.weight"), dC, wtw, 1, M, D, D, stream=st, wsplit=self._w(f"encoder.blocks.{i}.attention.fc.weight", 1))                    # d o (B,n,D)
            if self.attention == "fused":
                ops.attention_bwd(dC, bk["qkv"], bk["o"], bk["lse"], ws["pm"], ws["dqkv"], ws["dvec"], B, H, n, hd, stream=st)
            else:
                qs, ss, os_ = (n * 3 * D, hd, 3 * D, 1), (H * n * n, n * n, n, 1), (n * D, hd, D, 1)
                ops.bgemm(dC, bk["qkv"], ws["dP"], B, H, n, n, hd, os_, (n * 3 * D, hd, 1, 3 * D), ss, 1.0, b_off=2 * D, stream=st)          # dP = dO V^T
                ops.bgemm(bk["P"], dC, ws["dqkv"], B, H, n, hd, n, (H * n * n, n * n, 1, n), (n * D, hd, D, 1), qs, 1.0, c_off=2 * D, stream=st)   # dV = P^T dO
                ops.softmax_bwd(bk["P"], ws["dP"], B, H, n, stream=st)                                                                       # dS
                ops.bgemm(ws["dP"], bk["qkv"], ws["dqkv"], B, H, n, hd, n, ss, (n * 3 * D, hd, 3 * D, 1), qs, scale, b_off=D, stream=st)     # dQ = s dS K
                ops.bgemm(ws["dP"], bk["qkv"], ws["dqkv"], B, H, n, hd, n, (H * n * n, n * n, 1, n), (n * 3 * D, hd, 3 * D, 1), qs, scale,
                          c_off=D, stream=st)                                                                                                # dK = s dS^T Q
            self._wgrad_bias(bk["h1"], ws["dqkv"], G_("attention.qkv_linear.weight"), G_("attention.qkv_linear.bias"), wgw, red, M, D, 3 * D, st)
            ops.linear_dgrad(ws["dqkv"], P_("attention.qkv_linear.weight"), dC, wtw, 1, M, D, 3 * D, stream=st, wsplit=self._w(f"encoder.blocks.{i}.attention.qkv_linear.weight", 1))  # d h1
            ops.layernorm_bwd(dC, bk["x"], P_("layer_norm1.weight"), bk["st"][0], bk["st"][1], dC, G_("layer_norm1.weight"),
                              G_("layer_norm1.bias"), red, M, D, add=dB_, stream=st)                            # d x -> dC
            dx, dC = dC, dx                                                                                     # rotate buffers
            ws["dA"], ws["dC"] = dx, dC
        # ---- token assembly (m3ae.py:342-366)
        ops.colsum_rows(dx, ws["colsum"], red, M, D, stream=st)
        if self.kind == "audio":
            # x0 = conv(patches) + pos_embed_a + modality_a: d modality = d conv-bias = sum over all tokens,
            # d pos[i] = sum over the batch, d conv-weight = patches^T dx
            self.g["modality_a"].copy_(ws["colsum"])
            self.g["patch_embed_a.proj.bias"].copy_(ws["colsum"])
            if "red_pos" not in ws:
                ws["red_pos"] = torch.empty(ops.colreduce_ws_elems(B, L * D), device=self.device, dtype=torch.float32)
            ops.colsum_rows(dx, self.g["pos_embed_a"], ws["red_pos"], B, L * D, stream=st)
            ops.linear_wgrad(ws["patches"], dx, self.g["patch_embed_a.proj.weight"], wgw, 1, B * L, self.PD, D, stream=st, split=self.split)
        elif self.kind == "text":
            self.g["text_embedding.weight"].zero_()
            if "emb_ws" not in ws:
                ws["emb_ws"] = torch.empty(ops.tokens_assemble_bwd_ws_bytes(B, L, D), device=self.device, dtype=torch.uint8)
            ops.tokens_assemble_bwd(dx, ws["colsum"], ws["ids"], self.g["cls_token"], self.g["encoder_text_type_embedding"],
                                    self.g["text_embedding.weight"], B, L, D, stream=st, ws=ws["emb_ws"])
        else:
            ops.tokens_assemble_bwd(dx, ws["colsum"], None, self.g["cls_token"], self.g["encoder_image_type_embedding"], None,
                                    B, L, D, stream=st)
            dimg = dx.view(B, n, D)[:, 1:, :].contiguous().view(B * L, D)       # memory plumbing: drop the cls rows
            ops.colsum_rows(dimg, self.g["image_embedding.bias"], red, B * L, D, stream=st)
            ops.linear_wgrad(ws["patches"], dimg, self.g["image_embedding.weight"], wgw, 1, B * L, self.PD, D, stream=st, split=self.split)


class M3AEClassifier(_Classifier):
    """models/basic_model.py:127-200 under --gs_flag: mae_a (text) + mae_v (image) + ConcatFusion(768 -> C)."""

    def __init__(self, args, device="cuda", depth: int = 12, text_vocab_size: int = 30522, seed: Optional[int] = None,
                 conv_math: Optional[str] = None):
        super().__init__()
        fusion = getattr(args, "fusion_method", "concat")
        dataset = getattr(args, "dataset", "Food101")
        if dataset not in ("MVSA", "Food101", "CREMAD"):                            # basic_model.py:132-144
            raise NotImplementedError("Incorrect dataset name {}".format(dataset))
        if fusion != "concat":                                                      # basic_model.py:146-163
            raise NotImplementedError("Incorrect fusion method: {}!".format(fusion))
        if not getattr(args, "gs_flag", False):
            raise NotImplementedError("mla_hip implements the --gs_flag (MLA) path only")
        self.args, self.device = args, torch.device(device)
        s = (lambda k: None if seed is None else seed + k)
        self.fusion_module = ConcatFusion(768, N_CLASSES[dataset], device, s(2))   # basic_model.py:149
        self.mae_a = M3AEEncoder("text", device, depth=depth, text_vocab_size=text_vocab_size, seed=s(0), conv_math=conv_math)    # :166
        self.mae_v = M3AEEncoder("image", device, depth=depth, text_vocab_size=text_vocab_size, seed=s(1), conv_math=conv_math)   # :167

    def mla_encoders(self):
        return [("a", "text", self.mae_a), ("v", "image", self.mae_v)]

    def forward_raw(self, token: torch.Tensor, padding_mask: torch.Tensor, visual: torch.Tensor):
        """Kernel-level joint forward into the reused feature buffers (MLATrainer / Evaluator; no autograd)."""
        return self.mae_a.forward(token, padding_mask), self.mae_v.forward(visual)

    def forward(self, token: torch.Tensor, padding_mask: torch.Tensor, visual: torch.Tensor):
        """a, v = model(token, padding_mask, image)  (main.py:426; basic_model.py:182-200), with autograd history."""
        B, D = visual.shape[0], self.mae_a.D
        a = self._feature(self.mae_a, lambda out: self.mae_a.forward(token, padding_mask, out), B, D)
        v = self._feature(self.mae_v, lambda out: self.mae_v.forward(visual, None, out), B, D)
        return a, v

    def forward_split(self, token: torch.Tensor, padding_mask: torch.Tensor, visual: torch.Tensor):
        """Per-encoder forward closures in alternation order (for the trainer's per-encoder streams)."""
        return [lambda: self.mae_a.forward(token, padding_mask), lambda: self.mae_v.forward(visual)]


class ConcatFusion3(nn.Module):
    """models/fusion_modules.py:26-35; under --gs_flag only `fc_out` is touched (main.py:432, 444, 456)."""

    def __init__(self, input_dim: int = 768, output_dim: int = 4, device="cuda", seed: Optional[int] = None):
        super().__init__()
        self.fc_out = SharedHead(input_dim, output_dim, device, seed)

    def forward(self, x, y, z):
        raise NotImplementedError("mla_hip implements the --gs_flag (MLA) path only: fc_out is applied per modality")


class Modal3Classifier(_Classifier):
    """models/basic_model.py:202-275 under --gs_flag: CAV-MAE audio (mae_a) + M3AE image (mae_v) + M3AE text (mae_t),
    shared head Linear(768 -> 4); MLA alternates a -> v -> t (main.py:432-466)."""

    def __init__(self, args, device="cuda", depth: int = 12, text_vocab_size: int = 30522, seed: Optional[int] = None,
                 conv_math: Optional[str] = None):
        super().__init__()
        fusion = getattr(args, "fusion_method", "concat")
        dataset = getattr(args, "dataset", "IEMOCAP")
        if dataset != "IEMOCAP":                                                    # basic_model.py:208-211
            raise NotImplementedError("Incorrect dataset name {}".format(dataset))
        if fusion != "concat":                                                      # basic_model.py:213-229
            raise NotImplementedError("Incorrect fusion method: {}!".format(fusion))
        if not getattr(args, "gs_flag", False):
            raise NotImplementedError("mla_hip implements the --gs_flag (MLA) path only")
        self.args, self.device = args, torch.device(device)
        s = (lambda k: None if seed is None else seed + k)
        self.fusion_module = ConcatFusion3(768, N_CLASSES[dataset], device, s(3))  # basic_model.py:218
        self.mae_a = M3AEEncoder("audio", device, depth=depth, seed=s(0), conv_math=conv_math)                                     # :231 CAVMAEFT
        self.mae_v = M3AEEncoder("image", device, depth=depth, text_vocab_size=text_vocab_size, seed=s(1), conv_math=conv_math)   # :232
        self.mae_t = M3AEEncoder("text", device, depth=depth, text_vocab_size=text_vocab_size, seed=s(2), conv_math=conv_math)    # :233

    def mla_encoders(self):
        return [("a", "audio", self.mae_a), ("v", "image", self.mae_v), ("t", "text", self.mae_t)]

    def forward_raw(self, token, padding_mask, visual, audio):
        return self.mae_a.forward(audio), self.mae_v.forward(visual), self.mae_t.forward(token, padding_mask)

    def forward(self, token, padding_mask, visual, audio):
        """a, v, t = model(token, padding_mask, image, spec)  (main.py:424; basic_model.py:252-275), with autograd history."""
        B, D = visual.shape[0], self.mae_a.D
        a = self._feature(self.mae_a, lambda out: self.mae_a.forward(audio, None, out), B, D)
        v = self._feature(self.mae_v, lambda out: self.mae_v.forward(visual, None, out), B, D)
        t = self._feature(self.mae_t, lambda out: self.mae_t.forward(token, padding_mask, out), B, D)
        return a, v, t

    def forward_split(self, token, padding_mask, visual, audio):
        return [lambda: self.mae_a.forward(audio), lambda: self.mae_v.forward(visual),
                lambda: self.mae_t.forward(token, padding_mask)]
