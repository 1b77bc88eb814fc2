"""On-disk feature formats of the reference's offline preprocessing (second half) -> pinned host batches.

What the reference writes, one file per sample, with `np.save`:
    <audio_feature_path>/<name>.npy          fbank, float32 (1024, 128): kaldi fbank, 128 mel bins, 10 ms shift, zero-padded /
                                             cut to 1024 frames (data/extract_fbank.py:35-54); read by dataset/dataset.py:114-117
    <text_feature_path>/<name>_token.npy     BERT token ids, int64 (1, 256)  (data/extract_token.py:38-61)
    <text_feature_path>/<name>_pm.npy        padding mask, float32 (1, 256), 1 = padded   (read at dataset/dataset.py:452-457)
Frames / images are JPEGs decoded and augmented with PIL + torchvision in the reference (dataset/dataset.py:120-155,
401-446): that pipeline is out of scope; a caller-supplied function provides the image tensor of a sample.

`NpyBatcher` turns a list of sample names into batches in the reference's tuple order (dataset/dataset.py:161, 480, 803):
files are opened memory-mapped and copied straight into a small ring of PINNED staging tensors, so `DeviceFeeder`'s copies
are asynchronous without a DataLoader(pin_memory=True) thread in between.  Shapes and dtypes are validated per file: a
wrong file raises with its path instead of feeding garbage to the kernels.
"""
from __future__ import annotations

import os
from typing import Callable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import MLAHipError

FBANK_SHAPE, TOKEN_SHAPE = (1024, 128), (1, 256)


def _load(path: str, shape: Tuple[int, ...], dtype) -> np.ndarray:
    try:
        a = np.load(path, mmap_mode="r", allow_pickle=False)
    except Exception as e:                                   # missing file, truncated file, pickled payload, ...
        raise MLAHipError(f"{path}: cannot read ({e})") from e
    if tuple(a.shape) != tuple(shape) or a.dtype != np.dtype(dtype):
        raise MLAHipError(f"{path}: expected {np.dtype(dtype).name}{tuple(shape)}, found {a.dtype.name}{tuple(a.shape)}")
    return a


def load_fbank(audio_feature_path: str, name: str) -> np.ndarray:
    """float32 (1024, 128), as `np.load(audio_path)` at dataset/dataset.py:117."""
    return _load(os.path.join(audio_feature_path, name + ".npy"), FBANK_SHAPE, np.float32)


def load_token(text_feature_path: str, name: str) -> Tuple[np.ndarray, np.ndarray]:
    """(token int64 (1, 256), padding_mask float32 (1, 256)), dataset/dataset.py:452-455."""
    return (_load(os.path.join(text_feature_path, name + "_token.npy"), TOKEN_SHAPE, np.int64),
            _load(os.path.join(text_feature_path, name + "_pm.npy"), TOKEN_SHAPE, np.float32))


class NpyBatcher:
    def __init__(self, names: Sequence[str], labels: Sequence[int], batch_size: int, audio_feature_path: Optional[str] = None,
                 text_feature_path: Optional[str] = None, image_fn: Optional[Callable[[str], torch.Tensor]] = None,
                 order: str = "av", ring: int = 4, pin: Optional[bool] = None, drop_last: bool = False):
        """order: "av"  -> (spec, image, label, idx)                       AVDataset        dataset/dataset.py:161
                  "tv"  -> (token, padding_mask, image, label, idx)        TVDataset / M3AE dataset/dataset.py:480
                  "tva" -> (token, padding_mask, image, spec, label, idx)  Modal3Dataset    dataset/dataset.py:803
        `image_fn(name)` returns the image tensor of a sample (frames (3, T, 224, 224) or (3, 256, 256)); required."""
        if order not in ("av", "tv", "tva"):
            raise ValueError("order must be 'av', 'tv' or 'tva'")
        if len(names) != len(labels):
            raise ValueError("names and labels differ in length")
        if image_fn is None:
            raise ValueError("image_fn is required (the JPEG pipeline of the reference is not part of this package)")
        if ("a" in order and audio_feature_path is None) or ("t" in order and text_feature_path is None):
            raise ValueError(f"order {order!r} needs " + ("audio_feature_path" if "a" in order and audio_feature_path is None else "text_feature_path"))
        self.names, self.labels, self.B = list(names), [int(x) for x in labels], int(batch_size)
        self.audio, self.text, self.image_fn, self.order = audio_feature_path, text_feature_path, image_fn, order
        self.drop_last = drop_last
        self.pin = torch.cuda.is_available() if pin is None else bool(pin)
        # A staging tuple is reused only after the H2D copies that read it have COMPLETED: the consumer (DeviceFeeder) hands
        # the event it recorded behind those copies to `copied()`, and the slot's refill synchronizes on it.  ("Issued" is not
        # enough: DeviceFeeder's copy stream waits on device events of earlier steps and MLATrainer never host-syncs, so the
        # host can run many batches ahead of the DMA engine.)  Without a consumer that calls `copied()` -- list(batcher),
        # a plain loop -- nothing is fenced, and nothing needs to be: such a consumer reads a batch before it asks for the next.
        self.ring = max(2, ring)
        self._stage: List[Optional[tuple]] = [None] * self.ring
        self._fence: List[Optional[object]] = [None] * self.ring
        self._unfenced: List[int] = []    # staging slots yielded and not yet fenced, oldest first

    def __len__(self) -> int:
        n = len(self.names)
        return n // self.B if self.drop_last else (n + self.B - 1) // self.B

    def _staging(self, k: int, b: int, image_shape) -> tuple:
        st = self._stage[k]
        if st is None or st[0].shape[0] != b:
            mk = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=self.pin)
            parts = {"spec": mk((b,) + FBANK_SHAPE, torch.float32), "token": mk((b,) + TOKEN_SHAPE, torch.int64),
                     "pm": mk((b,) + TOKEN_SHAPE, torch.float32), "image": mk((b,) + tuple(image_shape), torch.float32),
                     "label": mk((b,), torch.int64), "idx": mk((b, 1), torch.int64)}
            keys = {"av": ("spec", "image", "label", "idx"), "tv": ("token", "pm", "image", "label", "idx"),
                    "tva": ("token", "pm", "image", "spec", "label", "idx")}[self.order]
            st = tuple(parts[k_] for k_ in keys) + (keys,)
            self._stage[k] = st
        return st

    def copied(self, event) -> None:
        """Consumer hook: `event` (anything with .synchronize(), e.g. torch.cuda.Event) completes when the asynchronous copies
        out of the OLDEST batch not yet reported have finished.  DeviceFeeder calls it once per batch, in order."""
        if self._unfenced:
            self._fence[self._unfenced.pop(0)] = event

    def __iter__(self) -> Iterator[tuple]:
        k = 0
        self._unfenced = []
        for b0 in range(0, len(self.names), self.B):
            ids = range(b0, min(b0 + self.B, len(self.names)))
            if self.drop_last and len(ids) < self.B:
                return
            if self._fence[k] is not None:             # the DMA out of this staging tuple must have run before it is refilled
                self._fence[k].synchronize()
                self._fence[k] = None
            if k in self._unfenced:                    # consumer without copied(): it has consumed the batch by now
                self._unfenced.remove(k)
            first_img = self.image_fn(self.names[ids[0]])
            *tensors, keys = self._staging(k, len(ids), first_img.shape)
            out = dict(zip(keys, tensors))
            for j, i in enumerate(ids):
                name = self.names[i]
                if "spec" in out:
                    np.copyto(out["spec"][j].numpy(), load_fbank(self.audio, name))      # mmap -> pinned staging, one copy
                if "token" in out:
                    tok, pm = load_token(self.text, name)
                    np.copyto(out["token"][j].numpy(), tok)
                    np.copyto(out["pm"][j].numpy(), pm)
                img = first_img if j == 0 else self.image_fn(name)
                if tuple(img.shape) != tuple(out["image"].shape[1:]):
                    raise MLAHipError(f"{name}: image tensor {tuple(img.shape)} does not match {tuple(out['image'].shape[1:])}")
                out["image"][j].copy_(img)
                out["label"][j] = self.labels[i]
                out["idx"][j, 0] = i
            self._unfenced.append(k)
            yield tuple(tensors)
            k = (k + 1) % self.ring
