"""CPU: the reference's on-disk feature formats: files written exactly the way data/extract_fbank.py:35-54 and
data/extract_token.py:38-61 write them (`np.save` of a float32 (1024,128) fbank; int64 (1,256) ids; float32 (1,256) mask) are
read back into batches in the dataset tuple orders of dataset/dataset.py:161, 480, 803."""
import os

import numpy as np
import pytest
import torch


def _write_reference_style(tmp, names):
    rng = np.random.default_rng(0)
    audio, text = os.path.join(tmp, "audio"), os.path.join(tmp, "text")
    os.makedirs(audio), os.makedirs(text)
    truth = {}
    for n in names:
        frames = int(rng.integers(300, 1400))
        fbank = torch.from_numpy(rng.standard_normal((frames, 128)).astype(np.float32))
        p = 1024 - frames                                           # data/extract_fbank.py:41-50: zero-pad or cut to 1024 frames
        fbank = torch.nn.ZeroPad2d((0, 0, 0, p))(fbank) if p > 0 else fbank[0:1024, :]
        np.save(os.path.join(audio, n + ".npy"), fbank.numpy())
        L = int(rng.integers(5, 257))
        ids = np.zeros((256,), np.int64)
        ids[:L] = rng.integers(1000, 30522, L)
        attn = (np.arange(256) < L).astype(np.int64)
        tokenized_caption = torch.from_numpy(ids)[None, ...]        # data/extract_token.py:50
        padding_mask = torch.from_numpy((1.0 - attn.astype(np.float32))[None, ...])      # :51-52
        np.save(os.path.join(text, n + "_token.npy"), np.array(tokenized_caption))       # :64
        np.save(os.path.join(text, n + "_pm.npy"), np.array(padding_mask))               # :65
        truth[n] = (fbank.numpy(), ids[None], padding_mask.numpy())
    return audio, text, truth


def test_reference_npy_formats_roundtrip(tmp_path):
    from mla_hip import MLAHipError, NpyBatcher, load_fbank, load_token
    names = [f"clip{i:03d}" for i in range(7)]
    labels = [i % 6 for i in range(7)]
    audio, text, truth = _write_reference_style(str(tmp_path), names)
    fb = load_fbank(audio, names[2])
    assert fb.shape == (1024, 128) and fb.dtype == np.float32 and np.array_equal(fb, truth[names[2]][0])
    tok, pm = load_token(text, names[3])
    assert tok.shape == (1, 256) and tok.dtype == np.int64 and pm.dtype == np.float32 and np.array_equal(pm, truth[names[3]][2])
    images = {n: torch.full((3, 2, 8, 8), float(i)) for i, n in enumerate(names)}
    seen = 0
    for order, width in (("av", 4), ("tv", 5), ("tva", 6)):
        batches = list(NpyBatcher(names, labels, 3, audio, text, image_fn=lambda n: images[n], order=order, pin=False))
        assert len(batches) == 3 and all(len(b) == width for b in batches) and batches[-1][0].shape[0] == 1
        b0 = 0
        for batch in batches:
            named = dict(zip({"av": ("spec", "image", "label", "idx"), "tv": ("token", "pm", "image", "label", "idx"),
                              "tva": ("token", "pm", "image", "spec", "label", "idx")}[order], batch))
            n_b = named["label"].shape[0]
            for j in range(n_b):
                n = names[b0 + j]
                if "spec" in named:
                    assert named["spec"].dtype == torch.float32 and np.array_equal(named["spec"][j].numpy(), truth[n][0])
                if "token" in named:
                    assert named["token"].shape[1:] == (1, 256) and named["token"].dtype == torch.int64
                    assert np.array_equal(named["token"][j].numpy(), truth[n][1]) and np.array_equal(named["pm"][j].numpy(), truth[n][2])
                assert torch.equal(named["image"][j], images[n]) and int(named["label"][j]) == labels[b0 + j] and int(named["idx"][j, 0]) == b0 + j
            b0 += n_b
            seen += n_b
    assert seen == 21
    assert len(list(NpyBatcher(names, labels, 3, audio, text, image_fn=lambda n: images[n], drop_last=True, pin=False))) == 2
    # wrong files fail loudly, with the path
    np.save(os.path.join(audio, "bad.npy"), np.zeros((512, 128), np.float32))
    with pytest.raises(MLAHipError, match="bad.npy"):
        load_fbank(audio, "bad")
    np.save(os.path.join(text, "bad_token.npy"), np.zeros((1, 256), np.int32))
    np.save(os.path.join(text, "bad_pm.npy"), np.zeros((1, 256), np.float32))
    with pytest.raises(MLAHipError, match="bad_token.npy"):
        load_token(text, "bad")
    with pytest.raises(MLAHipError, match="missing"):
        load_fbank(audio, "missing")
