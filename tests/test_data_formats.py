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


class _FakeEvent:
    def __init__(self, log, tag):
        self.log, self.tag = log, tag

    def synchronize(self):
        self.log.append(self.tag)


def test_staging_ring_is_refilled_only_after_its_copy_completed(tmp_path):
    """a staging tuple may be overwritten only after the copies out of it COMPLETED (not merely were issued).
    The consumer reports one event per batch through `copied()`; the refill of a slot must synchronize on that slot's event
    first, in order, and a consumer that never calls copied() must still work."""
    from mla_hip import NpyBatcher
    names = [f"c{i:02d}" for i in range(11)]
    audio, text, truth = _write_reference_style(str(tmp_path), names)
    img = torch.zeros(3, 1, 4, 4)
    nb = NpyBatcher(names, [0] * 11, 1, audio, text, image_fn=lambda n: img, order="av", ring=3, pin=False)
    log, seen = [], []
    for i, batch in enumerate(nb):
        assert np.array_equal(batch[0][0].numpy(), truth[names[i]][0])
        seen.append((i, list(log)))
        nb.copied(_FakeEvent(log, i))            # "the copies of batch i complete when this event does"
    # batch i (i >= ring) reuses the staging tuple of batch i - ring: exactly that event must have been waited for before
    for i, waited in seen:
        assert waited == list(range(0, max(0, i - 3 + 1))), (i, waited)
    assert len(list(NpyBatcher(names, [0] * 11, 4, audio, text, image_fn=lambda n: img, ring=2, pin=False))) == 3   # no copied(): unfenced


@pytest.mark.gpu
def test_npy_batcher_pinned_ring_through_device_feeder_without_host_sync(tmp_path):
    """The documented NpyBatcher -> DeviceFeeder path with pinned staging, more batches than ring + depth, the copy stream held
    back behind a long device-side delay and no host synchronisation by the consumer: every batch must arrive intact."""
    from mla_hip import DeviceFeeder, NpyBatcher
    names = [f"c{i:02d}" for i in range(12)]
    audio, text, truth = _write_reference_style(str(tmp_path), names)
    imgs = {n: torch.full((3, 1, 4, 4), float(i)) for i, n in enumerate(names)}
    nb = NpyBatcher(names, list(range(12)), 1, audio, text, image_fn=lambda n: imgs[n], order="av", ring=2, pin=True)
    feeder = DeviceFeeder(nb, depth=3)
    with torch.cuda.stream(feeder.copy_stream):
        torch.cuda._sleep(int(2e9))              # ~1 s: the host is far ahead of the DMA engine unless the ring is fenced
    got = []
    for spec, image, label, idx in feeder:
        got.append((spec.clone(), image.clone(), label.clone()))       # stream-ordered reads only
    torch.cuda.synchronize()
    assert len(got) == 12
    for i, (spec, image, label) in enumerate(got):
        assert np.array_equal(spec[0].cpu().numpy(), truth[names[i]][0]), f"batch {i}: spectrogram overwritten in the staging ring"
        assert torch.equal(image[0].cpu(), imgs[names[i]]) and int(label[0]) == i
