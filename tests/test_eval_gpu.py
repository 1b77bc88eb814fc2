"""Evaluation row: eval-mode encoders + shared head + fusion + accuracy, HIP vs the reference's
golden vectors (reference AVClassifier in eval() mode) and vs the oracle's restatement of main.py:65-106, 640-676.
Tolerance: features / logits 2e-4 absolute + 1e-5 relative (values reach 240 with the random running statistics);
fusion weights 1e-5; counters exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mla_oracle as O  # noqa: E402
from util import assert_close  # noqa: E402


class _Args:
    fusion_method, dataset, gs_flag, modulation = "concat", "CREMAD", True, "Normal"


def _model(seed, conv_math="f32"):
    from mla_hip import AVClassifier
    model = AVClassifier(_Args(), seed=0, conv_math=conv_math)
    pa, pv = O.make_resnet18_params("audio", seed), O.make_resnet18_params("visual", seed + 1)
    hd = O.make_head_params(512, 6, seed + 2)
    for params, off in ((pa, 0), (pv, 500)):
        for si, k in enumerate(sorted(k for k in params if k.endswith("running_mean"))):
            params[k] = O.portable_normal(seed, tuple(params[k].shape), stream=4000 + off + si, std=0.3)
            kv = k.replace("running_mean", "running_var")
            params[kv] = O.portable_normal(seed, tuple(params[kv].shape), stream=4250 + off + si, std=0.2).abs() + 0.5
    sd = {f"audio_net.{k}": v for k, v in pa.items()}
    sd.update({f"visual_net.{k}": v for k, v in pv.items()})
    sd.update({f"fusion_module.fc_out.{k}": v for k, v in hd.items()})
    model.load_state_dict(sd)
    return model


@pytest.mark.parametrize("conv_math", ["f32", "split"])
@pytest.mark.parametrize("dynamic", [True, False])
def test_valid_vs_reference_golden(dynamic, conv_math, golden_dir):
    from mla_hip import Evaluator
    fx = np.load(os.path.join(golden_dir, "eval_small.npz"))
    B, sh, sw, T, ih, iw, seed = [int(v) for v in fx["meta"]]
    model = _model(seed, conv_math)
    ev = Evaluator(model, dynamic=dynamic, av_alpha=0.5)
    spec = O.portable_normal(seed + 9, (B, sh, sw), stream=1, mean=-5.081, std=4.4849)
    image = O.portable_normal(seed + 9, (B, 3, T, ih, iw), stream=2)
    label = O.portable_labels(seed + 9, B, 6)
    outs = ev.update(spec.cuda(), image.cuda(), label.cuda())
    torch.cuda.synchronize()
    assert_close(model._feat_buffers(B)["a"], fx["a"], atol=2e-4, rtol=1e-5, name="eval feature a")   # |a| up to 240 here
    assert_close(model._feat_buffers(B)["v"], fx["v"], atol=2e-4, rtol=1e-5, name="eval feature v")
    assert_close(outs[0], fx["out_a"], atol=2e-4, rtol=1e-5, name="eval logits a")
    assert_close(outs[1], fx["out_v"], atol=2e-4, rtol=1e-5, name="eval logits v")
    tag = "dynamic" if dynamic else "fixed"
    assert_close(ev.weights[:2], fx[f"{tag}.weights"], atol=1e-5, name="fusion weights")
    got = ev.counts.view(4, 6).cpu().numpy()
    assert (got == fx[f"{tag}.counts"]).all(), (got, fx[f"{tag}.counts"])
    acc = ev.result()
    want = fx[f"{tag}.counts"].sum(1)
    assert acc == tuple(want[1:] / want[0])
    # second batch accumulates; training afterwards flips the BatchNorm layers back to batch statistics
    ev.update(spec.cuda(), image.cuda(), label.cuda())
    assert int(ev.counts.view(4, 6)[0].sum()) == 2 * B
    model.train()
    assert model.audio_net.training and model.visual_net.training


def test_fusion_three_modalities_vs_oracle():
    from mla_hip import ops
    B, C = 32, 4
    outs = [O.portable_normal(5 + m, (B, C), stream=1, std=1.5) for m in range(3)]
    label = O.portable_labels(5, B, C)
    for dynamic in (True, False):
        w_ref, c_ref = O.valid_batch(outs, label, C, dynamic, [0.35, 0.25, 0.4])       # main.py:488 defaults
        counts = torch.zeros(C * 5, device="cuda", dtype=torch.int32)
        w = torch.zeros(3, device="cuda")
        ops.eval_fuse([o.cuda() for o in outs], label.cuda(), counts, w, dynamic, [0.35, 0.25, 0.4])
        assert_close(w, torch.tensor(w_ref), atol=1e-5, name="weights (3 modalities)")
        assert (counts.view(5, C).cpu() == c_ref.int()).all()
