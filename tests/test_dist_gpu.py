"""GPU, world_size 2: the REAL MLATrainer data-parallel step (mla_hip/trainer.py + dist.py), two processes sharing the
one GPU of the test box over the `gloo` backend (it accepts device tensors; RCCL refuses two ranks on one device).
Everything but the transport is the production path: per-encoder streams, bucketed asynchronous all-reduce of the flat
encoder gradients, the packed head exchange, GSPlugin on the globally averaged feature, identical head update on every
rank.  Expected result = the reference's DataParallel semantics (main.py:732), computed by the CPU oracle in
tests/test_dist_gloo.py: per-replica BatchNorm statistics, head + CE + projection on the global batch."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import mla_oracle as O  # noqa: E402
from test_dist_gloo import B_GLOBAL, SEED, WORLD, _inputs, _reference_dataparallel  # noqa: E402
from util import assert_close, assert_close_robust  # noqa: E402


def _worker(rank, port, outdir, conv_math):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.cuda.set_device(0)
    from mla_hip import AVClassifier, Comm, MLATrainer

    class Args:
        fusion_method, dataset, gs_flag, modulation = "concat", "CREMAD", True, "Normal"

    model = AVClassifier(Args(), seed=0, conv_math=conv_math)
    pa, pv = O.make_resnet18_params("audio", SEED), O.make_resnet18_params("visual", SEED + 1)
    hd = O.make_head_params(512, 6, SEED + 2)
    sd = {f"audio_net.{k}": v for k, v in pa.items()}
    sd.update({f"visual_net.{k}": v for k, v in pv.items()})
    sd.update({f"fusion_module.fc_out.{k}": v for k, v in hd.items()})
    model.load_state_dict(sd)
    comm = Comm(bucket_bytes=1 << 20)                       # several async buckets per encoder
    assert comm.active and comm.world == WORLD
    tr = MLATrainer(model, lr=1e-3, momentum=0.9, weight_decay=1e-4, gs_mode="as_intended", comm=comm)
    tr.keep_debug = True
    spec, image, label = _inputs()
    per = B_GLOBAL // WORLD
    sl = slice(rank * per, (rank + 1) * per)
    losses = tr.train_step(spec[sl].cuda(), image[sl].cuda(), label[sl].cuda(), 0, 10)
    torch.cuda.synchronize()
    res = {"loss_a": losses["loss_a"].cpu(), "loss_v": losses["loss_v"].cpu(),
           "dW_a": tr.last["head_grad_a"].cpu(), "dW_v": model.fusion_module.fc_out.weight_grad.cpu().clone(),
           "db_v": model.fusion_module.fc_out.bias_grad.cpu().clone() if hasattr(model.fusion_module.fc_out, "bias_grad") else None,
           "Pl": tr.gs_plugin.Pl.cpu(), "head_w": model.fusion_module.fc_out.weight.cpu().clone()}
    if rank == 0:
        res["grads_a"] = {k: v.cpu() for k, v in model.audio_net.grads_as_reference().items()}
        res["grads_v"] = {k: v.cpu() for k, v in model.visual_net.grads_as_reference().items()}
    torch.save(res, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(900)
@pytest.mark.parametrize("conv_math", ["f32", "split"])
def test_two_rank_trainer_equals_dataparallel_semantics(tmp_path, conv_math):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, str(tmp_path), conv_math)) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    ref = _reference_dataparallel()
    assert_close(r0["loss_a"].reshape(()), ref["a"]["loss"], atol=2e-4, name="global loss a")
    assert_close(r0["loss_v"].reshape(()), ref["v"]["loss"], atol=2e-4, name="global loss v")
    assert_close(r0["dW_a"], ref["a"]["dW"], atol=2e-4, name="head grad a (global batch)")
    assert_close(r0["dW_v"], ref["v"]["dW"], atol=2e-4, name="projected head grad v (global batch)")
    assert_close(r0["Pl"], ref["Pl"], atol=1e-6, rtol=1e-4, name="Pl")
    for k in ("dW_a", "dW_v", "Pl", "head_w", "loss_a", "loss_v"):
        assert torch.equal(r0[k], r1[k]), f"ranks must hold identical {k}"
    for enc in ("a", "v"):
        for k, g in ref[enc]["grads"].items():
            assert_close_robust(r0["grads_" + enc][k], g, rel_l2=5e-2, elem_tol=1.0, frac=0.0, name=f"reduced grad {enc}.{k}")


# ---- transformer trainers (configs[3] M3AE text+image, configs[4] Modal3 audio+image+text) -------------------------------
# No BatchNorm in these encoders (LayerNorm is per token), so sharding the batch over ranks is exactly the single-process
# global-batch step up to summation order: the 2-rank result must equal ONE process running the whole batch.
T_DEPTH, T_VOCAB, T_B = 2, 200, 4


def _transformer_case(which, seed=211):
    """(model factory, state_dict, inputs tuple for train_step without label, label, n_classes)"""
    import numpy as np
    if which == "m3ae":
        class A:
            fusion_method, dataset, gs_flag, modulation = "concat", "MVSA", True, "Normal"
        C, encs = 3, (("mae_a", O.make_m3ae_params(seed, depth=T_DEPTH, vocab=T_VOCAB)), ("mae_v", O.make_m3ae_params(seed + 1, depth=T_DEPTH, vocab=T_VOCAB)))
    else:
        class A:
            fusion_method, dataset, gs_flag, modulation = "concat", "IEMOCAP", True, "Normal"
        C, encs = 4, (("mae_a", O.make_cavmae_audio_params(seed, depth=T_DEPTH)), ("mae_v", O.make_m3ae_params(seed + 1, depth=T_DEPTH, vocab=T_VOCAB)),
                      ("mae_t", O.make_m3ae_params(seed + 2, depth=T_DEPTH, vocab=T_VOCAB)))
    sd = {f"{nm}.{k}": v for nm, p in encs for k, v in p.items()}
    sd.update({f"fusion_module.fc_out.{k}": v for k, v in O.make_head_params(768, C, seed + 3).items()})
    token = torch.from_numpy(np.minimum((O.portable_uniform(seed, T_B * 256, 7) * T_VOCAB).astype(np.int64), T_VOCAB - 1)).view(T_B, 1, 256)
    pm = torch.zeros(T_B, 1, 256)
    for b in range(T_B):
        pm[b, 0, 30 + 41 * b:] = 1.0
    image = O.portable_normal(seed, (T_B, 3, 256, 256), stream=3)
    spec = O.portable_normal(seed, (T_B, 1024, 128), stream=4, mean=-5.081, std=4.4849)
    label = O.portable_labels(seed, T_B, C)
    inputs = (token, pm, image) if which == "m3ae" else (token, pm, image, spec)
    return A, sd, inputs, label


def _build_transformer(which, comm=None):
    from mla_hip import M3AEClassifier, MLATrainer, Modal3Classifier
    A, sd, inputs, label = _transformer_case(which)
    cls = M3AEClassifier if which == "m3ae" else Modal3Classifier
    model = cls(A(), depth=T_DEPTH, text_vocab_size=T_VOCAB, seed=0)
    model.load_state_dict(sd)
    tr = MLATrainer(model, gs_mode="as_intended", comm=comm)
    tr.keep_debug = True
    return model, tr, inputs, label


def _collect(model, tr, losses):
    res = {k: v.cpu().clone() for k, v in losses.items()}
    for tag, _g, enc in model.mla_encoders():
        res["raw_" + tag] = tr.last[f"head_grad_{tag}_raw"].cpu()
        res["proj_" + tag] = tr.last[f"head_grad_{tag}"].cpu()
        res["feat_" + tag] = tr.last[tag].cpu().clone()
        res["grads_" + tag] = {k: v.cpu() for k, v in enc.grads_as_reference().items()}
    res["Pl"], res["head"] = tr.gs_plugin.Pl.cpu(), model.fusion_module.fc_out.flat.cpu().clone()
    return res


def _worker_transformer(rank, port, outdir, which):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.cuda.set_device(0)
    from mla_hip import Comm
    comm = Comm(bucket_bytes=4 << 20)
    model, tr, inputs, label = _build_transformer(which, comm)
    per = T_B // WORLD
    sl = slice(rank * per, (rank + 1) * per)
    losses = tr.train_step(*[x[sl].cuda() for x in inputs], label[sl].cuda(), 0, 10)
    tr.join()
    torch.cuda.synchronize()
    res = _collect(model, tr, losses)
    res["feat_rows"] = (rank * per, (rank + 1) * per)
    torch.save(res, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("which", ["m3ae", "modal3"])
def test_two_rank_transformer_trainers_equal_global_batch(tmp_path, which):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker_transformer, args=(r, port, str(tmp_path), which)) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    model, tr, inputs, label = _build_transformer(which)                       # one process, whole batch
    losses = tr.train_step(*[x.cuda() for x in inputs], label.cuda(), 0, 10)
    tr.join()
    torch.cuda.synchronize()
    ref = _collect(model, tr, losses)
    tags = [t for t, _g, _e in model.mla_encoders()]
    for k in ["Pl", "head", "loss"] + ["loss_" + t for t in tags] + ["raw_" + t for t in tags] + ["proj_" + t for t in tags]:
        assert torch.equal(r0[k], r1[k]), f"ranks must hold identical {k}"
    for tag in tags:
        assert_close(r0["loss_" + tag].reshape(()), ref["loss_" + tag].reshape(()), atol=2e-5, name=f"global loss {tag}")
        assert_close(r0["raw_" + tag], ref["raw_" + tag], atol=1e-5, name=f"raw head grad {tag} (global batch)")
        lo, hi = r1["feat_rows"]
        assert_close(r1["feat_" + tag], ref["feat_" + tag][lo:hi], atol=2e-5, name=f"rank-1 features {tag}")
        for k, g in ref["grads_" + tag].items():
            assert_close_robust(r0["grads_" + tag][k], g, rel_l2=2e-4, elem_tol=1.0, frac=0.0, name=f"reduced grad {tag}.{k}")
    # first modality: the projection has not fired yet (Q5), so its "projected" gradient is the raw one -> head update equal
    assert_close(r0["proj_" + tags[0]], ref["proj_" + tags[0]], atol=1e-5, name="first-phase head gradient")


# ---- evaluation: --dynamic fusion on the global batch (Q9), real Evaluator on two ranks ---------------------------------------
def _eval_worker(rank, port, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.cuda.set_device(0)
    from mla_hip import Comm, Evaluator
    from test_eval_gpu import _model
    model = _model(71)
    ev = Evaluator(model, dynamic=True, comm=Comm())
    spec, image, label = _eval_inputs()
    per = spec.shape[0] // WORLD
    sl = slice(rank * per, (rank + 1) * per)
    ev.update(spec[sl].cuda(), image[sl].cuda(), label[sl].cuda())
    torch.cuda.synchronize()
    torch.save({"counts": ev.counts.cpu(), "weights": ev.weights.cpu()}, os.path.join(outdir, f"ev{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _eval_inputs(B=8):
    spec = O.portable_normal(80, (B, 128, 64), stream=1, mean=-5.081, std=4.4849)
    image = O.portable_normal(80, (B, 3, 2, 96, 96), stream=2)
    return spec, image, O.portable_labels(80, B, 6)


@pytest.mark.timeout(600)
def test_two_rank_dynamic_eval_equals_single_process_global_batch(tmp_path):
    """Eval-mode BatchNorm uses running statistics, so the logits of a sample do not depend on its shard: two ranks that
    all-gather their logits must reproduce the single-process evaluation of the whole batch exactly (counters) / to 1e-6
    (entropy weights: one scalar per modality over the GLOBAL batch, main.py:65-70)."""
    from mla_hip import Evaluator
    from test_eval_gpu import _model
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, port, str(tmp_path))) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=400)
        assert p.exitcode == 0
    spec, image, label = _eval_inputs()
    ev = Evaluator(_model(71), dynamic=True)
    ev.update(spec.cuda(), image.cuda(), label.cuda())
    torch.cuda.synchronize()
    r0 = torch.load(tmp_path / "ev0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "ev1.pt", weights_only=True)
    for r in (r0, r1):
        assert torch.equal(r["counts"], ev.counts.cpu())
        assert_close(r["weights"], ev.weights, atol=1e-6, name="global-batch entropy weights")
    assert int(r0["counts"][:6].sum()) == spec.shape[0]                       # every rank counted the whole batch


# ---- the PROTOCOL path under data parallel: mla_hip.DataParallel + autograd Functions + FusedSGD + GSPlugin ---------------------
def _protocol_steps(model, optimizer, gs_plugin, criterion, inputs, label, steps=2):
    """main.py:426-454, 468-470 (two modalities, --lorb m3ae) for `steps` batches; returns the last losses."""
    token, padding_mask, image = inputs
    out = None
    for batch_step in range(steps):
        optimizer.zero_grad()
        a, v = model(token, padding_mask, image)
        out_a = model.module.fusion_module.fc_out(a)
        loss_a = criterion(out_a, label)
        loss_a.backward()
        gs_plugin.before_update(model.module.fusion_module.fc_out, a, batch_step, 10, gs_plugin.exp_count)
        optimizer.step()
        optimizer.zero_grad()
        gs_plugin.exp_count += 1
        out_v = model.module.fusion_module.fc_out(v)
        loss_v = criterion(out_v, label)
        loss_v.backward()
        gs_plugin.before_update(model.module.fusion_module.fc_out, v, batch_step, 10, gs_plugin.exp_count)
        optimizer.step()
        optimizer.zero_grad()
        gs_plugin.exp_count += 1
        for n, p in model.named_parameters():
            if p.grad != None:
                del p.grad
        out = (loss_a.detach().clone(), loss_v.detach().clone())
    return out


def _build_protocol_m3ae(comm=None):
    import mla_hip
    A, sd, inputs, label = _transformer_case("m3ae")
    model = mla_hip.M3AEClassifier(A(), depth=T_DEPTH, text_vocab_size=T_VOCAB, seed=0)
    model.load_state_dict(sd)
    model = mla_hip.DataParallel(model, comm=comm)
    optimizer = mla_hip.FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    return model, optimizer, mla_hip.GSPlugin(mode="as_published"), mla_hip.CrossEntropyLoss(), inputs, label


def _worker_protocol(rank, port, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.cuda.set_device(0)
    from mla_hip import Comm
    model, optimizer, gs, crit, inputs, label = _build_protocol_m3ae(Comm(bucket_bytes=4 << 20))
    per = T_B // WORLD
    sl = slice(rank * per, (rank + 1) * per)
    model.train()
    la, lv = _protocol_steps(model, optimizer, gs, crit, [x[sl].cuda() for x in inputs], label[sl].cuda())
    torch.cuda.synchronize()
    m = model.module
    torch.save({"head": m.fusion_module.fc_out.flat.cpu(), "text": m.mae_a.flat.cpu(), "image": m.mae_v.flat.cpu(),
                "loss": torch.stack([la, lv]).cpu()}, os.path.join(outdir, f"proto{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_protocol_path_equals_global_batch(tmp_path):
    """mla_hip.DataParallel on the object protocol: d logits pre-scaled by 1/world and the packed dW|db all-reduced in
    HeadLinear.backward, the flat encoder gradient all-reduced asynchronously in EncoderFeature.backward and awaited by
    FusedSGD.step().  Two ranks (two steps, momentum included) must end with the parameters of ONE process that runs the
    same loop on the whole batch (transformer encoders: no per-rank statistics); the mean of the ranks' local losses is the
    global loss.  Projection off (as_published): its conditioning on transformer features is tested elsewhere."""
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker_protocol, args=(r, port, str(tmp_path))) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    r0 = torch.load(tmp_path / "proto0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "proto1.pt", weights_only=True)
    model, optimizer, gs, crit, inputs, label = _build_protocol_m3ae()
    model.train()
    la, lv = _protocol_steps(model, optimizer, gs, crit, [x.cuda() for x in inputs], label.cuda())
    torch.cuda.synchronize()
    m = model.module
    for k in ("head", "text", "image"):
        assert torch.equal(r0[k], r1[k]), f"ranks must hold identical {k} parameters"
    assert_close(r0["head"], m.fusion_module.fc_out.flat, atol=1e-6, name="head after 2 data-parallel steps")
    assert_close(r0["image"], m.mae_v.flat, atol=1e-6, name="image encoder after 2 data-parallel steps")
    assert_close(r0["text"], m.mae_a.flat, atol=1e-6, name="text encoder after 2 data-parallel steps")
    assert_close((r0["loss"] + r1["loss"]) / 2, torch.stack([la, lv]), atol=1e-5, name="mean of rank losses = global loss")
