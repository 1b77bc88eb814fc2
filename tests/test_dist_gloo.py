"""CPU, world_size 2, gloo: the data-parallel exchange of the MLA step (mla_hip/dist.py).

Each rank runs its shard of a global batch through the oracle's arithmetic (HIP kernels cannot run on
CPU; the oracle is only the stand-in compute here) and exchanges exactly what MLATrainer exchanges, through
the product's `Comm`: encoder gradients as one flat bucketed async all-reduce(SUM), and one packed message
(dW | db | feature column sum | loss) per modality phase.  Result must equal the single-process
DataParallel semantics of the reference (main.py:732): per-replica BatchNorm statistics, head + CE +
projection on the GLOBAL batch, gradients reduce-added.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mla_oracle as O

B_GLOBAL, WORLD, SEED = 4, 2, 17
SPEC_HW, T, IMG_HW = (64, 32), 2, (32, 32)


def _inputs():
    spec = O.portable_normal(SEED, (B_GLOBAL, 1) + SPEC_HW, stream=1, mean=-5.081, std=4.4849)
    image = O.portable_normal(SEED, (B_GLOBAL, 3, T) + IMG_HW, stream=2)
    label = O.portable_labels(SEED, B_GLOBAL, 6)
    return spec, image, label


def _flatten(g, keys):
    return torch.cat([g[k].reshape(-1) for k in keys])


def _phase_local(params, x, modality, feat_fn, bwd_fn, head, label, inv_batch):
    f, cache = O.resnet18_fwd(params, x, modality)
    feat = feat_fn(f)
    logits, loss, dW, db, dX = O.head_ce_fwd_bwd(feat, head["weight"], head["bias"], label)
    scale = feat.shape[0] * inv_batch                       # oracle normalises by the local batch; rescale to global
    grads = O.resnet18_bwd(params, cache, bwd_fn(dX * scale, f.shape))
    return feat, loss * scale, dW * scale, db * scale, grads


def _reference_dataparallel():
    """Single process: shards forwarded separately (per-replica BN), head/CE/projection on the global batch."""
    spec, image, label = _inputs()
    pa, pv = O.make_resnet18_params("audio", SEED), O.make_resnet18_params("visual", SEED + 1)
    head = O.make_head_params(512, 6, SEED + 2)
    Pl = torch.eye(512)
    out = {}
    per = B_GLOBAL // WORLD
    for name, params, x, mod in (("a", pa, spec, "audio"), ("v", pv, image, "visual")):
        feats, caches, shapes = [], [], []
        for r in range(WORLD):
            p_r = {k: v.clone() for k, v in params.items()}
            f, c = O.resnet18_fwd(p_r, x[r * per:(r + 1) * per], mod)
            feats.append(f.mean(dim=(2, 3)) if mod == "audio" else O.av_pool_fwd(f[:per], f, per)[1])
            caches.append((p_r, c)); shapes.append(f.shape)
        feat = torch.cat(feats)
        logits, loss, dW, db, dX = O.head_ce_fwd_bwd(feat, head["weight"], head["bias"], label)
        gsum = None
        for r in range(WORLD):
            d = dX[r * per:(r + 1) * per]
            dout = O.audio_pool_bwd(d, shapes[r]) if mod == "audio" else O.visual_pool_bwd(d, shapes[r], per)
            g = O.resnet18_bwd(caches[r][0], caches[r][1], dout)
            gsum = g if gsum is None else {k: gsum[k] + g[k] for k in g}
        exp = 0 if name == "a" else 1
        Pl, dWp = O.gs_before_update(Pl, feat, dW, 0, 10, exp, "as_intended")
        out[name] = {"loss": loss, "dW": dWp, "db": db, "grads": gsum}
        head = {"weight": O.sgd_step(head["weight"], dWp, None, 1e-3)[0], "bias": O.sgd_step(head["bias"], db, None, 1e-3)[0]}
        # note: momentum buffers of the head are irrelevant for this single-step comparison of gradients
    out["Pl"] = Pl
    return out


def _worker(rank, port, outdir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(2)
    from mla_hip.dist import Comm
    comm = Comm(bucket_bytes=1 << 20)                      # small buckets: several async all-reduces per encoder
    assert comm.world == WORLD and comm.rank == rank
    spec, image, label = _inputs()
    per = B_GLOBAL // WORLD
    sl = slice(rank * per, (rank + 1) * per)
    pa, pv = O.make_resnet18_params("audio", SEED), O.make_resnet18_params("visual", SEED + 1)
    head = O.make_head_params(512, 6, SEED + 2)
    Pl = torch.eye(512)
    inv_batch = 1.0 / B_GLOBAL
    res = {}
    for name, params, x, mod in (("a", pa, spec[sl], "audio"), ("v", pv, image[sl], "visual")):
        feat_fn = (lambda f: f.mean(dim=(2, 3))) if mod == "audio" else (lambda f: O.av_pool_fwd(f[:per], f, per)[1])
        bwd_fn = (lambda d, s: O.audio_pool_bwd(d, s)) if mod == "audio" else (lambda d, s: O.visual_pool_bwd(d, s, per))
        feat, loss, dW, db, grads = _phase_local(params, x, mod, feat_fn, bwd_fn, head, label[sl], inv_batch)
        keys = sorted(grads)
        flat = _flatten(grads, keys)
        works = comm.allreduce_flat_async(flat)            # encoder gradients: async, bucketed, SUM
        head_flat = torch.cat([dW.reshape(-1), db])
        colsum = feat.sum(0) * inv_batch
        loss_t = loss.reshape(1).clone()
        comm.exchange_head(head_flat, colsum, loss_t)      # ONE packed small message
        dW_g, db_g = head_flat[:dW.numel()].view_as(dW), head_flat[dW.numel():]
        exp = 0 if name == "a" else 1
        if exp != 0:                                       # GSPlugin with the pre-reduced global mean
            alpha = O.gs_alpha(0, 10)
            r = colsum.view(1, -1)
            k = Pl @ r.t()
            Pl = Pl - (k @ k.t()) / (alpha + k @ r)
            Pl = Pl / torch.linalg.norm(Pl)
            dW_g = dW_g @ Pl.t()
        comm.wait(works)
        res[name] = {"loss": loss_t[0].clone(), "dW": dW_g.clone(), "db": db_g.clone(),
                     "grads": {k: v for k, v in zip(keys, torch.split(flat, [grads[k].numel() for k in keys]))}}
        head = {"weight": O.sgd_step(head["weight"], dW_g, None, 1e-3)[0], "bias": O.sgd_step(head["bias"], db_g, None, 1e-3)[0]}
    res["Pl"] = Pl
    if rank != 0:
        res = {"a": {"dW": res["a"]["dW"]}, "v": {"dW": res["v"]["dW"]}, "Pl": res["Pl"]}
    torch.save(res, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_two_rank_exchange_equals_dataparallel_semantics(tmp_path):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, str(tmp_path))) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=500)
        assert p.exitcode == 0
    full = torch.load(tmp_path / "rank0.pt", weights_only=True)
    other = torch.load(tmp_path / "rank1.pt", weights_only=True)
    ref = _reference_dataparallel()
    for name in ("a", "v"):
        assert abs(full[name]["loss"].item() - ref[name]["loss"].item()) < 1e-5
        assert torch.allclose(full[name]["dW"], ref[name]["dW"], atol=1e-6, rtol=1e-4)
        assert torch.allclose(full[name]["db"], ref[name]["db"], atol=1e-6, rtol=1e-4)
        assert torch.allclose(other[name]["dW"], full[name]["dW"], atol=0, rtol=0), "ranks must hold identical head gradients"
        for k, g in ref[name]["grads"].items():
            err = (full[name]["grads"][k].view_as(g) - g).norm().item() / max(g.norm().item(), 1e-30)
            assert err < 1e-5, (name, k, err)
    assert torch.allclose(full["Pl"], ref["Pl"], atol=1e-7, rtol=1e-4)
    assert torch.equal(other["Pl"], full["Pl"])


# ---- evaluation under data parallel: --dynamic weights need the GLOBAL batch ------------------------------------
def _eval_worker(rank, port, outdir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from mla_hip.dist import Comm
    comm = Comm()
    B, C = 8, 6
    outs = [O.portable_normal(300 + m, (B, C), stream=1, std=1.5) for m in range(2)]
    label = O.portable_labels(300, B, C)
    per = B // WORLD
    sl = slice(rank * per, (rank + 1) * per)
    g_outs = [comm.allgather_rows(o[sl].contiguous()) for o in outs]             # what Evaluator.update exchanges
    g_label = comm.allgather_rows(label[sl].contiguous())
    w_glob, c_glob = O.valid_batch(g_outs, g_label, C, True, [0.5, 0.5])
    w_loc, _ = O.valid_batch([o[sl] for o in outs], label[sl], C, True, [0.5, 0.5])
    torch.save({"w_glob": torch.tensor(w_glob), "c_glob": c_glob, "w_loc": torch.tensor(w_loc), "g_label": g_label},
               os.path.join(outdir, f"eval{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_dynamic_eval_equals_global_batch(tmp_path):
    """main.py:65-70, 640-646: the entropy weights are computed over the batch axis, so a rank-local evaluation differs from
    the reference's (global-batch, GPU 0) one; after the all-gather both ranks reproduce the single-process result."""
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, port, str(tmp_path))) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=200)
        assert p.exitcode == 0
    B, C = 8, 6
    outs = [O.portable_normal(300 + m, (B, C), stream=1, std=1.5) for m in range(2)]
    label = O.portable_labels(300, B, C)
    w_ref, c_ref = O.valid_batch(outs, label, C, True, [0.5, 0.5])
    r0 = torch.load(tmp_path / "eval0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "eval1.pt", weights_only=True)
    for r in (r0, r1):
        assert torch.equal(r["g_label"], label)
        assert torch.allclose(r["w_glob"], torch.tensor(w_ref), atol=1e-7)
        assert torch.equal(r["c_glob"], c_ref)
    assert not torch.allclose(r0["w_loc"], torch.tensor(w_ref), atol=1e-4), "the rank-local weights differ: the exchange is needed"
