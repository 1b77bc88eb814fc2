"""CPU: the C-ABI library loads and exports every symbol include/mla_hip.h declares (no compute calls),
the ctypes prototype table mirrors the header, and host-side logic that needs no GPU."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "mla_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    decls = re.findall(r"\b(mla_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S)
    return {name: [a.strip() for a in args.split(",")] if args.strip() != "void" else [] for name, args in decls}


def test_library_exports_every_declared_symbol():
    from mla_hip import _lib
    lib = _lib.load()                     # built by __graft_entry__.build(); raises loudly if missing
    fns = header_functions()
    assert len(fns) >= 26
    for name in fns:
        assert hasattr(lib, name), f"libmla_hip.so does not export {name}"
    assert lib.mla_abi_version() == 3
    assert lib.mla_last_error() is not None


def test_ctypes_table_mirrors_header():
    from mla_hip import _lib
    fns = header_functions()
    assert set(fns) == set(_lib.PROTOTYPES), set(fns) ^ set(_lib.PROTOTYPES)
    for name, args in fns.items():
        assert len(args) == len(_lib.PROTOTYPES[name][1]), f"{name}: header has {len(args)} args"


def test_no_torch_types_in_abi():
    txt = open(os.path.join(ROOT, "include", "mla_hip.h")).read()
    assert "torch" not in txt.lower().replace("pytorch", "") or "at::" not in txt
    assert "at::Tensor" not in txt and "#include <torch" not in txt


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "multimodal-learning-with-alternating-unimodal-adaptation_amd")
    for dp, _dn, fn in os.walk(pkg):
        for f in fn:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"


def test_missing_library_fails_loudly(monkeypatch):
    from mla_hip import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmla_hip.so")
    with pytest.raises(_lib.MLAHipError):
        _lib.load()


def test_encoder_layout_and_state_dict_roundtrip_cpu():
    """Flat-buffer layout + reference key names / OIHW conversion (host logic only, CPU tensors)."""
    from mla_hip.encoder import ResNet18Encoder, conv_specs, bn_name_for_conv
    from oracle import mla_oracle as O
    assert conv_specs("audio") == O.resnet18_conv_specs("audio")
    assert conv_specs("visual") == O.resnet18_conv_specs("visual")
    with pytest.raises(NotImplementedError):
        conv_specs("text")                                  # backbone.py:84-85
    assert bn_name_for_conv("layer2.0.downsample.0") == "layer2.0.downsample.1"
    for mod, n_params in (("audio", 11170240), ("visual", 11176512)):     # SURVEY section 6 [probed]
        enc = ResNet18Encoder(mod, device="cpu", seed=0)
        assert enc.numel == n_params
        ref = O.make_resnet18_params(mod, 5)
        enc.load_state_dict(ref)
        sd = enc.state_dict()
        assert list(sd.keys()) == list(ref.keys())          # reference module order
        for k in ref:
            assert torch.equal(sd[k].cpu(), ref[k]), k
        # every segment of the flat buffer is 16-byte aligned (float4 SGD / all-reduce buckets)
        assert all(o % 4 == 0 for o, _ in enc.layout.values())


def test_avclassifier_error_behaviour_cpu():
    from mla_hip import AVClassifier

    class A:
        fusion_method, dataset, gs_flag, modulation = "concat", "CREMA-D", True, "Normal"
    with pytest.raises(NotImplementedError, match="Incorrect dataset name"):       # basic_model.py:26 (Q11 spelling)
        AVClassifier(A(), device="cpu")
    A.dataset, A.fusion_method = "CREMAD", "film"
    with pytest.raises(NotImplementedError, match="Incorrect fusion method"):      # basic_model.py:40
        AVClassifier(A(), device="cpu")
    A.fusion_method = "concat"
    m = AVClassifier(A(), device="cpu", seed=0)
    sd = m.state_dict(prefix="module.")
    assert "module.audio_net.conv1.weight" in sd and "module.fusion_module.fc_out.weight" in sd
    assert sd["module.visual_net.conv1.weight"].shape == (64, 3, 7, 7)
    assert m.module.fusion_module.fc_out.weight.shape == (6, 512)
    m.load_state_dict(sd)


def test_gs_alpha_and_sgd_state_machine_cpu():
    from mla_hip import FusedSGD, GSPlugin
    assert abs(GSPlugin.alpha(0, 10) - 0.1) < 1e-12 and abs(GSPlugin.alpha(5, 10) - 0.1 ** 1.5) < 1e-12

    class G:
        def __init__(self):
            self.flat, self.grad = torch.zeros(8), torch.zeros(8)
    for legacy, want in ((False, "none"), (True, "zero")):
        opt = FusedSGD({"audio": G(), "visual": G(), "head": G()}, legacy_zero_grad=legacy)
        opt.mark_ready("audio")
        opt.zero_grad()
        assert opt.grad_state["audio"] == want and opt.grad_state["visual"] == "none"     # Q6
        opt.drop_grads()
        assert set(opt.grad_state.values()) == {"none"}


def test_bgemm_rejects_descriptors_that_leave_their_buffers():
    """`mla_bgemm` takes raw strides; a descriptor whose largest reachable element lies outside the extent
    the caller vouches for must come back as MLA_ERR_INVALID_ARG *before* any launch (so this runs without a GPU: the
    pointers are never dereferenced).  The round-1 abort of test_attention_pieces[2-12-257-64] was exactly such a bad
    descriptor: ctypes stride arrays built as temporaries and freed before the call read them."""
    import ctypes
    from mla_hip import _lib
    lib = _lib.load()
    L4 = ctypes.c_long * 4
    B, H, n, hd = 2, 12, 257, 64
    D = H * hd
    qkv_elems, p_elems = B * n * 3 * D, B * H * n * n
    fake = 0x1000                                       # non-null, never dereferenced: validation fails first
    good_a, good_b, good_c = L4(n * 3 * D, hd, 3 * D, 1), L4(n * 3 * D, hd, 1, 3 * D), L4(H * n * n, n * n, n, 1)

    def call(sa, sb, sc, ea, eb, ec):
        return lib.mla_bgemm(fake, fake, fake, B, H, n, n, hd, ctypes.addressof(sa), ctypes.addressof(sb), ctypes.addressof(sc),
                             ea, eb, ec, 1.0, None)
    bad_stride = L4(n * 3 * D, hd, 3 * D * 3, 1)                        # row stride three times too large
    assert call(bad_stride, good_b, good_c, qkv_elems, qkv_elems - D, p_elems) == -1
    assert b"operand 0" in lib.mla_last_error()
    assert call(good_a, good_b, good_c, qkv_elems, qkv_elems - D, p_elems - 1) == -1       # C one element short
    assert b"operand 2" in lib.mla_last_error()
    assert call(good_a, L4(n * 3 * D, hd, -1, 3 * D), good_c, qkv_elems, qkv_elems - D, p_elems) == -1    # negative stride


def test_host_side_sizing_and_selection_logic():
    """Host functions of the round-3 kernels that launch nothing: workspaces cover every kernel a shape may be routed to, the folded-BatchNorm
    entry points are offered only for the 64 -> 64 3x3 / 1 / 1 shapes whose grid the persistent patch kernel fills, hooks answer queries."""
    from mla_hip import _lib
    lib = _lib.load()
    # all-taps weight gradient on flat tiles: one full-size [9][Cin][Cout] slab per tile split (256 CUs / block pairs)
    for (N, H, W, C) in ((64, 28, 28, 128), (192, 14, 14, 256), (64, 4, 4, 512), (3, 7, 7, 512)):
        pairs = (C // 64) ** 2
        splits = max(1, min((N * H * W + 63) // 64, 256 // pairs))
        assert lib.mla_conv2d_wgrad_split_ws_bytes(N, H, W, C, C, 3, 3, 1, 1) >= splits * 9 * C * C * 4
    # Linear weight gradient on 192 x 192 tiles: slabs + bias rows per split
    for (M, K, N) in ((16448, 768, 2304), (16448, 3072, 768), (771, 768, 768)):
        pairs = (K // 192) * (N // 192)
        splits = max(1, min((M + 31) // 32, 256 // pairs))
        assert lib.mla_linear_wgrad_split_ws_bytes(M, K, N) >= splits * (K * N + N) * 4
    # folded BatchNorm: layer1 shapes at batch 64 yes; other channel counts / strides / under-filled grids no
    assert lib.mla_conv2d_bnfold_supported(64, 256, 32, 64, 64, 3, 3, 1, 1) == 1
    assert lib.mla_conv2d_bnfold_supported(192, 56, 56, 64, 64, 3, 3, 1, 1) == 1
    assert lib.mla_conv2d_bnfold_supported(2, 56, 56, 64, 64, 3, 3, 1, 1) == 0          # 25 tiles: the gather-GEMM runs such a conv
    assert lib.mla_conv2d_bnfold_supported(64, 128, 16, 128, 128, 3, 3, 1, 1) == 0
    assert lib.mla_conv2d_bnfold_supported(64, 256, 32, 64, 64, 3, 3, 2, 1) == 0
    assert lib.mla_conv2d_bnfold_supported(64, 256, 32, 64, 64, 1, 1, 1, 0) == 0
    for hook in ("mla_conv2d_patch", "mla_conv2d_wgrad_tr", "mla_conv2d_dgrad_merge", "mla_conv2d_two_phase"):
        assert getattr(lib, hook)(-1) in (0, 1, 2)                                       # query only

