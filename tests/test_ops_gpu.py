"""Per-kernel parity: HIP (through the C ABI) vs the CPU oracle, same seeded inputs.

fp32 everywhere.  Tolerances: conv contractions 2e-5 relative to max|ref| (fp32 MFMA is an exact
k-ordered fmaf chain; only the summation order differs from ATen's), BN/pool/head/GS/SGD 1e-5.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mla_oracle as O  # noqa: E402
from util import assert_close  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    from mla_hip import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _ops


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def hwio(w):
    return w.permute(2, 3, 1, 0).contiguous()


def oihw(w):
    return w.permute(3, 2, 0, 1).contiguous()


# every conv configuration of ResNet-18 (backbone.py) at reduced spatial size, plus ragged sizes
CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad
    (2, 40, 24, 1, 64, 7, 2, 3),      # audio stem
    (3, 36, 36, 3, 64, 7, 2, 3),      # visual stem
    (2, 20, 12, 64, 64, 3, 1, 1),     # layer1
    (2, 20, 12, 64, 128, 3, 2, 1),    # layer2.0.conv1
    (2, 20, 12, 64, 128, 1, 2, 0),    # layer2.0.downsample
    (2, 10, 6, 128, 128, 3, 1, 1),
    (3, 14, 14, 128, 256, 3, 2, 1),   # 14 -> 7 (odd output)
    (3, 14, 14, 128, 256, 1, 2, 0),
    (3, 7, 7, 256, 256, 3, 1, 1),
    (3, 7, 7, 256, 512, 3, 2, 1),     # odd input
    (3, 7, 7, 256, 512, 1, 2, 0),
    (2, 4, 4, 512, 512, 3, 1, 1),
    (1, 5, 3, 64, 64, 3, 1, 1),       # tiny / ragged: M=15 << tile
    (5, 9, 11, 64, 128, 3, 2, 1),     # odd everything
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_fwd_dgrad_wgrad(ops, case):
    N, H, W, Cin, Cout, k, s, p = case
    seed = sum(case)
    x = O.portable_normal(seed, (N, Cin, H, W), stream=1)
    w = O.portable_normal(seed, (Cout, Cin, k, k), stream=2, std=math.sqrt(2.0 / (Cin * k * k)))
    y_ref = O.conv2d_fwd(x, w, s, p)
    dy = O.portable_normal(seed, tuple(y_ref.shape), stream=3)
    xd, wd, dyd = nhwc(x).cuda(), hwio(w).cuda(), nhwc(dy).cuda()

    # forward + fused BN partial statistics
    part = torch.zeros(ops.conv2d_fwd_partial_elems(N, H, W, Cin, Cout, k, k, s, p), device="cuda")
    y, tiles = ops.conv2d_fwd(xd, wd, s, p, bn_partial=part)
    assert_close(nchw(y.cpu()), y_ref, atol=0, rtol=2e-5, name="conv fwd")
    pt = part.view(torch.float64)[:tiles * 2 * Cout].view(tiles, 2, Cout).sum(0).cpu()    # fp64 partials; upper-bound allocation
    assert_close(pt[0], y_ref.double().sum(dim=(0, 2, 3)), atol=1e-3, rtol=2e-5, name="fused colsum")
    assert_close(pt[1], (y_ref.double() ** 2).sum(dim=(0, 2, 3)), atol=1e-3, rtol=2e-5, name="fused colsumsq")

    # weight gradient
    dw_ref = O.conv2d_wgrad(x, dy, w.shape, s, p)
    ws = torch.empty(ops.conv2d_wgrad_ws_bytes(N, H, W, Cin, Cout, k, k, s, p) // 4 + 4, device="cuda")
    dw = torch.empty_like(wd)
    ops.conv2d_wgrad(xd, dyd, dw, s, p, ws)
    assert_close(oihw(dw.cpu()), dw_ref, atol=0, rtol=2e-5, name="conv wgrad")

    # input gradient (+ residual + relu mask epilogue); the stem needs none
    if Cin % 64 == 0:
        dx_ref = O.conv2d_dgrad(dy, w, x.shape, s, p)
        wt_ws = torch.empty(w.numel(), device="cuda")
        dx = ops.conv2d_dgrad(dyd, wd, (N, H, W, Cin), s, p, wt_ws)
        assert_close(nchw(dx.cpu()), dx_ref, atol=0, rtol=2e-5, name="conv dgrad")
        res = O.portable_normal(seed, (N, Cin, H, W), stream=4)
        msk = O.portable_normal(seed, (N, Cin, H, W), stream=5)
        dx2 = torch.empty_like(dx)
        ops.conv2d_dgrad(dyd, wd, (N, H, W, Cin), s, p, wt_ws, dx=dx2, residual=nhwc(res).cuda(), relu_src=nhwc(msk).cuda())
        assert_close(nchw(dx2.cpu()), (dx_ref + res) * (msk > 0), atol=0, rtol=2e-5, name="conv dgrad+res+mask")
        # accumulate in place (residual aliases dx)
        ops.conv2d_dgrad(dyd, wd, (N, H, W, Cin), s, p, wt_ws, dx=dx, residual=dx)
        assert_close(nchw(dx.cpu()), 2 * dx_ref, atol=0, rtol=2e-5, name="conv dgrad accumulate")


SPLIT_CASES = [c for c in CONV_CASES if c[3] % 64 == 0] + [(2, 33, 17, 128, 128, 3, 1, 1)]   # M = 1122: ragged vs every tile


WGRAD_TR_CASES = [
    # N, H, W, Cin, Cout.  64 -> 64: 8 x 8 spatial tiles (wgrad_tr_split_kernel)
    (2, 20, 12, 64, 64), (3, 16, 8, 64, 64), (1, 8, 8, 64, 64), (5, 9, 11, 64, 64), (2, 56, 56, 64, 64), (64, 56, 24, 64, 64),
    # wider layers on maps up to 28 pixels wide: 64-pixel flat tiles, one workgroup per 64 x 64 channel block pair and tile split
    # (wgrad_flat_tr_kernel): tap validity by per-pixel mask bits, tiles that span image rows and whole images
    (2, 28, 28, 128, 128),        # layer2 maps: 1568 pixels = 24.5 tiles, 4 block pairs
    (3, 14, 14, 256, 256),        # 588 pixels: ragged last tile, a tile spans 4.6 image rows
    (5, 7, 7, 512, 512),          # 7 x 7 maps: a tile holds 1.3 images; 64 block pairs x 4 splits
    (2, 16, 8, 128, 256),         # Cin != Cout, audio-like
    (1, 5, 3, 128, 128),          # smaller than one tile
    (4, 32, 4, 512, 512),         # audio layer4: 4 pixels wide (every pixel touches a left / right border in some tap)
    (3, 3, 1, 128, 128),          # one pixel wide
    (64, 28, 28, 128, 128),       # 784 tiles: ~12 per workgroup
    (48, 14, 14, 256, 256),
]


@pytest.mark.parametrize("case", WGRAD_TR_CASES, ids=lambda c: "x".join(map(str, c)))
def test_wgrad_all_taps_tr_kernel(ops, case):
    """Weight gradient of the 3x3 / 1 / 1 convolutions by the all-taps kernels (wgrad_tr_split.hip: transposing LDS reads, every
    tap from one staged patch) vs the oracle / the per-tap kernel at the conv tolerance, ragged tiles included, bitwise
    reproducible."""
    N, H, W, Cin, Cout = case
    seed = sum(case) + 7
    big = N * H * W * max(Cin, Cout) > 1_300_000
    x = O.portable_normal(seed, (N, Cin, H, W), stream=1)
    dy = O.portable_normal(seed, (N, Cout, H, W), stream=3)
    xd, dyd = nhwc(x).cuda(), nhwc(dy).cuda()
    ws = torch.empty(ops.conv2d_wgrad_split_ws_bytes(N, H, W, Cin, Cout, 3, 3, 1, 1) // 4 + 4, device="cuda")
    assert ops.conv2d_wgrad_tr() == 1
    dw = torch.empty((3, 3, Cin, Cout), device="cuda")
    ops.conv2d_wgrad_split(xd, dyd, dw, 1, 1, ws)
    dw_b = torch.empty_like(dw)
    ops.conv2d_wgrad_split(xd, dyd, dw_b, 1, 1, ws)
    ops.conv2d_wgrad_tr(0)
    try:
        dw_old = torch.empty_like(dw)
        ops.conv2d_wgrad_split(xd, dyd, dw_old, 1, 1, ws)
    finally:
        ops.conv2d_wgrad_tr(1)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw_b), "bitwise reproducible"
    ref = oihw(dw_old.cpu()) if big else O.conv2d_wgrad(x, dy, (Cout, Cin, 3, 3), 1, 1)
    if not big:
        assert_close(oihw(dw_old.cpu()), ref, atol=0, rtol=2e-5, name="per-tap split wgrad")
    assert_close(oihw(dw.cpu()), ref, atol=0, rtol=2e-5, name="all-taps split wgrad")
    if N * H * W > 64:      # (a single k-chunk can legitimately give the same bits)
        assert not torch.equal(dw, dw_old), "the all-taps kernel did not run (same bits as the per-tap kernel)"


PATCH_CASES = [
    # N, H, W, Cin, Cout: 3x3 / stride 1 / pad 1 (every BasicBlock conv but the three stride-2 ones, backbone.py:28, 31)
    (2, 20, 12, 64, 64),      # layer1 shape class: BN = 64, two taps per K stage; M = 480: one full + one ragged 256-pixel tile
    (2, 10, 6, 128, 128),     # BN = 128
    (3, 7, 7, 256, 256),      # 7 x 7 maps: a tile spans five images (halo rows belong to other images)
    (2, 4, 4, 512, 512),      # 16 chunks
    (5, 32, 4, 64, 64),       # audio layer4-like narrow maps
    (1, 5, 3, 64, 128),       # smaller than one tile, Cin != Cout
    (3, 56, 56, 64, 64),      # visual layer1 rows: the widest patch (8 rows x 58 pixels)
    (40, 14, 14, 256, 256),   # many tiles: the grid exceeds the CU count
    (25, 56, 56, 64, 64),     # 64 -> 64 persistent kernel: 307 tiles on 256 workgroups (two tiles for some: the epilogue of the first rides
                              # in the MFMA stages of the second), ragged last tile
    (40, 64, 32, 64, 64),     # ... 320 whole tiles, audio-like maps
]


@pytest.mark.parametrize("case", PATCH_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_patch_fwd_dgrad(ops, case):
    """Forward (+ fused fp64 BatchNorm statistics) and input gradient (+ residual, ReLU mask) of the 3x3 / 1 / 1 convolutions
    by the LDS-patch kernel (conv_patch_split.hip) vs the oracle / the per-tap gather-GEMM at the conv tolerance."""
    N, H, W, Cin, Cout = case
    seed = sum(case) + 3
    big = N * H * W * max(Cin, Cout) > 1_500_000
    x = O.portable_normal(seed, (N, Cin, H, W), stream=1)
    w = O.portable_normal(seed, (Cout, Cin, 3, 3), stream=2, std=math.sqrt(2.0 / (Cin * 9)))
    xd, wd = nhwc(x).cuda(), hwio(w).cuda()
    wT, wS = ops.conv2d_wsplit(wd, True), ops.conv2d_wsplit(wd, False)
    default = ops.conv2d_patch()
    assert default in (0, 1, 2)
    part = torch.zeros(ops.conv2d_fwd_partial_elems(N, H, W, Cin, Cout, 3, 3, 1, 1), device="cuda")
    ops.conv2d_patch(2)                       # the patch kernel wherever the geometry allows (by default only where its grid fills the chip)
    try:
        y, tiles = ops.conv2d_fwd_split(xd, wT, wd.shape, 1, 1, bn_partial=part)
        y2, _ = ops.conv2d_fwd_split(xd, wT, wd.shape, 1, 1)
        ops.conv2d_patch(0)
        y_old, _ = ops.conv2d_fwd_split(xd, wT, wd.shape, 1, 1)
    finally:
        ops.conv2d_patch(default)
    torch.cuda.synchronize()
    ntiles = (N * H * W + 255) // 256
    # statistics rows: one per 256-pixel tile; the persistent 64 -> 64 kernel keeps them per workgroup (at most one per CU)
    assert tiles == (min(ntiles, 256) if (Cin == 64 and Cout == 64 and ntiles >= 2) else ntiles) and torch.equal(y, y2)
    y_ref = nchw(y_old.cpu()) if big else O.conv2d_fwd(x, w, 1, 1)
    if not big:
        assert_close(nchw(y_old.cpu()), y_ref, atol=0, rtol=2e-5, name="per-tap fwd")
    assert_close(nchw(y.cpu()), y_ref, atol=0, rtol=2e-5, name="patch fwd")
    pt = part.view(torch.float64)[:tiles * 2 * Cout].view(tiles, 2, Cout).sum(0).cpu()
    yd = y.double().cpu().reshape(-1, Cout)
    assert_close(pt[0], yd.sum(0), atol=1e-6, rtol=1e-9, name="fused colsum == sums of the stored y")
    assert_close(pt[1], (yd ** 2).sum(0), atol=1e-6, rtol=1e-9, name="fused colsumsq")
    # input gradient with residual and ReLU mask
    dy = O.portable_normal(seed, (N, Cout, H, W), stream=3)
    res = O.portable_normal(seed, (N, Cin, H, W), stream=4)
    msk = O.portable_normal(seed, (N, Cin, H, W), stream=5)
    dyd, resd, mskd = nhwc(dy).cuda(), nhwc(res).cuda(), nhwc(msk).cuda()
    ops.conv2d_patch(2)
    try:
        dx = ops.conv2d_dgrad_split(dyd, wS, wd.shape, xd.shape, 1, 1, residual=resd, relu_src=mskd)
        ops.conv2d_patch(0)
        dx_old = ops.conv2d_dgrad_split(dyd, wS, wd.shape, xd.shape, 1, 1, residual=resd, relu_src=mskd)
    finally:
        ops.conv2d_patch(default)
    torch.cuda.synchronize()
    if big:
        dx_ref = nchw(dx_old.cpu())
    else:
        dx_ref = (O.conv2d_dgrad(dy, w, x.shape, 1, 1) + res) * (msk > 0)
        assert_close(nchw(dx_old.cpu()), dx_ref, atol=0, rtol=2e-5, name="per-tap dgrad")
    assert_close(nchw(dx.cpu()), dx_ref, atol=0, rtol=2e-5, name="patch dgrad")
    # ... and with the reduction pass of the BatchNorm backward that consumes dx fused into the epilogue (with / without residual)
    M = N * H * W
    z = nhwc(O.portable_normal(seed, (N, Cin, H, W), stream=6, mean=0.3, std=1.5)).cuda()
    zpart = torch.empty(ops.bn_stats_partial_elems(M, Cin), device="cuda")
    zt = ops.bn_stats_partial(z.view(M, Cin), M, Cin, zpart)
    mean, invstd = torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
    ops.bn_finalize(zpart, zt, M, Cin, mean, invstd, None, None)
    gamma = O.portable_normal(seed, (Cin,), stream=9, mean=1.0, std=0.2).cuda()
    for with_res in (True, False):
        got = {}
        try:
            for mode in (2, 0):
                ops.conv2d_patch(mode)
                rpart = torch.full((ops.conv2d_dgrad_bn_partial_elems(N, H, W, Cin),), float("nan"), device="cuda")
                dxr, rtiles = ops.conv2d_dgrad_split(dyd, wS, wd.shape, xd.shape, 1, 1, residual=resd if with_res else None, relu_src=mskd,
                                                     bn_reqs=[(z, mean, invstd, rpart)])
                o, dg, db = torch.empty_like(dxr), torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
                ops.bn_bwd_from_partial(dxr.view(M, Cin), z.view(M, Cin), mean, invstd, gamma, o.view(M, Cin), dg, db, rpart, rtiles, M, Cin)
                torch.cuda.synchronize()
                got[mode] = (dxr, dg, db, o)
        finally:
            ops.conv2d_patch(default)
        if with_res:
            assert torch.equal(got[2][0], dx), "the fused reductions must not change dx"
        assert_close(got[2][1], got[0][1], atol=2e-6 * got[0][1].abs().max().item(), name=f"patch fused dgamma (residual={with_res})")
        assert_close(got[2][2], got[0][2], atol=2e-6 * got[0][2].abs().max().item(), name=f"patch fused dbeta (residual={with_res})")
        assert_close(got[2][3], got[0][3], atol=2e-6 * got[0][3].abs().max().item(), rtol=2e-5, name=f"patch fused BN dx (residual={with_res})")


@pytest.mark.parametrize("case", [(2, 20, 12, 64, 128), (3, 14, 14, 128, 256), (3, 7, 7, 256, 512), (5, 9, 11, 64, 128), (40, 28, 28, 128, 256),
                                  (64, 16, 8, 256, 512)], ids=lambda c: "x".join(map(str, c)))
def test_conv_dgrad_merged_parity_classes(ops, case):
    """Stride-2 input gradient (3x3, pad 1): the four output parity classes in one launch (igemm_split_classes_kernel, longest K first) against
    one launch per class -- same tiles, same K order: bit-identical dx; the fused BatchNorm-backward sums agree to summation order of the
    partial rows; class subsets with a per-class residual (the downsample fold of the encoder) included."""
    N, H, W, Cin, Cout = case
    seed = sum(case) + 13
    OH, OW = ops.conv_out(H, 3, 2, 1), ops.conv_out(W, 3, 2, 1)
    w = O.portable_normal(seed, (Cout, Cin, 3, 3), stream=2, std=math.sqrt(2.0 / (Cin * 9)))
    dy = O.portable_normal(seed, (N, Cout, OH, OW), stream=3)
    res = O.portable_normal(seed, (N, Cin, H, W), stream=4)
    msk = O.portable_normal(seed, (N, Cin, H, W), stream=5)
    z = nhwc(O.portable_normal(seed, (N, Cin, H, W), stream=6, mean=0.3, std=1.5)).cuda()
    wd, dyd, resd, mskd = hwio(w).cuda(), nhwc(dy).cuda(), nhwc(res).cuda(), nhwc(msk).cuda()
    wS = ops.conv2d_wsplit(wd, False)
    M = N * H * W
    zpart = torch.empty(ops.bn_stats_partial_elems(M, Cin), device="cuda")
    zt = ops.bn_stats_partial(z.view(M, Cin), M, Cin, zpart)
    mean, invstd = torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
    ops.bn_finalize(zpart, zt, M, Cin, mean, invstd, None, None)
    gamma = O.portable_normal(seed, (Cin,), stream=9, mean=1.0, std=0.2).cuda()
    assert ops.conv2d_dgrad_merge() == 1
    got = {}
    try:
        for merge in (1, 0):
            ops.conv2d_dgrad_merge(merge)
            rpart = torch.full((ops.conv2d_dgrad_bn_partial_elems(N, H, W, Cin),), float("nan"), device="cuda")
            dx, rt = ops.conv2d_dgrad_split(dyd, wS, wd.shape, (N, H, W, Cin), 2, 1, residual=resd, relu_src=mskd, bn_reqs=[(z, mean, invstd, rpart)])
            o, dg, db = torch.empty_like(dx), torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
            ops.bn_bwd_from_partial(dx.view(M, Cin), z.view(M, Cin), mean, invstd, gamma, o.view(M, Cin), dg, db, rpart, rt, M, Cin)
            # class subset: classes 1-3 only, residual on class 3 only, into a prefilled buffer (what the downsample fold launches)
            dx2 = torch.full((N, H, W, Cin), 7.0, device="cuda")
            ops.conv2d_dgrad_split(dyd, wS, wd.shape, (N, H, W, Cin), 2, 1, dx=dx2, residual=resd, relu_src=mskd, class_mask=0xE, residual_mask=0x8)
            torch.cuda.synchronize()
            got[merge] = (dx, dg, db, o, dx2)
    finally:
        ops.conv2d_dgrad_merge(1)
    assert torch.equal(got[1][0], got[0][0]), "merged launch: dx must not change"
    assert torch.equal(got[1][4], got[0][4]), "merged launch of a class subset"
    assert torch.all(got[1][4][:, 0::2, 0::2] == 7.0), "class (0, 0) was not requested: its pixels stay untouched"
    dx_ref = (O.conv2d_dgrad(dy, w, (N, Cin, H, W), 2, 1) + res) * (msk > 0) if M * Cin < 1_500_000 else nchw(got[0][0].cpu())
    assert_close(nchw(got[1][0].cpu()), dx_ref, atol=0, rtol=2e-5, name="merged stride-2 dgrad")
    for k, nm in ((1, "dgamma"), (2, "dbeta"), (3, "BN dx")):
        assert_close(got[1][k], got[0][k], atol=2e-6 * got[0][k].abs().max().item(), name=f"merged launch fused {nm}")


@pytest.mark.parametrize("case", [(48, 28, 28, 256, 256, True), (50, 28, 28, 256, 256, True), (70, 28, 28, 128, 128, False)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv_two_phase_launch(ops, case):
    """Row counts between whole rounds of the CUs: whole rounds of a big tile + one launch of the best tile for the remaining rows
    (plan_two_phase) against the single launch -- every output element sees the same K order: bit-identical y / dx; the statistics rows of the
    second launch follow the first's (fp64 BatchNorm statistics of the forward, BatchNorm-backward sums of the input gradient)."""
    N, H, W, Cin, Cout, expect = case         # expect: the planner splits this shape (37 632 / 39 200 rows x 256 columns); 215 tiles of 256x128 fit one round
    seed = sum(case[:5]) + 17
    M = N * H * W
    g = torch.Generator(device="cuda").manual_seed(seed)
    xd = torch.randn((N, H, W, Cin), device="cuda", generator=g)
    wd = torch.randn((3, 3, Cin, Cout), device="cuda", generator=g) * math.sqrt(2.0 / (Cin * 9))
    dyd = torch.randn((N, H, W, Cout), device="cuda", generator=g)
    resd = torch.randn((N, H, W, Cin), device="cuda", generator=g)
    mskd = torch.randn((N, H, W, Cin), device="cuda", generator=g)
    z = torch.randn((N, H, W, Cin), device="cuda", generator=g) * 1.5 + 0.3
    wT, wS = ops.conv2d_wsplit(wd, True), ops.conv2d_wsplit(wd, False)
    zpart = torch.empty(ops.bn_stats_partial_elems(M, Cin), device="cuda")
    zt = ops.bn_stats_partial(z.view(M, Cin), M, Cin, zpart)
    mean, invstd = torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
    ops.bn_finalize(zpart, zt, M, Cin, mean, invstd, None, None)
    gamma = torch.randn((Cin,), device="cuda", generator=g) * 0.2 + 1.0
    default_patch = ops.conv2d_patch()
    assert ops.conv2d_two_phase() == 1
    got = {}
    try:
        ops.conv2d_patch(0)
        for tp in (1, 0):
            ops.conv2d_two_phase(tp)
            part = torch.zeros(ops.conv2d_fwd_partial_elems(N, H, W, Cin, Cout, 3, 3, 1, 1), device="cuda")
            y, tiles = ops.conv2d_fwd_split(xd, wT, wd.shape, 1, 1, bn_partial=part)
            pt = part.view(torch.float64)[:tiles * 2 * Cout].view(tiles, 2, Cout).sum(0)
            rpart = torch.full((ops.conv2d_dgrad_bn_partial_elems(N, H, W, Cin),), float("nan"), device="cuda")
            dx, rt = ops.conv2d_dgrad_split(dyd, wS, wd.shape, xd.shape, 1, 1, residual=resd, relu_src=mskd, bn_reqs=[(z, mean, invstd, rpart)])
            o, dg, db = torch.empty_like(dx), torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
            ops.bn_bwd_from_partial(dx.view(M, Cin), z.view(M, Cin), mean, invstd, gamma, o.view(M, Cin), dg, db, rpart, rt, M, Cin)
            torch.cuda.synchronize()
            got[tp] = (y, tiles, pt, dx, dg, db, rt)
    finally:
        ops.conv2d_two_phase(1)
        ops.conv2d_patch(default_patch)
    assert (got[1][1] != got[0][1]) == expect and (got[1][6] != got[0][6]) == expect, "two-phase schedule: statistics-row count tells whether it ran"
    assert torch.equal(got[1][0], got[0][0]) and torch.equal(got[1][3], got[0][3]), "two launches: y / dx must not change"
    yd = got[1][0].double().reshape(-1, Cout)
    assert_close(got[1][2][0], yd.sum(0), atol=1e-6, rtol=1e-9, name="two-phase fused colsum == sums of the stored y")
    assert_close(got[1][2][1], (yd ** 2).sum(0), atol=1e-6, rtol=1e-9, name="two-phase fused colsumsq")
    assert_close(got[1][4], got[0][4], atol=2e-6 * got[0][4].abs().max().item(), name="two-phase fused dgamma")
    assert_close(got[1][5], got[0][5], atol=2e-6 * got[0][5].abs().max().item(), name="two-phase fused dbeta")


@pytest.mark.parametrize("case", [(3, 20, 12), (2, 56, 56), (25, 56, 56), (25, 64, 32)], ids=lambda c: "x".join(map(str, c)))
def test_conv_bn_fold(ops, case):
    """relu(bn(y)) folded into the operands of the 64 -> 64 3x3 convolutions (conv2 of the layer1 blocks): forward (+ statistics), weight
    gradient and the ReLU mask of the input gradient re-form the activation from y with bn_apply's expression -- every result is
    BIT-IDENTICAL to the path that materialises a = relu(bn(y)) first."""
    N, H, W = case
    seed = sum(case) + 23
    g = torch.Generator(device="cuda").manual_seed(seed)
    M = N * H * W
    y1 = torch.randn((N, H, W, 64), device="cuda", generator=g) * 1.3 + 0.2
    wd = torch.randn((3, 3, 64, 64), device="cuda", generator=g) * math.sqrt(2.0 / 576)
    dyd = torch.randn((N, H, W, 64), device="cuda", generator=g)
    gamma = torch.randn((64,), device="cuda", generator=g) * 0.3 + 1.0
    beta = torch.randn((64,), device="cuda", generator=g) * 0.3
    part = torch.empty(ops.bn_stats_partial_elems(M, 64), device="cuda")
    t = ops.bn_stats_partial(y1.view(M, 64), M, 64, part)
    mean, invstd = torch.empty(64, device="cuda"), torch.empty(64, device="cuda")
    ops.bn_finalize(part, t, M, 64, mean, invstd, None, None)
    bn_in = (mean, invstd, gamma, beta)
    a1 = torch.empty_like(y1)
    ops.bn_apply(y1, mean, invstd, gamma, beta, a1, M, 64, True)
    wT, wS = ops.conv2d_wsplit(wd, True), ops.conv2d_wsplit(wd, False)
    default_patch = ops.conv2d_patch()
    tiles = (N * H * W + 255) // 256         # the patch kernel is the default where its grid uses >= 75 % of the slots of its rounds of 256 CUs
    default_on = tiles * 4 >= ((tiles + 255) // 256) * 256 * 3
    assert ops.conv2d_bnfold_supported(N, H, W, 64, 64, 3, 3, 1, 1) == default_on, "folded only where the patch kernel would run anyway"
    ops.conv2d_patch(2)            # the reference launches on the same (persistent patch) kernel: another kernel sums K in another order
    try:
        assert ops.conv2d_bnfold_supported(N, H, W, 64, 64, 3, 3, 1, 1)
        assert not ops.conv2d_bnfold_supported(N, H, W, 128, 128, 3, 3, 1, 1) and not ops.conv2d_bnfold_supported(N, H, W, 64, 64, 3, 3, 2, 1)
        _bn_fold_checks(ops, N, H, W, M, y1, a1, wd, wT, wS, dyd, bn_in, mean, invstd, gamma, beta)
    finally:
        ops.conv2d_patch(default_patch)


def _bn_fold_checks(ops, N, H, W, M, y1, a1, wd, wT, wS, dyd, bn_in, mean, invstd, gamma, beta):
    # forward with statistics
    p1 = torch.zeros(ops.conv2d_fwd_partial_elems(N, H, W, 64, 64, 3, 3, 1, 1), device="cuda")
    p2 = torch.zeros_like(p1)
    y_ref, t1 = ops.conv2d_fwd_split(a1, wT, wd.shape, 1, 1, bn_partial=p1)
    y_f, t2 = ops.conv2d_fwd_split_bnin(y1, wT, wd.shape, 1, 1, bn_in, bn_partial=p2)
    y_e, _ = ops.conv2d_fwd_split_bnin(y1, wT, wd.shape, 1, 1, bn_in)
    torch.cuda.synchronize()
    assert t1 == t2 and torch.equal(y_f, y_ref) and torch.equal(y_e, y_ref), "folded forward"
    assert torch.equal(p1.view(torch.float64)[:t1 * 128], p2.view(torch.float64)[:t2 * 128]), "folded forward: statistics rows"
    # weight gradient
    ws = torch.empty(ops.conv2d_wgrad_split_ws_bytes(N, H, W, 64, 64, 3, 3, 1, 1) // 4 + 4, device="cuda")
    dw_ref, dw_f = torch.empty_like(wd), torch.empty_like(wd)
    ops.conv2d_wgrad_split(a1, dyd, dw_ref, 1, 1, ws)
    ops.conv2d_wgrad_split_bnin(y1, dyd, dw_f, 1, 1, ws, bn_in)
    torch.cuda.synchronize()
    assert torch.equal(dw_f, dw_ref), "folded weight gradient"
    # input gradient: mask relu(bn(y1)) > 0 re-formed from the BatchNorm input of the fused reduction
    need = ops.conv2d_dgrad_bn_partial_elems(N, H, W, 64)
    r1, r2 = torch.full((need,), float("nan"), device="cuda"), torch.full((need,), float("nan"), device="cuda")
    dx_ref, rt1 = ops.conv2d_dgrad_split(dyd, wS, wd.shape, (N, H, W, 64), 1, 1, relu_src=a1, bn_reqs=[(y1, mean, invstd, r1)])
    dx_f = torch.empty_like(dx_ref)
    _, rt2 = ops.conv2d_dgrad_split_bnmask(dyd, wS, wd.shape, (N, H, W, 64), 1, 1, dx_f, (y1, mean, invstd, r2), gamma, beta)
    torch.cuda.synchronize()
    assert rt1 == rt2 and torch.equal(dx_f, dx_ref) and torch.equal(r1[:rt1 * 128], r2[:rt2 * 128]), "folded input-gradient mask"
    assert (dx_ref == 0).float().mean().item() > 0.2, "the mask must actually mask"


STEM_CASES = [
    # N, H, W, Cin: the 7x7 / 2 / 3 stem (backbone.py:79-83) on the persistent split-arithmetic kernels (stem_split.hip)
    (2, 40, 24, 1),        # audio, ragged tiles (OH x OW = 20 x 12)
    (3, 36, 36, 3),        # visual, 18 x 18 outputs: one full + three partial tiles per image
    (2, 128, 64, 1),       # smoke()-sized spectrogram: whole tiles only (64 x 32)
    (5, 64, 64, 3),        # test-sized frames, 32 x 32 outputs
    (1, 7, 9, 3),          # smaller than one tile, odd sizes
    (70, 96, 33, 1),       # more tiles (70 x 3 x 2) than one round of some grids; odd width
    (40, 224, 224, 3),     # real frame size: 1960 tiles > 256 workgroups: the persistent loop runs ~8 tiles per workgroup
]


@pytest.mark.parametrize("case", STEM_CASES, ids=lambda c: "x".join(map(str, c)))
def test_stem_split_fwd_wgrad(ops, case):
    """Stem forward (+ fused fp64 BatchNorm statistics) and weight gradient on the split arithmetic vs the oracle at the
    tolerance of every other conv (2e-5 of max|ref|), vs the exact-fp32 stem kernels, bitwise reproducible, and -- like the
    reference's convolution -- a non-finite input pixel only reaches the outputs whose 7x7 window contains it."""
    N, H, W, Cin = case
    seed = sum(case)
    big = N * H * W * Cin > 2_000_000
    x = O.portable_normal(seed, (N, Cin, H, W), stream=1)
    w = O.portable_normal(seed, (64, Cin, 7, 7), stream=2, std=math.sqrt(2.0 / (Cin * 49)))
    xd, wd = nhwc(x).cuda(), hwio(w).cuda()
    part = torch.zeros(ops.conv2d_stem_fwd_partial_elems(), device="cuda")
    y, tiles = ops.conv2d_stem_fwd_split(xd, wd, bn_partial=part)
    y32, _ = ops.conv2d_fwd(xd, wd, 2, 3)
    torch.cuda.synchronize()
    if big:        # the oracle takes minutes at this size: the exact-fp32 MFMA kernel (pinned to the oracle above) is the reference
        y_ref = nchw(y32.cpu())
    else:
        y_ref = O.conv2d_fwd(x, w, 2, 3)
        assert_close(nchw(y32.cpu()), y_ref, atol=0, rtol=2e-5, name="fp32 stem fwd")
    assert_close(nchw(y.cpu()), y_ref, atol=0, rtol=2e-5, name="split stem fwd")
    # statistics: per tile and lane fp32 sums of deviations from the lane's first value, folded into fp64 (stem_split.hip): compare
    # with the fp64 sums of the stored y relative to sum |y| resp. sum y^2, and through the quantities BatchNorm forms from them
    pt = part.view(torch.float64)[:tiles * 2 * 64].view(tiles, 2, 64).sum(0).cpu()
    yd = y.double().cpu().reshape(-1, 64)
    e0 = ((pt[0] - yd.sum(0)).abs() / yd.abs().sum(0)).max().item()
    e1 = ((pt[1] - (yd ** 2).sum(0)).abs() / (yd ** 2).sum(0)).max().item()
    assert e0 < 2e-6 and e1 < 2e-6, (e0, e1)
    Mtot = yd.shape[0]
    var_f, var_r = pt[1] / Mtot - (pt[0] / Mtot) ** 2, yd.var(0, unbiased=False)
    assert ((var_f - var_r).abs() / var_r).max().item() < 1e-5, "batch variance from the fused statistics"
    for wv in (4, 8):                       # both workgroup shapes give the same bits
        ops.conv2d_stem_waves(wv)
        y2, _ = ops.conv2d_stem_fwd_split(xd, wd)
        assert torch.equal(y, y2), "stem forward must be bitwise reproducible (and independent of the statistics request / wave count)"
    ops.conv2d_stem_waves(0)
    # weight gradient
    dy = O.portable_normal(seed, tuple(y_ref.shape), stream=3)
    dyd = nhwc(dy).cuda()
    dw = torch.empty((7, 7, Cin, 64), device="cuda")
    ws = torch.empty(ops.conv2d_stem_wgrad_split_ws_bytes(Cin) // 4, device="cuda")
    ops.conv2d_stem_wgrad_split(xd, dyd, dw, 2, 3, ws)
    dw32 = torch.empty_like(dw)
    ws32 = torch.empty(ops.conv2d_wgrad_ws_bytes(N, H, W, Cin, 64, 7, 7, 2, 3) // 4 + 4, device="cuda")
    ops.conv2d_wgrad(xd, dyd, dw32, 2, 3, ws32)
    torch.cuda.synchronize()
    if big:
        dw_ref = oihw(dw32.cpu())
    else:
        dw_ref = O.conv2d_wgrad(x, dy, w.shape, 2, 3)
        assert_close(oihw(dw32.cpu()), dw_ref, atol=0, rtol=2e-5, name="fp32 stem wgrad")
    assert_close(oihw(dw.cpu()), dw_ref, atol=0, rtol=2e-5, name="split stem wgrad")
    dw2 = torch.empty_like(dw)
    ops.conv2d_stem_wgrad_split(xd, dyd, dw2, 2, 3, ws)
    assert torch.equal(dw, dw2), "stem weight gradient must be bitwise reproducible"
    # window semantics: poison one pixel; exactly the outputs whose window covers it become non-finite
    if not big:
        n0, iy, ix = N - 1, H // 2, W // 2 + 1
        xp = xd.clone()
        xp[n0, iy, ix, Cin - 1] = float("inf")
        yp, _ = ops.conv2d_stem_fwd_split(xp, wd)
        bad = ~torch.isfinite(yp).all(dim=3).cpu()
        want = torch.zeros_like(bad)
        OH, OW = bad.shape[1], bad.shape[2]
        for oy in range(OH):
            for ox in range(OW):
                if 0 <= iy - (2 * oy - 3) < 7 and 0 <= ix - (2 * ox - 3) < 7:
                    want[n0, oy, ox] = True
        assert torch.equal(bad, want), "an Inf input pixel must poison exactly the outputs whose window contains it"
        good = ~want
        assert torch.equal(yp.cpu()[good], y.cpu()[good])


@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 3, 4], ids=lambda c: f"tile{c}")
@pytest.mark.parametrize("case", SPLIT_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_split_fwd_dgrad(ops, case, cfg):
    """Split-bf16 arithmetic (6 bf16 products per fp32 product): same parity bar as the fp32-MFMA kernels
    (2e-5 of max|ref|), every tile configuration (256x128 / 128x128 / 256x64 on 8 waves, 128x64 / 64x64 on 4 waves; K stage 32 or 16)."""
    N, H, W, Cin, Cout, k, s, p = case
    seed = sum(case)
    x = O.portable_normal(seed, (N, Cin, H, W), stream=1)
    w = O.portable_normal(seed, (Cout, Cin, k, k), stream=2, std=math.sqrt(2.0 / (Cin * k * k)))
    y_ref = O.conv2d_fwd(x, w, s, p)
    dy = O.portable_normal(seed, tuple(y_ref.shape), stream=3)
    xd, wd, dyd = nhwc(x).cuda(), hwio(w).cuda(), nhwc(dy).cuda()
    ops.conv2d_split_cfg(cfg)
    try:
        w_t, w_n = ops.conv2d_wsplit(wd, True), ops.conv2d_wsplit(wd, False)
        part = torch.zeros(ops.conv2d_fwd_partial_elems(N, H, W, Cin, Cout, k, k, s, p), device="cuda")
        y, tiles = ops.conv2d_fwd_split(xd, w_t, wd.shape, s, p, bn_partial=part)
        assert_close(nchw(y.cpu()), y_ref, atol=0, rtol=2e-5, name="split conv fwd")
        pt = part.view(torch.float64)[:tiles * 2 * Cout].view(tiles, 2, Cout).sum(0).cpu()      # fp64 partials
        assert_close(pt[0], y_ref.double().sum(dim=(0, 2, 3)), atol=1e-3, rtol=2e-5, name="split fused colsum")
        assert_close(pt[1], (y_ref.double() ** 2).sum(dim=(0, 2, 3)), atol=1e-3, rtol=2e-5, name="split fused colsumsq")
        dx_ref = O.conv2d_dgrad(dy, w, x.shape, s, p)
        dx = ops.conv2d_dgrad_split(dyd, w_n, wd.shape, (N, H, W, Cin), s, p)
        assert_close(nchw(dx.cpu()), dx_ref, atol=0, rtol=2e-5, name="split conv dgrad")
        res = O.portable_normal(seed, (N, Cin, H, W), stream=4)
        msk = O.portable_normal(seed, (N, Cin, H, W), stream=5)
        dx2 = torch.empty_like(dx)
        ops.conv2d_dgrad_split(dyd, w_n, wd.shape, (N, H, W, Cin), s, p, dx=dx2, residual=nhwc(res).cuda(),
                               relu_src=nhwc(msk).cuda())
        assert_close(nchw(dx2.cpu()), (dx_ref + res) * (msk > 0), atol=0, rtol=2e-5, name="split dgrad+res+mask")
        ops.conv2d_dgrad_split(dyd, w_n, wd.shape, (N, H, W, Cin), s, p, dx=dx, residual=dx)
        assert_close(nchw(dx.cpu()), 2 * dx_ref, atol=0, rtol=2e-5, name="split dgrad accumulate")
        if cfg == -1:   # the weight gradient has its own tiles (128x128 / 64x64)
            dw_ref = O.conv2d_wgrad(x, dy, w.shape, s, p)
            ws = torch.empty(ops.conv2d_wgrad_split_ws_bytes(N, H, W, Cin, Cout, k, k, s, p) // 4 + 4, device="cuda")
            dw = ops.conv2d_wgrad_split(xd, dyd, torch.empty_like(wd), s, p, ws)
            assert_close(oihw(dw.cpu()), dw_ref, atol=0, rtol=2e-5, name="split conv wgrad")
    finally:
        ops.conv2d_split_cfg(-1)


def test_wsplit_batch_matches_per_conv(ops):
    """The one-launch weight split of an encoder equals the per-conv entry point, both orientations, bit for bit,
    and hi + mid + lo reproduces the fp32 weights exactly."""
    from mla_hip.encoder import ResNet18Encoder
    enc = ResNet18Encoder("visual", "cuda", seed=3, conv_math="split")
    enc._refresh_wsplit(ops.cur_stream())
    for name, (w_fwd, w_dgrad) in enc.wsp.items():
        w = enc.p[name + ".weight"]
        assert torch.equal(w_fwd, ops.conv2d_wsplit(w, True)), name
        assert torch.equal(w_dgrad, ops.conv2d_wsplit(w, False)), name
        planes = w_dgrad.view(3, *w.shape).view(torch.bfloat16).float()
        assert torch.equal(planes.sum(0), w), name          # exact 3-way split (sum of three bf16 values in fp32)


@pytest.mark.parametrize("math_", ["f32", "split"])
@pytest.mark.parametrize("case", [c for c in SPLIT_CASES if c[5] in (1, 3)], ids=lambda c: "x".join(map(str, c)))
def test_conv_dgrad_fused_bn_reduction(ops, case, math_):
    """mla_conv2d_dgrad[_split]_bn: the input-gradient epilogue also forms the reduction pass (sum g, sum g * xhat per tile) of
    up to two BatchNorm backwards that consume its output.  Against (1) the stand-alone BatchNorm backward on the same dx
    (dgamma, dbeta, dx': same formula, different summation order) and (2) the oracle's BatchNorm backward."""
    N, H, W, Cin, Cout, k, s, p = case
    seed = sum(case) + 7
    x = O.portable_normal(seed, (N, Cin, H, W), stream=1)
    w = O.portable_normal(seed, (Cout, Cin, k, k), stream=2, std=math.sqrt(2.0 / (Cin * k * k)))
    y_ref = O.conv2d_fwd(x, w, s, p)
    dy = O.portable_normal(seed, tuple(y_ref.shape), stream=3)
    res = O.portable_normal(seed, (N, Cin, H, W), stream=4)
    msk = O.portable_normal(seed, (N, Cin, H, W), stream=5)
    wd, dyd = hwio(w).cuda(), nhwc(dy).cuda()
    M = N * H * W
    # two BatchNorm layers (inputs z0, z1: any tensors of dx's shape with their batch statistics)
    zs = [nhwc(O.portable_normal(seed, (N, Cin, H, W), stream=6 + q, mean=0.3 * q, std=1.0 + q)).cuda() for q in range(2)]
    stats = []
    for z in zs:
        part = torch.empty(ops.bn_stats_partial_elems(M, Cin), device="cuda")
        tiles = ops.bn_stats_partial(z.view(M, Cin), M, Cin, part)
        mean, invstd = torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
        ops.bn_finalize(part, tiles, M, Cin, mean, invstd, None, None)
        stats.append((mean, invstd))
    gamma = O.portable_normal(seed, (Cin,), stream=9, mean=1.0, std=0.2).cuda()
    need = ops.conv2d_dgrad_bn_partial_elems(N, H, W, Cin)
    for nreq in (1, 2):
        parts = [torch.full((need,), float("nan"), device="cuda") for _ in range(nreq)]
        reqs = [(zs[q],) + stats[q] + (parts[q],) for q in range(nreq)]
        dx = torch.empty((N, H, W, Cin), device="cuda")
        kw = dict(dx=dx, residual=nhwc(res).cuda(), relu_src=nhwc(msk).cuda(), bn_reqs=reqs)
        if math_ == "split":
            _, tiles = ops.conv2d_dgrad_split(dyd, ops.conv2d_wsplit(wd, False), wd.shape, (N, H, W, Cin), s, p, **kw)
            dx_plain = ops.conv2d_dgrad_split(dyd, ops.conv2d_wsplit(wd, False), wd.shape, (N, H, W, Cin), s, p,
                                              residual=kw["residual"], relu_src=kw["relu_src"])
        else:
            wt_ws = torch.empty(wd.numel(), device="cuda")
            _, tiles = ops.conv2d_dgrad(dyd, wd, (N, H, W, Cin), s, p, wt_ws, **kw)
            dx_plain = ops.conv2d_dgrad(dyd, wd, (N, H, W, Cin), s, p, wt_ws, residual=kw["residual"], relu_src=kw["relu_src"])
        assert torch.equal(dx, dx_plain), "the fused reductions must not change dx"
        assert tiles >= 1
        ws = torch.empty(ops.bn_bwd_ws_elems(M, Cin), device="cuda")
        for q in range(nreq):
            mean, invstd = stats[q]
            o1, dg1, db1 = torch.empty_like(dx), torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
            ops.bn_bwd(dx.view(M, Cin), zs[q].view(M, Cin), mean, invstd, gamma, o1.view(M, Cin), dg1, db1, ws, M, Cin)
            o2, dg2, db2 = torch.empty_like(dx), torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
            ops.bn_bwd_from_partial(dx.view(M, Cin), zs[q].view(M, Cin), mean, invstd, gamma, o2.view(M, Cin), dg2, db2, parts[q],
                                    tiles, M, Cin)
            sc_g, sc_b = dg1.abs().max().item(), db1.abs().max().item()
            assert_close(dg2, dg1, atol=2e-6 * sc_g, name=f"fused dgamma (req {q})")
            assert_close(db2, db1, atol=2e-6 * sc_b, name=f"fused dbeta (req {q})")
            assert_close(o2, o1, atol=2e-6 * o1.abs().max().item(), name=f"fused BN dx (req {q})")
            g_ref = nchw(dx.cpu())
            dx_ref, dgamma_ref, dbeta_ref = O.bn_train_bwd(g_ref, nchw(zs[q].cpu()), gamma.cpu(), mean.cpu(), invstd.cpu())
            assert_close(dg2, dgamma_ref, atol=1e-4, rtol=5e-5, name=f"fused dgamma vs oracle (req {q})")
            assert_close(db2, dbeta_ref, atol=1e-4, rtol=5e-5, name=f"fused dbeta vs oracle (req {q})")
            assert_close(nchw(o2.cpu()), dx_ref, atol=2e-6 * max(1.0, dx_ref.abs().max().item()), rtol=5e-5, name=f"fused BN dx vs oracle (req {q})")
    if math_ == "f32":      # the torch.ops face of the same pair
        dx_t, part_t, tiles_t = torch.ops.mla_hip.conv2d_dgrad_bn(dyd, wd, [N, H, W, Cin], s, p, zs[0], stats[0][0], stats[0][1],
                                                                  nhwc(res).cuda(), nhwc(msk).cuda())
        o_t, dg_t, db_t = torch.ops.mla_hip.bn_bwd_from_partial(dx_t, zs[0], stats[0][0], stats[0][1], gamma, part_t, tiles_t)
        assert torch.equal(dx_t, dx) and tiles_t == tiles
        ops.bn_bwd_from_partial(dx.view(M, Cin), zs[0].view(M, Cin), stats[0][0], stats[0][1], gamma, o2.view(M, Cin), dg2, db2, parts[0], tiles, M, Cin)
        assert torch.equal(o_t, o2) and torch.equal(dg_t, dg2) and torch.equal(db_t, db2)


def test_conv_split_is_not_reduced_precision(ops):
    """The claim behind conv_math="split": against an fp64 reference the six-product bf16 split is at least as
    accurate as the exact-fp32 MFMA kernel (same inputs, K = 576 .. 4608), element-wise maximum and rms.  The
    three-product set (bf16x3, TF32-like) is measured too and must be visibly worse -- it is not used."""
    import torch.nn.functional as F
    torch.manual_seed(0)
    for (N, H, W, Cin, Cout, k, s, p) in [(4, 28, 28, 64, 64, 3, 1, 1), (4, 14, 14, 256, 256, 3, 1, 1), (8, 7, 7, 512, 512, 3, 1, 1),
                                          (4, 28, 28, 64, 128, 3, 2, 1)]:
        x = torch.randn((N, H, W, Cin), device="cuda") * 3.0 + 0.5        # non-zero mean: no cancellation luck
        w = torch.randn((k, k, Cin, Cout), device="cuda") * 0.05
        yr = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), stride=s, padding=p).permute(0, 2, 3, 1)
        scale = yr.pow(2).mean().sqrt()
        y32, _ = ops.conv2d_fwd(x, w, s, p)
        ys, _ = ops.conv2d_fwd_split(x, ops.conv2d_wsplit(w, True), w.shape, s, p)
        ops.conv2d_split_terms(3)
        try:
            y3, _ = ops.conv2d_fwd_split(x, ops.conv2d_wsplit(w, True), w.shape, s, p)
        finally:
            ops.conv2d_split_terms(6)
        e32 = (y32.double() - yr).abs()
        es = (ys.double() - yr).abs()
        e3 = (y3.double() - yr).abs()
        rms = lambda e: float(e.pow(2).mean().sqrt() / scale)
        assert rms(es) <= 1.1 * rms(e32) + 1e-9, (rms(es), rms(e32))
        assert float(es.max()) <= 1.5 * float(e32.max()), (float(es.max()), float(e32.max()))
        assert rms(e3) > 3 * rms(es), "bf16x3 should be measurably less accurate than the six-product split"


@pytest.mark.parametrize("cfg", [0, 1, 2, 3], ids=lambda c: f"f32tile{c}")
def test_conv_f32_every_tile(ops, cfg):
    """Every fp32 tile (128x128, 256x64, 64x64, 128x64) forced on a ragged problem: forward + input gradient parity."""
    N, H, W, Cin, Cout, k, s, p = 2, 33, 17, 128, 128, 3, 1, 1
    seed = 123 + cfg
    x = O.portable_normal(seed, (N, Cin, H, W), stream=1)
    w = O.portable_normal(seed, (Cout, Cin, k, k), stream=2, std=math.sqrt(2.0 / (Cin * k * k)))
    y_ref = O.conv2d_fwd(x, w, s, p)
    dy = O.portable_normal(seed, tuple(y_ref.shape), stream=3)
    xd, wd, dyd = nhwc(x).cuda(), hwio(w).cuda(), nhwc(dy).cuda()
    ops.conv2d_f32_cfg(cfg)
    try:
        part = torch.zeros(ops.conv2d_fwd_partial_elems(N, H, W, Cin, Cout, k, k, s, p), device="cuda")
        y, tiles = ops.conv2d_fwd(xd, wd, s, p, bn_partial=part)
        assert_close(nchw(y.cpu()), y_ref, atol=0, rtol=2e-5, name="conv fwd")
        pt = part.view(torch.float64)[:tiles * 2 * Cout].view(tiles, 2, Cout).sum(0).cpu()      # fp64 partials
        assert_close(pt[0], y_ref.double().sum(dim=(0, 2, 3)), atol=1e-3, rtol=2e-5, name="fused colsum")
        dx = ops.conv2d_dgrad(dyd, wd, (N, H, W, Cin), s, p, torch.empty(w.numel(), device="cuda"))
        assert_close(nchw(dx.cpu()), O.conv2d_dgrad(dy, w, x.shape, s, p), atol=0, rtol=2e-5, name="conv dgrad")
    finally:
        ops.conv2d_f32_cfg(-1)


def test_conv_size_limit_counts_real_tensors(ops):
    """The 32-bit buffer offsets limit each TENSOR of a convolution to 4 GiB.  The check used to multiply the input pixel count
    by max(Cin, Cout), which rejected the audio stem at a per-GPU batch of 128 (input 1 channel: 67 MB, output 1.07 GB)."""
    from mla_hip import MLAHipError
    N, H, W = 130, 1024, 128
    x = torch.randn((N, H, W, 1), device="cuda")
    w = torch.randn((7, 7, 1, 64), device="cuda") * 0.1
    y, _ = ops.conv2d_fwd(x, w, 2, 3)
    y2, _ = ops.conv2d_fwd(x[-2:].contiguous(), w, 2, 3)
    assert torch.equal(y[-2:], y2)
    del y, y2
    with pytest.raises(MLAHipError):                       # a genuinely too large output is still refused
        ops.conv2d_fwd(torch.empty((520, H, W, 1), device="cuda"), w, 2, 3, y=torch.empty((1,), device="cuda"))


def test_conv_rejects_bad_shapes(ops):
    from mla_hip import MLAHipError
    x = torch.zeros((1, 8, 8, 48), device="cuda")
    w = torch.zeros((3, 3, 48, 64), device="cuda")
    with pytest.raises(MLAHipError):
        ops.conv2d_fwd(x, w, 1, 1)                      # Cin not a multiple of 64 and > 4
    with pytest.raises(MLAHipError):
        ops.conv2d_fwd(torch.zeros((1, 8, 8, 64), device="cuda"), torch.zeros((3, 3, 64, 64), device="cuda"), 3, 1)
    with pytest.raises(MLAHipError):
        ops.conv2d_fwd(x.cpu(), w, 1, 1)                # host tensor


@pytest.mark.parametrize("M,C", [(1000, 64), (37, 128), (4096, 256), (300, 512), (70000, 64)])
def test_bn_fwd_bwd(ops, M, C):
    # treat as N=1, H=M, W=1
    x = O.portable_normal(M + C, (1, C, M, 1), stream=1, mean=0.7, std=1.8)
    gamma = O.portable_normal(M + C, (C,), stream=2, mean=1.0, std=0.2)
    beta = O.portable_normal(M + C, (C,), stream=3, std=0.3)
    res = O.portable_normal(M + C, (1, C, M, 1), stream=4)
    rm, rv = torch.zeros(C) + 0.25, torch.ones(C) * 1.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y_ref, mean_ref, invstd_ref = O.bn_train_fwd(x, gamma, beta, rm_ref, rv_ref)
    out_ref = torch.relu(y_ref + res)

    xd = nhwc(x).cuda().view(M, C)
    part = torch.empty(ops.bn_stats_partial_elems(M, C), device="cuda")
    tiles = ops.bn_stats_partial(xd, M, C, part)
    mean, invstd = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    rmd, rvd = rm.cuda(), rv.cuda()
    ops.bn_finalize(part, tiles, M, C, mean, invstd, rmd, rvd)
    assert_close(mean, mean_ref, atol=1e-6, rtol=1e-5, name="bn mean")
    assert_close(invstd, invstd_ref, atol=0, rtol=1e-5, name="bn invstd")
    assert_close(rmd, rm_ref, atol=1e-6, rtol=1e-5, name="running_mean")
    assert_close(rvd, rv_ref, atol=0, rtol=1e-5, name="running_var")
    out = torch.empty((M, C), device="cuda")
    ops.bn_apply(xd, mean, invstd, gamma.cuda(), beta.cuda(), out, M, C, True, residual=nhwc(res).cuda().view(M, C))
    assert_close(out.cpu(), nhwc(out_ref).view(M, C), atol=1e-5, rtol=1e-5, name="bn apply+res+relu")

    # backward with relu mask and residual-branch gradient output
    dout = O.portable_normal(M + C, (1, C, M, 1), stream=5)
    g_ref = dout * (out_ref > 0)
    dx_ref, dgamma_ref, dbeta_ref = O.bn_train_bwd(g_ref, x, gamma, mean_ref, invstd_ref)
    dx, g_out = torch.empty((M, C), device="cuda"), torch.empty((M, C), device="cuda")
    dgamma, dbeta = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ws = torch.empty(ops.bn_bwd_ws_elems(M, C), device="cuda")
    ops.bn_bwd(nhwc(dout).cuda().view(M, C), xd, mean, invstd, gamma.cuda(), dx, dgamma, dbeta, ws, M, C, relu_out=out, g_out=g_out)
    assert_close(g_out.cpu(), nhwc(g_ref).view(M, C), atol=1e-6, name="bn bwd g")
    assert_close(dgamma, dgamma_ref, atol=1e-4, rtol=2e-5, name="dgamma")
    assert_close(dbeta, dbeta_ref, atol=1e-4, rtol=2e-5, name="dbeta")
    assert_close(dx.cpu(), nhwc(dx_ref).view(M, C), atol=1e-6, rtol=2e-5, name="bn dx")
    # in place (dx aliases dout), no mask
    d2 = nhwc(dout).cuda().view(M, C).clone()
    ops.bn_bwd(d2, xd, mean, invstd, gamma.cuda(), d2, dgamma, dbeta, ws, M, C)
    dx_ref2, _, _ = O.bn_train_bwd(dout, x, gamma, mean_ref, invstd_ref)
    assert_close(d2.cpu(), nhwc(dx_ref2).view(M, C), atol=1e-6, rtol=2e-5, name="bn dx in place")


@pytest.mark.parametrize("ratio", [50.0, 2000.0])
def test_bn_large_mean(ops, ratio):
    """the batch variance is formed as E[x^2] - E[x]^2, which cancels catastrophically in fp32 when
    |mean| >> std (ATen uses Welford).  The per-tile sums are therefore fp64 (x * x is exact in fp64), in the conv epilogue
    and in the standalone statistics kernel alike.  Channels with mean / std = 50 and 2000: mean, invstd, running_var and the
    normalised output against an fp64 evaluation."""
    M, C = 20000, 64
    g = torch.Generator().manual_seed(int(ratio))
    z = torch.randn((M, C), generator=g, dtype=torch.float64)
    x64 = (z * 0.37 + 0.37 * ratio * torch.linspace(-1.0, 1.0, C, dtype=torch.float64).sign())          # every channel: |mean| / std = ratio
    x = x64.float()
    xd = x.cuda()
    m_ref, v_ref = x.double().mean(0), x.double().var(0, unbiased=False)
    part = torch.empty(ops.bn_stats_partial_elems(M, C), device="cuda")
    tiles = ops.bn_stats_partial(xd, M, C, part)
    mean, invstd = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    ops.bn_finalize(part, tiles, M, C, mean, invstd, rm, rv)
    assert_close(mean, m_ref, atol=0, rtol=2e-7, name="mean")
    assert_close(invstd, 1.0 / torch.sqrt(v_ref + 1e-5), atol=0, rtol=1e-6, name="invstd (standalone statistics)")
    assert_close(rv, 0.9 + 0.1 * v_ref * M / (M - 1), atol=0, rtol=1e-6, name="running_var")
    out = torch.empty((M, C), device="cuda")
    ones, zeros = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    ops.bn_apply(xd, mean, invstd, ones, zeros, out, M, C, False)
    ref = (x.double() - m_ref) / torch.sqrt(v_ref + 1e-5)
    # the subtraction x - mean is exact-ish in fp32 only to ulp(x) = ratio * 6e-8 relative to std: that floor is inherent to fp32 data
    assert_close(out, ref, atol=4e-7 * ratio + 2e-6, rtol=0, name="normalised output")
    # the same statistics out of the conv epilogue: a 1x1 convolution with identity weights reproduces x as its output
    if ratio <= 50:
        xin = x.view(100, 200, 1, C).contiguous().cuda()
        w = torch.eye(C, device="cuda").view(1, 1, C, C).contiguous()
        for conv in ("f32", "split"):
            part2 = torch.empty(ops.conv2d_fwd_partial_elems(100, 200, 1, C, C, 1, 1, 1, 0), device="cuda")
            if conv == "f32":
                y, t2 = ops.conv2d_fwd(xin, w, 1, 0, bn_partial=part2)
            else:
                y, t2 = ops.conv2d_fwd_split(xin, ops.conv2d_wsplit(w, True), w.shape, 1, 0, bn_partial=part2)
            assert torch.equal(y.view(M, C), xd), "identity 1x1 convolution"
            ops.bn_finalize(part2, t2, M, C, mean, invstd, None, None)
            assert_close(mean, m_ref, atol=0, rtol=2e-7, name=f"mean (conv epilogue, {conv})")
            assert_close(invstd, 1.0 / torch.sqrt(v_ref + 1e-5), atol=0, rtol=1e-6, name=f"invstd (conv epilogue, {conv})")


@pytest.mark.parametrize("N,H,W,C", [(2, 16, 12, 64), (3, 9, 7, 64), (1, 2, 2, 128)])
def test_maxpool(ops, N, H, W, C):
    x = torch.relu(O.portable_normal(N * H + W, (N, C, H, W), stream=1))       # many exact-zero ties, like the stem
    y_ref, idx_ref = O.maxpool3x3s2_fwd(x)
    OH, OW = y_ref.shape[2:]
    y = torch.empty((N, OH, OW, C), device="cuda")
    idx = torch.empty((N, OH, OW, C), device="cuda", dtype=torch.uint8)
    xd = nhwc(x).cuda()
    ops.maxpool_fwd(xd, y, idx)
    assert_close(nchw(y.cpu()), y_ref, atol=0, name="maxpool fwd")           # bit exact
    dy = O.portable_normal(N * H + W, tuple(y_ref.shape), stream=2)
    dx_ref = O.maxpool3x3s2_bwd(dy, idx_ref, x.shape) * (x > 0)
    dx = torch.empty((N, H, W, C), device="cuda")
    ops.maxpool_bwd(nhwc(dy).cuda(), idx, dx, (N, H, W, C), relu_src=xd)
    assert_close(nchw(dx.cpu()), dx_ref, atol=1e-6, name="maxpool bwd (+relu mask)")
    # without the mask the tie-breaking itself must match ATen (first maximum in window order)
    ops.maxpool_bwd(nhwc(dy).cuda(), idx, dx, (N, H, W, C))
    assert_close(nchw(dx.cpu()), O.maxpool3x3s2_bwd(dy, idx_ref, x.shape), atol=1e-6, name="maxpool bwd ties")


@pytest.mark.parametrize("N,H,W,C", [(2, 16, 12, 64), (3, 9, 7, 64), (1, 2, 2, 128), (4, 112, 40, 64)])
def test_stem_fused(ops, N, H, W, C):
    """The fused stem kernels (BN + ReLU inside the max-pool; BN backward fed by the pooled gradient) against (1) the
    unfused kernel sequence bn_apply -> maxpool_fwd / maxpool_bwd -> bn_bwd (same decisions bit for bit, same pooled output
    bit for bit) and (2) the oracle (backbone.py:150-152 and its autograd)."""
    M = N * H * W
    x = O.portable_normal(M + C, (N, C, H, W), stream=1, mean=0.2, std=1.3)
    gamma = O.portable_normal(M + C, (C,), stream=2, mean=1.0, std=0.2)
    beta = O.portable_normal(M + C, (C,), stream=3, std=0.3)
    xd = nhwc(x).cuda()
    part = torch.empty(ops.bn_stats_partial_elems(M, C), device="cuda")
    tiles = ops.bn_stats_partial(xd.view(M, C), M, C, part)
    mean, invstd = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_finalize(part, tiles, M, C, mean, invstd, None, None)
    ga, be = gamma.cuda(), beta.cuda()
    # unfused
    a = torch.empty_like(xd)
    ops.bn_apply(xd.view(M, C), mean, invstd, ga, be, a.view(M, C), M, C, True)
    OH, OW = ops.conv_out(H, 3, 2, 1), ops.conv_out(W, 3, 2, 1)
    p_u = torch.empty((N, OH, OW, C), device="cuda")
    i_u = torch.empty((N, OH, OW, C), device="cuda", dtype=torch.uint8)
    ops.maxpool_fwd(a, p_u, i_u)
    # fused
    p_f, i_f = torch.empty_like(p_u), torch.empty_like(i_u)
    ops.bn_relu_maxpool_fwd(xd, mean, invstd, ga, be, p_f, i_f)
    assert torch.equal(p_f, p_u), "fused stem forward: pooled output differs from bn_apply -> maxpool"
    assert torch.equal(i_f, i_u), "fused stem forward: max-pool decisions differ"
    # oracle forward
    y_ref, mean_ref, invstd_ref = O.bn_train_fwd(x, gamma, beta, torch.zeros(C), torch.ones(C))
    a_ref = torch.relu(y_ref)
    p_ref, idx_ref = O.maxpool3x3s2_fwd(a_ref)
    assert_close(nchw(p_f.cpu()), p_ref, atol=2e-6, rtol=1e-5, name="fused stem forward vs oracle")

    dpool = O.portable_normal(M + C, tuple(p_ref.shape), stream=4)
    dpd = nhwc(dpool).cuda()
    ws = torch.empty(ops.bn_bwd_ws_elems(M, C), device="cuda")
    dstem = torch.empty_like(xd)
    ops.maxpool_bwd(dpd, i_u, dstem, (N, H, W, C), relu_src=a)
    dy_u, dg_u, db_u = torch.empty_like(xd), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_bwd(dstem.view(M, C), xd.view(M, C), mean, invstd, ga, dy_u.view(M, C), dg_u, db_u, ws, M, C)
    dy_f, dg_f, db_f = torch.empty_like(xd), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_bwd_pooled(dpd, i_f, xd, mean, invstd, ga, be, dy_f, dg_f, db_f, ws)
    scale = dy_u.abs().max().item()
    assert_close(dg_f, dg_u, atol=1e-5 * dg_u.abs().max().item(), name="fused stem dgamma vs unfused")
    assert_close(db_f, db_u, atol=1e-5 * db_u.abs().max().item(), name="fused stem dbeta vs unfused")
    assert_close(dy_f, dy_u, atol=2e-6 * scale, name="fused stem dy vs unfused")
    # oracle backward with the HIP path's own decisions (flip-immune): scatter through idx, mask, BN backward
    code = nchw(i_f.cpu()).long()
    oy = torch.arange(OH).view(1, 1, OH, 1)
    ox = torch.arange(OW).view(1, 1, 1, OW)
    flat = (oy * 2 - 1 + code // 3) * W + (ox * 2 - 1 + code % 3)
    g_ref = O.maxpool3x3s2_bwd(dpool, flat, x.shape) * (nchw(a.cpu()) > 0)
    dx_ref, dgamma_ref, dbeta_ref = O.bn_train_bwd(g_ref, x, gamma, mean_ref, invstd_ref)
    assert_close(dg_f, dgamma_ref, atol=1e-4, rtol=5e-5, name="fused stem dgamma vs oracle")
    assert_close(db_f, dbeta_ref, atol=1e-4, rtol=5e-5, name="fused stem dbeta vs oracle")
    assert_close(nchw(dy_f.cpu()), dx_ref, atol=2e-6 * max(scale, 1.0), rtol=5e-5, name="fused stem dy vs oracle")
    # the torch.ops face
    p_t, i_t = torch.ops.mla_hip.bn_relu_maxpool_fwd(xd, mean, invstd, ga, be)
    dy_t, dg_t, db_t = torch.ops.mla_hip.bn_bwd_pooled(dpd, i_t, xd, mean, invstd, ga, be)
    assert torch.equal(p_t, p_f) and torch.equal(i_t, i_f) and torch.equal(dy_t, dy_f) and torch.equal(dg_t, dg_f)


@pytest.mark.parametrize("B,T,h,w,C", [(4, 1, 32, 4, 512), (4, 3, 7, 7, 512), (3, 2, 1, 1, 512), (2, 1, 5, 3, 64)])
def test_avgpool(ops, B, T, h, w, C):
    f = torch.relu(O.portable_normal(B + T + h, (B * T, C, h, w), stream=1))
    if T == 1:
        ref = f.mean(dim=(2, 3))
    else:
        _, ref = O.av_pool_fwd(f[:B], f, B)
    fd = nhwc(f).cuda()
    y = torch.empty((B, C), device="cuda")
    ops.avgpool_fwd(fd, y, B, T * h * w, C)
    assert_close(y, ref, atol=1e-6, rtol=1e-6, name="avgpool fwd")
    dy = O.portable_normal(B + T + h, (B, C), stream=2)
    dref = (O.visual_pool_bwd(dy, f.shape, B) if T > 1 else O.audio_pool_bwd(dy, f.shape)) * (f > 0)
    dx = torch.empty_like(fd)
    ops.avgpool_bwd(dy.cuda(), dx, B, T * h * w, C, relu_src=fd)
    assert_close(nchw(dx.cpu()), dref, atol=1e-7, rtol=1e-6, name="avgpool bwd")


def test_video_to_nhwc_and_transposes(ops):
    v = O.portable_normal(3, (2, 3, 3, 10, 6), stream=1)
    got = ops.video_to_nhwc(v.cuda()).cpu()
    want = v.permute(0, 2, 1, 3, 4).contiguous().view(6, 3, 10, 6).permute(0, 2, 3, 1)
    assert torch.equal(got, want.contiguous())
    t = O.portable_normal(4, (3, 70, 9, 5), stream=1)
    assert torch.equal(ops.nchw_to_nhwc(t.cuda()).cpu(), nhwc(t))
    assert torch.equal(ops.nhwc_to_nchw(nhwc(t).cuda()).cpu(), t)


@pytest.mark.parametrize("B,D,C", [(64, 512, 6), (8, 512, 6), (5, 768, 101), (32, 768, 4), (1, 512, 6)])
def test_head_ce(ops, B, D, C):
    X = torch.relu(O.portable_normal(B + C, (B, D), stream=1, mean=0.3))
    hd = O.make_head_params(D, C, seed=B)
    labels = O.portable_labels(B + C, B, C)
    inv = 1.0 / (2 * B)                                                         # e.g. world size 2
    logits_r, loss_r, dW_r, db_r, dX_r = O.head_ce_fwd_bwd(X, hd["weight"], hd["bias"], labels)
    scale = B * inv
    f = lambda *s: torch.empty(s, device="cuda")
    logits, loss, dW, db, dX = f(B, C), f(1), f(C, D), f(C), f(B, D)
    ws = f(ops.head_ws_elems(B, C))
    ops.head_ce_fwd_bwd(X.cuda(), hd["weight"].cuda(), hd["bias"].cuda(), labels.cuda(), logits, loss, dW, db, dX, ws, inv)
    assert_close(logits, logits_r, atol=2e-6, rtol=2e-6, name="logits")
    assert_close(loss, (loss_r * scale).reshape(1), atol=2e-6, rtol=2e-6, name="loss")
    assert_close(dW, dW_r * scale, atol=1e-7, rtol=1e-5, name="dW")
    assert_close(db, db_r * scale, atol=1e-7, rtol=1e-5, name="db")
    assert_close(dX, dX_r * scale, atol=1e-8, rtol=1e-5, name="dX")


@pytest.mark.parametrize("D", [512, 768])
def test_gs_project_golden(ops, D, golden_dir):
    """GSPlugin.before_update trajectory vs the REFERENCE's own outputs (tests/golden/gs_kat_d*.npz)."""
    import numpy as np
    import os
    from mla_hip import GSPlugin, SharedHead
    fx = np.load(os.path.join(golden_dir, f"gs_kat_d{D}.npz"))
    D_, C, B, calls, seed = [int(v) for v in fx["meta"]]
    head = SharedHead(D, C, seed=0)
    gs = GSPlugin(dim=D)
    for i in range(calls):
        X = O.portable_normal(seed + i, (B, D), stream=5, mean=0.3, std=0.7).abs()
        G = O.portable_normal(seed + i, (C, D), stream=6, std=0.05)
        head.weight.grad = G.cuda()                        # the plugin reads / rewrites w.grad (utils/utils.py:30-41)
        gs.before_update(head, X.cuda(), i % 7, 7, gs.exp_count)
        gs.exp_count += 1
        # tolerance: 1e-3 (north star) would be loose; fp32 re-association gives ~1e-6
        assert_close(head.weight.grad, fx[f"c{i}.G"], atol=1e-8, rtol=2e-5, name=f"G call {i}")
        Pl = gs.Pl.cpu()
        assert_close(Pl[:8, :8], fx[f"c{i}.Pl.corner"], atol=1e-8, rtol=2e-5, name="Pl corner")
        assert_close(Pl[::16, ::16], fx[f"c{i}.Pl.sub"], atol=1e-8, rtol=2e-5, name="Pl sub")
        assert_close(Pl.sum(1), fx[f"c{i}.Pl.rowsum"], atol=1e-7, rtol=2e-5, name="Pl rowsum")
        assert abs(torch.linalg.norm(Pl).item() - float(fx[f"c{i}.Pl.fro"])) < 1e-5


def test_gs_project_vs_oracle_and_modes(ops):
    from mla_hip import GSPlugin, SharedHead
    D, C, B = 512, 6, 16
    head = SharedHead(D, C, seed=1)
    for mode in ("as_published", "as_intended"):
        gs = GSPlugin(dim=D, mode=mode)
        Pl = torch.eye(D)
        for i in range(3):
            X = O.portable_normal(50 + i, (B, D), stream=1).abs()
            G = O.portable_normal(50 + i, (C, D), stream=2, std=0.1)
            head.weight_grad.copy_(G)                      # trainer path: the flat gradient view is handed over explicitly
            gs.before_update(head, X.cuda(), i, 5, gs.exp_count, grad=head.weight_grad)
            Pl, Gr = O.gs_before_update(Pl, X, G, i, 5, gs.exp_count, mode)
            gs.exp_count += 1
            assert_close(head.weight_grad, Gr, atol=1e-8, rtol=2e-5, name=f"{mode} G {i}")
            assert_close(gs.Pl, Pl, atol=1e-8, rtol=2e-5, name=f"{mode} Pl {i}")
        if mode == "as_published":
            assert torch.equal(gs.Pl.cpu(), torch.eye(D))                      # Q1: never fires


@pytest.mark.parametrize("n", [1, 3, 4, 1027, 3078, 1 << 20])
def test_sgd(ops, n):
    p = O.portable_normal(n, (n,), stream=1)
    g = O.portable_normal(n, (n,), stream=2)
    pd, buf = p.cuda(), torch.zeros(n, device="cuda")
    pr, br = p.clone(), None
    for step, grad in enumerate([g, 0.5 * g, None, g]):                         # None = zeroed gradient (Q6)
        ops.sgd_step(pd, None if grad is None else grad.cuda(), buf, 1e-3, 0.9, 1e-4, first=(step == 0))
        pr, br = O.sgd_step(pr, grad, br, 1e-3)
        assert_close(pd, pr, atol=1e-7, rtol=1e-6, name=f"sgd p step {step}")
        assert_close(buf, br, atol=1e-7, rtol=1e-6, name=f"sgd buf step {step}")
