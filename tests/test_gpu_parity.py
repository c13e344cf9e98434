"""GPU (MI355X): the HIP path, called through the Python mirror -> ctypes -> C ABI, against
  (a) the golden vectors taken from the imported reference (tests/golden), and
  (b) the CPU oracle on fresh seeded inputs (sizes the oracle finishes in seconds),
  (c) size-independent properties at the full 512x512x96 size (round trips, log-det antisymmetry).
Tolerances (BASELINE.json north_star): index work bit-exact, Haar bit-exact (same two fp32 ops), fp32 outputs with
convolutions max|d|/max|ref| <= 1e-4 and L2-relative <= 1e-4."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_close, load_golden, sd_of

pytestmark = pytest.mark.gpu
TOL = 1e-4
T = torch.from_numpy


def cu(a):
    return (T(a) if isinstance(a, np.ndarray) else a).cuda()


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cwfa_amd import _lib
    _lib.lib()                        # the HIP extension must be the thing that runs
    yield
    torch.cuda.synchronize()


@pytest.fixture(autouse=True)
def _inference_mode():
    """This file is the parity suite of the INFERENCE path (fused step plans, epilogue couplings, merged launches): grad mode off, as
    the reference's evaluation code runs it (CWFA.py:134 `torch.no_grad()`).  With grad mode on the same modules run as autograd nodes --
    that path has its own suites (test_gpu_autograd.py, test_gpu_backward.py)."""
    with torch.no_grad():
        yield


def names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(f"{GOLDEN}/{prefix}*.npz"))


# ------------------------------------------------------------------------------------------------ wavelets / gathers
def test_haar1d_golden_bit_exact():
    from cwfa_amd.INN_utils import HaarTransform1D
    fx = load_golden("g01_haar1d")
    m = HaarTransform1D([tuple(fx["x"].shape[1:])])
    (yf,), jf = m((cu(fx["x"]),), rev=False)
    (yr,), jr = m((cu(fx["x"]),), rev=True)
    assert torch.equal(yf.cpu(), T(fx["y_fwd"])) and torch.equal(yr.cpu(), T(fx["y_rev"]))
    assert jf == float(fx["jac_fwd"]) and jr == float(fx["jac_rev"])


@pytest.mark.parametrize("shape", [(1, 2, 1, 1), (3, 6, 5, 7), (2, 10, 9, 12), (1, 48, 64, 64), (2, 4, 3, 1)])
def test_haar1d_vs_oracle_ragged(shape):
    from cwfa_amd import ops
    from oracle import cwfa_oracle as O
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape)))
    for rev in (False, True):
        assert torch.equal(ops.haar1d(x.cuda(), rev).cpu(), O.haar1d(x, rev)[0])
    # strided (channel-sliced) input view and empty batch
    big = torch.randn(shape[0], shape[1] + 4, shape[2], shape[3])
    assert torch.equal(ops.haar1d(big.cuda()[:, 2:-2], False).cpu(), O.haar1d(big[:, 2:-2], False)[0])
    assert ops.haar1d(torch.zeros(0, 4, 3, 3).cuda(), False).shape == (0, 4, 3, 3)


def test_haar1d_full_size_roundtrip():
    """Config-3 size: orthonormal transform -> x recovered to 1 ulp-level, energy preserved."""
    from cwfa_amd import ops
    x = torch.randn(1, 96, 512, 512, device="cuda")
    y = ops.haar1d(x, False)
    xr = ops.haar1d(y, True)
    assert float((xr - x).abs().max()) <= 4e-7 * float(x.abs().max())
    assert abs(float(y.double().pow(2).sum() / x.double().pow(2).sum()) - 1) < 1e-6
    lo = y[:, :48]
    assert torch.equal(ops.haar1d(None, True, lo=lo, hi=y[:, 48:]), xr)


@pytest.mark.parametrize("name", names("g02_"))
def test_haar2d(name):
    from cwfa_amd.FrEIA import modules as Fm
    fx = load_golden(name)
    kw = dict(order_by_wavelet=bool(fx["order_by_wavelet"]), rebalance=float(fx["rebalance"]))
    m = Fm.HaarDownsampling([tuple(fx["x"].shape[1:])], **kw)
    (yf,), jf = m((cu(fx["x"]),), rev=False)
    (xr,), jr = m((cu(fx["z"]),), rev=True)
    assert_close(yf, fx["y_fwd"], 2e-6)
    assert_close(xr, fx["x_rev"], 2e-6)
    assert abs(jf - float(fx["jac_fwd"])) < 1e-9 and abs(jr - float(fx["jac_rev"])) < 1e-9
    up = Fm.HaarUpsampling([tuple(fx["z"].shape[1:])], **kw)
    assert_close(up((cu(fx["z"]),))[0][0], fx["x_rev"], 2e-6)


@pytest.mark.parametrize("axis", [1, 2, 3])
def test_gather_bit_exact(axis):
    from cwfa_amd import ops
    g = torch.Generator().manual_seed(axis)
    x = torch.randn(2, 6, 9, 11, generator=g)
    perm = torch.randperm(x.shape[axis], generator=g)
    assert torch.equal(ops.gather(x.cuda(), perm.cuda(), axis).cpu(), x.index_select(axis, perm))


def test_haar2d_then_haar1d_is_the_3d_tile():
    """north_star's '3-D 2x2x2 Haar' = FrEIA 2-D Haar composed with the depth Haar."""
    from cwfa_amd import ops
    from oracle import cwfa_oracle as O
    x = torch.randn(1, 8, 16, 16, generator=torch.Generator().manual_seed(5))
    a = ops.haar2d(ops.haar1d(x.cuda(), False), False, True, 0.5)
    b = O.haar2d(O.haar1d(x, False)[0], False, True, 1.0)[0]
    assert_close(a, b, 2e-6)


@pytest.mark.parametrize("shape", [(1, 8, 16, 16), (2, 6, 10, 14), (1, 4, 6, 20), (3, 2, 2, 2)])
@pytest.mark.parametrize("obw", [True, False])
def test_haar3d_is_the_two_launch_composition_bit_exact(shape, obw):
    """cwfa_haar3d_*: one pass over 2x2x2 tiles == haar2d(haar1d(x)) bit for bit (both channel orders, W % 4 != 0 included),
    against the oracle, and back."""
    from cwfa_amd import ops
    from oracle import cwfa_oracle as O
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape)))
    y = ops.haar3d(x.cuda(), False, obw, 0.5)
    assert torch.equal(y, ops.haar2d(ops.haar1d(x.cuda(), False), False, obw, 0.5))
    assert_close(y, O.haar2d(O.haar1d(x, False)[0], False, obw, 1.0)[0], 2e-6)
    xr = ops.haar3d(y, True, obw, 0.5)
    assert torch.equal(xr, ops.haar1d(ops.haar2d(y, True, obw, 0.5), True))
    assert_close(xr, x, 2e-6)
    big = torch.randn(shape[0], shape[1] + 2, shape[2], shape[3], device="cuda")         # channel-sliced input (batch stride)
    assert torch.equal(ops.haar3d(big[:, 2:], False, obw, 0.5), ops.haar3d(big[:, 2:].contiguous(), False, obw, 0.5))


def test_haar3d_full_size_round_trip():
    from cwfa_amd import ops
    x = torch.randn(1, 96, 512, 512, device="cuda")
    y = ops.haar3d(x)
    assert y.shape == (1, 384, 256, 256)
    assert_close(ops.haar3d(y, True), x, 2e-6)
    assert abs(float(y.double().pow(2).sum() / x.double().pow(2).sum()) - 1.0) < 1e-6       # orthonormal with fac = 0.5


# ------------------------------------------------------------------------------------------------ convolutions
@pytest.mark.parametrize("cfg", [
    # (B, Cin, H, W, Cout, ks)
    (1, 3, 5, 7, 2, 3), (2, 29, 20, 33, 6, 3), (1, 8, 16, 32, 64, 3), (1, 64, 33, 70, 64, 3), (2, 64, 16, 16, 96, 3),
    (1, 20, 17, 40, 130, 3), (1, 64, 18, 34, 64, 1), (2, 12, 9, 31, 8, 1), (1, 70, 16, 32, 200, 1),
    (1, 6, 23, 41, 6, 7), (1, 10, 16, 38, 40, 7), (1, 256, 16, 32, 128, 3),
    (2, 40, 9, 36, 24, 1), (1, 64, 10, 64, 64, 1),          # 1x1 on 16-byte aligned rows: vector-staged kernels
])
def test_conv2d_vs_torch_cpu(cfg):
    """Direct implicit-GEMM kernels: 1x1 and 7x7 always, 3x3 with the Winograd path switched off (it is the default)."""
    from cwfa_amd import ops
    ops.set_option("winograd_min_cout", 1 << 20)
    try:
        _conv2d_case(cfg)
    finally:
        ops.set_option("winograd_min_cout", 1)


def _conv2d_case(cfg):
    from cwfa_amd import ops
    B, Cin, H, W, Cout, ks = cfg
    g = torch.Generator().manual_seed(hash(cfg) % 1000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    pc = ops.pack_conv_weight(w.cuda())
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=ks // 2)
    assert_close(ops.conv2d(x.cuda(), pc, bias=b.cuda()), ref, 2e-6, "plain")
    y = ops.conv2d(x.cuda(), pc, bias=b.cuda(), act="elu", residual=res.cuda(), act2="elu")
    assert_close(y, torch.nn.functional.elu(torch.nn.functional.elu(ref) + res.double()), 3e-6, "elu+res+elu")
    alpha = torch.tensor([0.2])
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)
    add = torch.randn(B, Cin, H, W, generator=g)
    xin = x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + add.double()
    ref2 = torch.nn.functional.prelu(torch.nn.functional.conv2d(xin, w.double(), None, padding=ks // 2), alpha.double())
    y2 = ops.conv2d(x.cuda(), pc, act="prelu", prelu_alpha=alpha.cuda(), in_scale=sc.cuda(), in_shift=sh.cuda(),
                    in_add=add.cuda())
    assert_close(y2, ref2, 3e-6, "input affine + add + prelu")


@pytest.mark.parametrize("shape", [(1, 16, 32), (2, 19, 45), (1, 5, 7), (1, 64, 64)])
def test_subnet_layer_fused_vs_torch_cpu(shape):
    """y = ELU(conv1x1(ELU(conv3x3(x)+b3)) + b1 + x) in one launch (networks.py:624-631,660-665), ragged tiles."""
    from cwfa_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(B, 64, H, W, generator=g)
    w3, b3 = torch.randn(64, 64, 3, 3, generator=g) / 24, torch.randn(64, generator=g) * 0.1
    w1, b1 = torch.randn(64, 64, 1, 1, generator=g) / 8, torch.randn(64, generator=g) * 0.1
    F = torch.nn.functional
    xd = x.double()
    ref = F.elu(F.conv2d(F.elu(F.conv2d(xd, w3.double(), b3.double(), padding=1)), w1.double(), b1.double()) + xd)
    for min_cout in (1, 1 << 20):            # Winograd fused kernel (default) and the direct fused kernel
        ops.set_option("winograd_min_cout", min_cout)
        try:
            y = ops.subnet_layer(x.cuda(), ops.pack_conv_weight(w3.cuda()), b3.cuda(), ops.pack_1x1_panel(w1.cuda()),
                                 b1.cuda())
        finally:
            ops.set_option("winograd_min_cout", 1)
        assert_close(y, ref, 3e-6, f"fused layer, winograd_min_cout={min_cout}")
    # the unfused two-launch form agrees too
    h = ops.conv2d(x.cuda(), ops.pack_conv_weight(w3.cuda()), bias=b3.cuda(), act="elu")
    y2 = ops.conv2d(h, ops.pack_conv_weight(w1.cuda()), bias=b1.cuda(), residual=x.cuda(), act2="elu")
    assert_close(y2, ref, 3e-6)


@pytest.mark.parametrize("cfg", [
    # (B, Cin, H, W, Cout): 3x3 convs through the Winograd F(2,3) kernels (the default for every 3x3), all three tilings
    (1, 64, 16, 64, 64), (2, 29, 21, 37, 48), (1, 8, 9, 130, 40), (1, 70, 7, 63, 130), (1, 256, 8, 64, 128), (1, 6, 12, 66, 256),
    (2, 29, 21, 37, 6), (1, 64, 16, 64, 24), (1, 5, 3, 3, 32), (2, 40, 13, 132, 96), (1, 16, 4, 64, 65),
])
def test_conv3x3_winograd_vs_torch_cpu(cfg):
    from cwfa_amd import ops
    B, Cin, H, W, Cout = cfg
    F = torch.nn.functional
    g = torch.Generator().manual_seed(Cin * Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    alpha = torch.tensor([0.2])
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)
    add = torch.randn(B, Cin, H, W, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xin = x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + add.double()
    lin_pro = F.conv2d(xin, w.double(), b.double(), padding=1)
    ref_pro = F.prelu(lin_pro, alpha.double())
    want = {"plain": ref, "elu": F.elu(ref), "prelu": F.prelu(ref, alpha.double()),
            "res_prelu": F.prelu(ref + res.double(), alpha.double()), "generic": F.relu(F.gelu(ref) + res.double()),
            "pro_prelu": ref_pro, "pro_generic": F.elu(lin_pro)}
    # Cout > 64: the 1-D F(2,3) W128 tiling (default) and the opt-in 2-D F(2x2,3x3) kernel
    for two_d in ((0, 1) if Cout > 64 else (0,)):
        _winograd_case(ops, two_d, x, w, b, res, alpha, sc, sh, add, want)


def _winograd_case(ops, two_d, x, w, b, res, alpha, sc, sh, add, want):
    ops.set_option("winograd_min_cout", 1)
    ops.set_option("winograd_2d", two_d)
    try:
        pc = ops.pack_conv_weight(w.cuda())
        got = {
            "plain": ops.conv2d(x.cuda(), pc, bias=b.cuda()),
            "elu": ops.conv2d(x.cuda(), pc, bias=b.cuda(), act="elu"),
            "prelu": ops.conv2d(x.cuda(), pc, bias=b.cuda(), act="prelu", prelu_alpha=alpha.cuda()),
            "res_prelu": ops.conv2d(x.cuda(), pc, bias=b.cuda(), residual=res.cuda(), act2="prelu", prelu_alpha=alpha.cuda()),
            "generic": ops.conv2d(x.cuda(), pc, bias=b.cuda(), act="gelu", residual=res.cuda(), act2="relu"),
            "pro_prelu": ops.conv2d(x.cuda(), pc, bias=b.cuda(), act="prelu", prelu_alpha=alpha.cuda(), in_scale=sc.cuda(),
                                    in_shift=sh.cuda(), in_add=add.cuda()),
            "pro_generic": ops.conv2d(x.cuda(), pc, bias=b.cuda(), act="elu", in_scale=sc.cuda(), in_shift=sh.cuda(),
                                      in_add=add.cuda()),
        }
    finally:
        ops.set_option("winograd_2d", ops.WINOGRAD_2D_DEFAULT)
    for k in want:
        assert_close(got[k], want[k], 5e-6, f"winograd (2-D {two_d}) {k}")


def test_conv2d_generic_epilogue_combo():
    """A combination without a specialised epilogue (GELU -> +residual -> ReLU) takes the runtime path."""
    from cwfa_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 10, 13, 37, generator=g)
    w, b = torch.randn(40, 10, 3, 3, generator=g) * 0.1, torch.randn(40, generator=g)
    res = torch.randn(2, 40, 13, 37, generator=g)
    F = torch.nn.functional
    ref = F.relu(F.gelu(F.conv2d(x.double(), w.double(), b.double(), padding=1)) + res.double())
    y = ops.conv2d(x.cuda(), ops.pack_conv_weight(w.cuda()), bias=b.cuda(), act="gelu", residual=res.cuda(), act2="relu")
    assert_close(y, ref, 3e-6)


@pytest.mark.parametrize("wd", [13, 16])
def test_conv_transpose_as_pixel_shuffle(wd):
    from cwfa_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 24, 9, wd, generator=g)
    w = torch.randn(24, 10, 2, 2, generator=g) * 0.2
    b = torch.randn(10, generator=g)
    pc = ops.pack_conv_weight(w.cuda(), transposed=True)
    ref = torch.nn.functional.conv_transpose2d(x.double(), w.double(), b.double(), stride=2)
    assert_close(ops.conv2d(x.cuda(), pc, bias=b.cuda()), ref, 2e-6)


@pytest.mark.parametrize("cfg", [(1, 6, 9, 11, 4), (2, 8, 16, 40, 32), (1, 12, 8, 32, 32), (1, 5, 3, 3, 3), (1, 7, 20, 70, 32),
                                 (1, 6, 9, 33, 40)])
def test_conv3d_1k1_vs_torch_cpu(cfg):
    from cwfa_amd import ops
    B, D, H, W, K = cfg
    g = torch.Generator().manual_seed(K)
    x = torch.randn(B, D, H, W, generator=g)
    w1, b1 = torch.randn(K, 1, 3, 3, 3, generator=g) * 0.3, torch.randn(K, generator=g) * 0.1
    w2, b2 = torch.randn(1, K, 3, 3, 3, generator=g) * 0.1, torch.randn(1, generator=g)
    a = torch.tensor([0.25])
    F = torch.nn.functional
    v = x.double().permute(0, 2, 3, 1).unsqueeze(1)
    v = F.conv3d(F.prelu(F.conv3d(v, w1.double(), b1.double(), padding=1), a.double()), w2.double(), b2.double(), padding=1)
    ref = v[:, 0].permute(0, 3, 1, 2)
    y = ops.conv3d_1k1(x.cuda(), w1.cuda(), b1.cuda(), a.cuda(), w2.cuda(), b2.cuda())
    assert_close(y, ref, 5e-6)


def _conv3d_ref64(x, w1, b1, a, w2, b2):
    F = torch.nn.functional
    v = x.double().permute(0, 2, 3, 1).unsqueeze(1)
    v = F.conv3d(F.prelu(F.conv3d(v, w1.double(), b1.double(), padding=1), a.double()), w2.double(), b2.double(), padding=1)
    return v[:, 0].permute(0, 3, 1, 2)


@pytest.mark.parametrize("cfg", [(1, 6, 9, 11, 4, 0.25), (2, 8, 16, 40, 32, 0.25), (1, 12, 8, 32, 32, -0.5), (1, 5, 3, 3, 3, 1.5),
                                 (1, 7, 20, 70, 32, 0.0), (1, 1, 1, 1, 1, 0.25), (1, 48, 33, 65, 32, 0.25), (2, 3, 29, 31, 17, 1.0)])
def test_conv3d_1k1_split_is_fp32_accurate(cfg):
    """networks.py:221-225,239 on the split-bf16 kernel (csrc/conv3d_split.hip): depth ring over chunk borders (D > the depth
    chunk), ragged tiles, every PReLU slope regime, K < 32 -- against float64 torch under the SAME bound as the fp32 MFMA
    kernel; then the bf16 configuration (BASELINE.json configs[4]) under its restated bound."""
    from cwfa_amd import ops
    B, D, H, W, K, alpha = cfg
    g = torch.Generator().manual_seed(K + D)
    x = torch.randn(B, D, H, W, generator=g)
    w1, b1 = torch.randn(K, 1, 3, 3, 3, generator=g) * 0.3, torch.randn(K, generator=g) * 0.1
    w2, b2 = torch.randn(1, K, 3, 3, 3, generator=g) * 0.1, torch.randn(1, generator=g)
    a = torch.tensor([alpha])
    ref = _conv3d_ref64(x, w1, b1, a, w2, b2)
    args = [t.cuda() for t in (x, w1, b1, a, w2, b2)]
    y32 = ops.conv3d_1k1(*args)
    ops.set_precision("split_bf16")
    try:
        y = ops.conv3d_1k1(*args)
        ops.set_precision("bf16")
        yb = ops.conv3d_1k1(*args)
    finally:
        ops.set_precision("fp32")
    assert_close(y, ref, 5e-6, "split")
    assert_close(yb, ref, 2e-2, "bf16")
    if K * D * H * W > 64:
        assert not torch.equal(y, y32), "the split mode must run its own kernel"
        assert not torch.equal(yb, y), "the bf16 mode must differ from the split mode"


def test_conv3d_1k1_split_step0_shape_vs_fp32_kernel():
    """The finest condition net's 3-D stage at its real size (48 x 512 x 512, K = 32): split-bf16 kernel against the fp32 MFMA
    kernel (itself pinned to float64 above) -- every depth chunk, tile border and the XCD-spread grid."""
    from cwfa_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 48, 512, 512, generator=g).cuda()
    w1, b1 = (torch.randn(32, 1, 3, 3, 3, generator=g) * 0.3).cuda(), (torch.randn(32, generator=g) * 0.1).cuda()
    w2, b2 = (torch.randn(1, 32, 3, 3, 3, generator=g) * 0.1).cuda(), torch.randn(1, generator=g).cuda()
    a = torch.tensor([0.25]).cuda()
    ref = ops.conv3d_1k1(x, w1, b1, a, w2, b2)
    ops.set_precision("split_bf16")
    try:
        y = ops.conv3d_1k1(x, w1, b1, a, w2, b2)
    finally:
        ops.set_precision("fp32")
    assert_close(y, ref, 5e-6)


# ------------------------------------------------------------------------------------------------ blocks
def _block(bname, cl, fx):
    from cwfa_amd import networks as N
    from cwfa_amd.FrEIA import modules as Fm
    N.networks_n_chans = 8
    cls = {"CAT": Fm.ConditionalAffineTransform, "GLOW": Fm.GLOWCouplingBlock, "RNVP": Fm.RNVPCouplingBlock,
           "GIN": Fm.GINCouplingBlock, "NICE": Fm.NICECouplingBlock, "ONESIDED": Fm.AffineCouplingOneSided}[bname]
    kw = {"subnet_constructor": N.wavelet_flow_subnetwork2D}
    if bname != "NICE":
        kw.update(clamp=1.5, clamp_activation=cl)
    blk = cls([tuple(fx["x"].shape[1:])], dims_c=[tuple(fx["c"].shape[1:])], **kw)
    blk.load_state_dict(sd_of(fx))
    return blk.eval().cuda()


@pytest.mark.parametrize("name", names("g04_"))
def test_coupling_blocks_golden(name):
    fx = load_golden(name)
    _, bname, cl = name.split("_")
    blk = _block(bname, cl, fx)
    x, c = cu(fx["x"]), (cu(fx["c"]),)
    for rev, key in ((False, "fwd"), (True, "rev")):
        (y,), j = blk((x,), c=c, rev=rev)
        assert_close(y, fx["y_" + key], TOL, f"{name} y {key}")
        j = j if torch.is_tensor(j) else torch.full((x.shape[0],), float(j))
        if np.abs(fx["jac_" + key]).max() > 0:
            assert_close(j, fx["jac_" + key], TOL, f"{name} jac {key}")
        else:
            assert float(j.abs().max()) == 0
    (y,), jf = blk((x,), c=c, rev=False)
    (xr,), jr = blk((y,), c=c, rev=True)
    assert_close(xr, fx["x"], 1e-5, "round trip")


@pytest.mark.parametrize("name", names("g05_ai1_"))
def test_all_in_one_golden(name):
    from cwfa_amd import networks as N
    from cwfa_amd.FrEIA import modules as Fm
    fx = load_golden(name)
    N.networks_n_chans = 8
    cond = fx["c"].size > 0
    blk = Fm.AllInOneBlock([tuple(fx["x"].shape[1:])], dims_c=[tuple(fx["c"].shape[1:])] if cond else [],
                           subnet_constructor=N.wavelet_flow_subnetwork2D, gin_block="gin1" in name)
    blk.load_state_dict(sd_of(fx))
    blk = blk.eval().cuda()
    c = (cu(fx["c"]),) if cond else ()
    for rev, key in ((False, "fwd"), (True, "rev")):
        (y,), j = blk((cu(fx["x"]),), c=c, rev=rev)
        assert_close(y, fx["y_" + key], TOL, f"{name} y {key}")
        _assert_ai1_jac(j, fx["jac_" + key], "gin1" in name, f"{name} jac {key}")


def _assert_ai1_jac(j, ref, gin, what):
    """GIN: the log-det is sum(s - mean(s)) = 0; the reference returns its fp32 rounding noise (~1e-6, all_in_one_block.py:
    218-224), the HIP path an exact 0 -> absolute bound instead of a relative one."""
    if gin:
        assert float((j.cpu().double() - torch.as_tensor(ref).double()).abs().max()) <= 1e-4, what
    else:
        assert_close(j, ref, TOL, what)


def test_mean_volume_cache_and_output_step_golden():
    """SURVEY.md 8f row 3 on the HIP path: mean-volume cache bands from the forward pyramid (CWFA.py:646-655; pyramid levels
    through the fused forward chain with zero conditions as evaluate_INN_forward runs them) and the output step
    (CWFA.py:1035-1044) -- bit-exact against the reference's own tensors."""
    from cwfa_amd import CWFA, ops
    fx = load_golden("g19_meanvol")
    levels = [cu(fx[f"level_{i}"]) for i in range(3)]
    for i, v in enumerate(CWFA.mean_volume_cache(levels)):
        assert torch.equal(v.cpu(), T(fx[f"cache_{i}"])), i
    lv = cu(fx["gt_volume"])
    for i in range(1, 3):                                   # the low band of the depth Haar is the next level
        y = ops.haar1d(lv, False)
        lv = y[:, :y.shape[1] // 2].contiguous()
        assert_close(lv, fx[f"level_{i}"], 2e-6, f"level {i}")
    fx = load_golden("g19_denorm")
    pred = CWFA.denormalise_prediction(cu(fx["stored0"]), T(fx["std_vols"]), T(fx["mean_vols"]))
    assert torch.equal(pred.cpu(), T(fx["vol_out_pred"]))
    gt = CWFA.denormalise_ground_truth(cu(fx["gt0"]), T(fx["std_vols"]), T(fx["mean_vols"]))
    assert torch.equal(gt.cpu(), T(fx["vol_out"]))


@pytest.mark.parametrize("name", names("g18_ai1_opt_"))
def test_all_in_one_options_golden(name):
    """AllInOneBlock outside CWFA's defaults (all_in_one_block.py:122-196): soft (SO(C)) permutation and learned householder
    reflections as dense 1x1 mixes, reverse permutation, GIN, SIGMOID / EXP global affine."""
    from cwfa_amd.FrEIA import modules as Fm
    fx = load_golden(name)
    kw = {"soft": dict(permute_soft=True), "house": dict(learned_householder_permutation=2),
          "revperm": dict(reverse_permutation=True), "gin": dict(gin_block=True),
          "sigmoid": dict(global_affine_type="SIGMOID", global_affine_init=0.7), "exp": dict(global_affine_type="EXP", global_affine_init=1.3),
          "soft_rev_gin": dict(permute_soft=True, reverse_permutation=True, gin_block=True)}[name[len("g18_ai1_opt_"):]]

    class Sub(torch.nn.Module):
        def __init__(self, cin, cout):
            super().__init__()
            self.c = torch.nn.Conv2d(cin, cout, 3, padding=1)
            self._pc = None

        def forward(self, t):
            from cwfa_amd import ops
            if self._pc is None:
                self._pc = ops.pack_conv_weight(self.c.weight)
            return ops.conv2d(t, self._pc, bias=self.c.bias)

    blk = Fm.AllInOneBlock([tuple(fx["x"].shape[1:])], dims_c=[tuple(fx["c"].shape[1:])], subnet_constructor=Sub, **kw)
    assert sorted(blk.state_dict()) == sorted(sd_of(fx))
    blk.load_state_dict(sd_of(fx))
    blk = blk.cuda()
    for rev, key in ((False, "fwd"), (True, "rev")):
        (y,), j = blk((cu(fx["x"]),), c=(cu(fx["c"]),), rev=rev)
        assert_close(y, fx["y_" + key], TOL, f"{name} y {key}")
        _assert_ai1_jac(j, fx["jac_" + key], "gin" in name, f"{name} jac {key}")


def test_actnorm_golden():
    from cwfa_amd.FrEIA import modules as Fm
    fx = load_golden("g06_actnorm")
    an = Fm.ActNorm([tuple(fx["x"].shape[1:])]).cuda()
    (y,), j = an((cu(fx["x"]),), rev=False)            # data-dependent init on the first batch
    assert_close(an.scale, fx["sd/scale"], 1e-5)
    assert_close(an.bias, fx["sd/bias"], 1e-5)
    assert_close(y, fx["y_fwd"], 1e-5)
    assert_close(j, fx["jac_fwd"], 1e-5)
    (xr,), jr = an((cu(fx["z"]),), rev=True)
    assert_close(xr, fx["x_rev"], 1e-5)
    assert_close(jr, fx["jac_rev"], 1e-5)
    an2 = Fm.ActNorm([tuple(fx["x"].shape[1:])])
    an2.load_state_dict(sd_of(fx))
    assert an2.init_on_next_batch is False


def test_concat_golden():
    """Concat (graph_topology.py:92-152): forward through the strided plane-copy kernel, reverse hands out views."""
    from cwfa_amd.FrEIA import modules as Fm
    fx = load_golden("g16_concat")
    dims = [tuple(fx[f"x{i}"].shape[1:]) for i in range(3)]
    cat = Fm.Concat(dims, dim=0)
    assert tuple(cat.output_dims(dims)[0]) == tuple(int(v) for v in fx["out_dims"])
    (y,), j = cat([cu(fx["x0"]), cu(fx["x1"]), cu(fx["x2"])], rev=False)
    assert torch.equal(y.cpu(), T(fx["y_fwd"])) and j == float(fx["jac_fwd"])
    parts, jr = cat((cu(fx["z"]),), rev=True)
    assert len(parts) == 3 and all(torch.equal(p_.cpu(), T(fx[f"r{i}"])) for i, p_ in enumerate(parts)) and jr == float(fx["jac_rev"])
    # channel-sliced (strided) inputs and the deprecated aliases
    big = cu(fx["z"])
    (y2,), _ = cat([big[:, :3], big[:, 3:5], big[:, 5:]], rev=False)
    assert torch.equal(y2, big)
    with pytest.warns(DeprecationWarning):
        Fm.ConcatChannel(dims)


def test_fixed1x1conv_golden():
    """Fixed1x1Conv (fixed_transforms.py:95-133): dense 1x1 on the MFMA conv kernel, both directions, log-det."""
    from cwfa_amd.FrEIA import modules as Fm
    fx = load_golden("g17_fixed1x1conv")
    m = Fm.Fixed1x1Conv([tuple(fx["x"].shape[1:])], M=T(fx["M"])).cuda()
    for k, v in sd_of(fx).items():
        assert_close(m.state_dict()[k], v, 1e-6, k)
    (yf,), jf = m((cu(fx["x"]),), rev=False)
    (yr,), jr = m((cu(fx["x"]),), rev=True)
    assert_close(yf, fx["y_fwd"], 2e-6, "forward")
    assert_close(yr, fx["y_rev"], 2e-5, "reverse")
    assert abs(float(jf) - float(fx["jac_fwd"])) <= 1e-5 * abs(float(fx["jac_fwd"]))
    assert abs(float(jr) - float(fx["jac_rev"])) <= 1e-5 * abs(float(fx["jac_rev"]))
    (xb,), _ = m((yf,), rev=True)
    assert_close(xb, fx["x"], 2e-5, "round trip")


@pytest.mark.parametrize("name", names("g07_"))
def test_subnets_golden(name):
    from cwfa_amd import networks as N
    fx = load_golden(name)
    N.networks_n_chans = int(fx["n_ch"])
    ctor = N.wavelet_flow_subnetwork2D_first if "first1" in name else N.wavelet_flow_subnetwork2D
    net = ctor(int(fx["c_in"]), int(fx["c_out"]))
    net.load_state_dict(sd_of(fx))
    assert_close(net.cuda()(cu(fx["x"])), fx["y"], TOL, name)


@pytest.mark.parametrize("name", names("g08_"))
def test_omega_golden(name):
    from cwfa_amd import networks as N
    fx = load_golden(name)
    net = N.cond_network(29, int(fx["c_out"]), 1, 5, [], int(fx["chans3d"]))
    net.load_state_dict(sd_of(fx))
    assert_close(net.eval().cuda()(cu(fx["x"]))[-1], fx["y"], TOL, name)


# ------------------------------------------------------------------------------------------------ whole steps
def _step(name, fx):
    from test_host_logic import build_step
    bt, ix = name.split("_")[2], int(name[-1])
    _, g = build_step(bt, ix)
    g.load_state_dict(sd_of(fx))
    return bt, g.eval().cuda()


@pytest.mark.parametrize("name", [n for n in names("g04_") if n.split("_")[1] in ("GLOW", "ONESIDED")] + names("g05_ai1_gin0") +
                         [n for n in names("g09_") if "GLOW" in n or "AI1" in n])
@torch.no_grad()            # the fused epilogue is the INFERENCE form; with a graph being recorded the blocks take cwfa_amd.autograd's nodes
def test_coupling_in_the_conv_epilogue_golden(name, monkeypatch):
    """Split precision: the last convolution of the sub-network applies the coupling from its accumulators (s, t never reach
    memory; cwfa_conv3x3_split_couple_f32) -- same golden vectors, same bound as the unfused path."""
    from cwfa_amd import ops
    fx = load_golden(name)
    calls = []
    real = ops.conv3x3_couple
    monkeypatch.setattr(ops, "conv3x3_couple", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    ops.set_precision("split_bf16")
    try:
        if name.startswith("g04_"):
            _, bname, cl = name.split("_")
            blk = _block(bname, cl, fx)
            x, c = cu(fx["x"]), (cu(fx["c"]),)
        elif name.startswith("g05_"):
            from cwfa_amd import networks as N
            from cwfa_amd.FrEIA import modules as Fm
            N.networks_n_chans = 8
            cond = fx["c"].size > 0
            blk = Fm.AllInOneBlock([tuple(fx["x"].shape[1:])], dims_c=[tuple(fx["c"].shape[1:])] if cond else [],
                                   subnet_constructor=N.wavelet_flow_subnetwork2D)
            blk.load_state_dict(sd_of(fx))
            blk = blk.eval().cuda()
            x, c = cu(fx["x"]), ((cu(fx["c"]),) if cond else ())
        else:
            bt, g = _step(name, fx)
            c = [cu(fx["c0"]), cu(fx["c1"])]
            (z, low), jf = g(cu(fx["x"]), c=c)
            assert_close(z, fx["z"], TOL, "z")
            assert_close(jf, fx["jac_fwd"], TOL, "jac fwd")
            xr, jr = g([cu(fx["z"]), cu(fx["low"])], c=c, rev=True)
            assert_close(xr, fx["x_rev"], TOL, "x_rev")
            assert_close(jr, fx["jac_rev"], TOL, "jac rev")
            assert len(calls) == (16 if bt == "GLOW" else 8), len(calls)
            return
        for rev, key in ((False, "fwd"), (True, "rev")):
            (y,), j = blk((x.clone(),), c=c, rev=rev)
            assert_close(y, fx["y_" + key], TOL, f"{name} y {key}")
            assert_close(j, fx["jac_" + key], TOL, f"{name} jac {key}")
        assert len(calls) == (4 if "GLOW" in name else 2), len(calls)
    finally:
        ops.set_precision("fp32")


@pytest.mark.parametrize("block", ["CAT", "GLOW"])
def test_actnorm_stays_in_the_step_plan(block):
    """A step graph with ActNorm nodes (invertible_resnet.py:27-85; the reference keeps reset_ActNorm for such graphs,
    networks.py:137-163) still lowers to a plan: ActNorm is a per-channel stage of the fused chain.  The first forward call
    initialises the ActNorms from their input batch (node walk); afterwards plan == node walk in both directions."""
    from cwfa_amd import networks as N, INN_utils
    from cwfa_amd.FrEIA import framework as Ff, modules as Fm
    N.networks_n_chans = 8
    torch.manual_seed(3)
    np.random.seed(3)
    D, H, W, B = 8, 8, 64, 2
    C_ = D // 2
    nodes = [Ff.InputNode(D, H, W, name="input")]
    nodes.append(Ff.Node(nodes[-1], INN_utils.HaarTransform1D, {"order_by_wavelet": True}, name="haar"))
    split = Ff.Node(nodes[-1], Fm.Split, {"section_sizes": (C_, C_), "dim": 0}, name="split")
    nodes.append(split)
    cond = Ff.ConditionNode(C_, H, W, name="cond")
    nodes.append(cond)
    nodes.append(Ff.Node(split.out1, Fm.ConditionalAffineTransform, {"subnet_constructor": N.wavelet_flow_subnetwork2D},
                         conditions=[cond], name="cat0"))
    nodes.append(Ff.Node(nodes[-1], Fm.ActNorm, {}, name="an0"))
    nodes.append(Ff.Node(nodes[-1], Fm.PermuteRandom, {"seed": 1}, name="p1"))
    blk = Fm.ConditionalAffineTransform if block == "CAT" else Fm.GLOWCouplingBlock
    nodes.append(Ff.Node(nodes[-1], blk, {"subnet_constructor": N.wavelet_flow_subnetwork2D}, conditions=[cond], name="b1"))
    nodes.append(Ff.Node(nodes[-1], Fm.ActNorm, {}, name="an1"))
    nodes.append(Ff.Node(nodes[-1], INN_utils.PermuteDim, {"seed": 2}, name="p2"))
    nodes.append(Ff.OutputNode(nodes[-1], name="z"))
    nodes.append(Ff.OutputNode(split.out0, name="low"))
    g = Ff.GraphINN(nodes).cuda()
    assert g._plan is not None and sum(k == "act" for k, _ in g._plan.chain) == 2
    gen = torch.Generator().manual_seed(4)
    x = (2.0 * torch.randn(B, D, H, W, generator=gen) + 0.5).cuda()
    c = [torch.randn(B, C_, H, W, generator=gen).cuda()]
    assert g._plan.needs_walk()
    (z0, low0), j0 = g(x, c=c)                                   # data-dependent initialisation: node walk
    assert not g._plan.needs_walk()
    an = [m for m in g.module_list if isinstance(m, Fm.ActNorm)]
    assert all(float(m.scale.abs().max()) > 0 for m in an)
    (z, low), j = g(x, c=c)                                      # the plan
    assert_close(z, z0, 1e-5, "plan vs walk: z")
    assert torch.equal(low, low0)
    assert_close(j, j0, 1e-5, "plan vs walk: log-det")
    xr, jr = g([z, low], c=c, rev=True)
    assert_close(xr, x, 1e-5, "round trip")
    assert_close(jr, -j, 1e-5, "log-det of the inverse")
    plan, g._plan = g._plan, None
    try:
        xw, jw = g([z, low], c=c, rev=True)
    finally:
        g._plan = plan
    assert_close(xr, xw, 1e-5, "plan vs walk: inverse")
    assert_close(jr, jw, 1e-5)


@pytest.mark.parametrize("cfg", [(1, 24, 64, 40, 72, "ATAN", 1.0, False), (2, 48, 64, 33, 50, "TANH", 0.1, True),
                                 (1, 3, 16, 17, 31, "SIGMOID", 1.0, True), (1, 33, 40, 24, 64, "ATAN", 1.0, False),
                                 (2, 64, 64, 16, 32, "NONE", 0.5, True)])
def test_conv3x3_couple_vs_unfused(cfg):
    """cwfa_conv3x3_split_couple_f32 against conv (fp64 torch) + the coupling formula, both row interleavings (n <= 32 / > 32),
    ragged sizes, in place, log-det."""
    from cwfa_amd import ops
    B, n, cin, H, W, kind, pre, rev = cfg
    g = torch.Generator().manual_seed(n + H)
    w = torch.randn(2 * n, cin, 3, 3, generator=g) * (1.5 / (3 * cin ** 0.5))
    bias = torch.randn(2 * n, generator=g) * 0.1
    u = torch.randn(B, cin, H, W, generator=g)
    x = torch.randn(B, n, H, W, generator=g)
    a = torch.nn.functional.conv2d(u.double(), w.double(), bias.double(), padding=1) * pre
    clamp = 1.7
    sr = a[:, :n]
    s = {"ATAN": lambda v: clamp * 0.636 * torch.atan(v), "TANH": lambda v: clamp * torch.tanh(v),
         "SIGMOID": lambda v: clamp * 2. * (torch.sigmoid(v) - 0.5), "NONE": lambda v: clamp * v}[kind](sr)
    t = a[:, n:]
    ref = (x.double() - t) * torch.exp(-s) if rev else torch.exp(s) * x.double() + t
    jref = (-1 if rev else 1) * s.sum(dim=(1, 2, 3))
    ops.set_precision("split_bf16")
    try:
        bank = ops.pack_couple_weight(w.cuda(), bias.cuda())
        full = torch.randn(B, n + 5, H, W, generator=g).cuda()          # the active half as a channel slice of a larger tensor
        full[:, 5:] = x.cuda()
        ld = torch.zeros(B, dtype=torch.float64, device="cuda")
        out = torch.empty(B, n, H, W, device="cuda")
        ops.conv3x3_couple(u.cuda(), bank, full[:, 5:], out, kind, clamp, pre, rev, logdet=ld)
        assert_close(out, ref, 1e-5, "coupled output")
        assert_close(ld, jref, 1e-5, "log-det")
        head = full[:, :5].clone()
        ops.conv3x3_couple(u.cuda(), bank, full[:, 5:], full[:, 5:], kind, clamp, pre, rev)          # in place, no log-det
        assert torch.equal(full[:, 5:], out) and torch.equal(full[:, :5], head)
    finally:
        ops.set_precision("fp32")


@pytest.mark.parametrize("name", names("g09_"))
def test_flow_step_golden(name):
    fx = load_golden(name)
    bt, g = _step(name, fx)
    c = [cu(fx["c0"]), cu(fx["c1"])]
    (z, low), jf = g(cu(fx["x"]), c=c)
    assert_close(z, fx["z"], TOL, "z")
    assert torch.equal(low.cpu(), T(fx["low"])), "low-pass half must be bit-exact"
    assert_close(jf, fx["jac_fwd"], TOL, "jac fwd")
    xr, jr = g([cu(fx["z"]), cu(fx["low"])], c=c, rev=True)
    assert_close(xr, fx["x_rev"], TOL, "x_rev")
    assert_close(jr, fx["jac_rev"], TOL, "jac rev")
    x0, _ = g([torch.zeros_like(z), cu(fx["low"])], c=c, rev=True)
    assert_close(x0, fx["x_rev_z0"], TOL, "x_rev from z=0")
    assert g._plan is not None, "every CWFA step graph lowers to a plan"
    if True:
        # the fused plan (z=None: never read) and the node-by-node walk agree with each other and the reference
        x0n, _ = g([None, cu(fx["low"])], c=c, rev=True)
        assert torch.equal(x0n, x0)
        plan, g._plan = g._plan, None
        try:
            (zw, loww), jw = g(cu(fx["x"]), c=c)
            xw, jrw = g([cu(fx["z"]), cu(fx["low"])], c=c, rev=True)
        finally:
            g._plan = plan
        assert_close(zw, fx["z"], TOL, "walk z")
        assert_close(xw, fx["x_rev"], TOL, "walk x_rev")
        assert_close(jw, fx["jac_fwd"], TOL)
        assert_close(jrw, fx["jac_rev"], TOL)


def _pipeline(fx):
    from cwfa_amd import CWFA, networks as N
    S = int(fx["S"])
    D, H, W = fx["gt"].shape[1:]
    conv_inn, cond_nets = [], []
    for ix in range(S - 1):
        cn, inns = N.conditional_wavelet_flow([D, H, W], [1, 29, H, W], N.wavelet_flow_subnetwork2D,
                                              lambda: N.cond_network(29, D // 2 ** (ix + 1), ix + 1, S, [], 4),
                                              n_internal_ch=8, n_down_steps=ix + 1, use_permutations=True,
                                              block_type="CAT", n_blocks=4)
        inns[ix].load_state_dict(sd_of(fx, f"inn{ix}/"))
        cn.load_state_dict(sd_of(fx, f"omega{ix}/"))
        conv_inn.append(inns[ix].eval().cuda())
        cond_nets.append(cn.eval().cuda())
    return CWFA, conv_inn, cond_nets, S


def test_pipeline_golden():
    import argparse
    fx = load_golden("g10_pipeline")
    CWFA, conv_inn, cond_nets, S = _pipeline(fx)
    cond_input = (cu(fx["views"]) - float(fx["mean_imgs"])) / float(fx["std_imgs"])
    mean_cache = [cu(fx[f"mean_cache_{n}"]) for n in range(S - 1)]
    vols = CWFA.inverse_pass(conv_inn, cond_nets, cond_input, mean_cache, low=cu(fx["low"]), keep_all=True)
    for i, n in enumerate(range(S - 2, -1, -1)):
        assert_close(cond_nets[n](cond_input)[-1], fx[f"omega_{n}"], TOL, f"omega_{n}")
        assert_close(vols[i + 1], fx[f"up_{n}"], TOL, f"up_{n}")
    args = argparse.Namespace(INN_max_down_steps=S, force_all_steps_NF=0)
    stats = (torch.tensor(float(fx["mean_imgs"])).cuda(), torch.tensor(float(fx["std_imgs"])).cuda())
    losses, gt_cache, prior, logj = CWFA.evaluate_INN_forward(conv_inn, cond_nets, args, [args] * S, cu(fx["gt"]).clone(),
                                                              cu(fx["views"]), stats)
    for n in range(S):
        assert torch.equal(gt_cache[n].cpu(), T(fx[f"gt_cache_{n}"]))
    for n in range(S - 1):
        assert abs(float(losses[n]) - fx["losses"][n]) <= TOL * abs(fx["losses"][n])
        assert abs(float(prior[n]) - fx["prior"][n]) <= TOL * abs(fx["prior"][n])
        assert abs(float(logj[n]) - fx["logjac"][n]) <= TOL * abs(fx["logjac"][n]) + 1e-9
    # per-sample, per-step log-likelihoods (OOD score): against the oracle, and for a batch of one against -losses
    from oracle import cwfa_oracle as O
    ll = CWFA.step_log_likelihoods(conv_inn, cond_nets, args, cu(fx["gt"]).clone(), cu(fx["views"]), stats)
    assert ll.shape == (fx["gt"].shape[0], S - 1) and ll.dtype == torch.float64
    xo = T(fx["gt"])
    for n in range(S - 1):
        gi = conv_inn[n]
        axes = {i: (m.axis if hasattr(m, "axis") else 1) for i, m in enumerate(gi.module_list) if hasattr(m, "perm")}
        zc = torch.zeros(xo.shape[0], xo.shape[1] // 2, xo.shape[2], xo.shape[3])
        (zr, lowr), jr = O.flow_step({k: v.detach().cpu() for k, v in gi.state_dict().items()}, xo, [zc, zc], False, axes)
        assert_close(ll[:, n], O.step_log_likelihood(zr, jr, lowr[0].numel()), TOL, f"step LL {n}")
        xo = lowr
    one = CWFA.step_log_likelihoods(conv_inn, cond_nets, args, cu(fx["gt"])[:1].clone(), cu(fx["views"])[:1], stats)
    l1 = CWFA.evaluate_INN_forward(conv_inn, cond_nets, args, [args] * S, cu(fx["gt"])[:1].clone(), cu(fx["views"])[:1], stats)[0]
    for n in range(S - 1):
        assert abs(float(one[0, n]) + float(l1[n])) <= 1e-5 * abs(float(l1[n]))
    assert_close(one[0], ll[0], 1e-6, "score independent of the batch it is computed in")
    flags = CWFA.detect_ood(ll, 0, float(ll[:, 0].median()))
    assert flags.dtype == torch.bool and flags.shape == (ll.shape[0],)
    # NLL of CWFA.py:978 through the shard-sum form, single process
    x = cu(fx["gt"])
    cz = [torch.zeros(x.shape[0], 8, *x.shape[2:], device="cuda")] * 2
    nll, Z, jac = CWFA.nll_step(conv_inn[0], x, cz)
    ref = (0.5 * torch.norm(Z[0]) ** 2 - jac.mean()) / x.numel()          # CWFA.py:978: / upsampled_vol.numel()
    assert abs(float(nll) - float(ref)) <= 1e-5 * abs(float(ref))


# ------------------------------------------------------------------------------------------------ LRNN
@pytest.mark.parametrize("bias", [0, 1])
def test_unet_golden(bias):
    from cwfa_amd.unet import UNet
    fx = load_golden(f"g11_unet_bias{bias}")
    u = UNet(5, 4, depth=3, wf=3, drop_out=0, use_bias=bool(bias), skip_conn=True, up_mode="upconv", batch_norm=True)
    u.load_state_dict(sd_of(fx))
    u = u.cuda()
    x = cu(fx["x"])
    assert_close(u.eval()(x), fx["y_eval"], TOL, "eval")
    u.train()
    assert_close(u(x), fx["y_train"], TOL, "train (batch statistics)")
    assert_close(u(x[:1]), fx["y_train_b1"], TOL, "train B=1")


@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("prec", ["fp32", "split_bf16"])
def test_unet_with_live_dropout2d_vs_oracle(mode, prec):
    """``F.dropout2d(x, self.drop_out)`` of the UNet (unet.py:80,86: live in train AND eval mode) with the SAME uniform draws on both
    sides: the HIP path forms the factor (u >= p) / (1 - p) in cwfa_bn_finish_f32 and applies it on the next convolution's load
    side (skip tensor included); the oracle multiplies where the reference calls dropout2d.  Golden weights of g11, p = 0.3."""
    from cwfa_amd import ops, unet as U
    from oracle import cwfa_oracle as O
    fx = load_golden("g11_unet_bias1")
    net = U.UNet(5, 4, depth=3, wf=3, drop_out=0.3, use_bias=True, skip_conn=True, up_mode="upconv", batch_norm=True)
    net.load_state_dict(sd_of(fx))
    net = net.cuda()
    net.train() if mode == "train" else net.eval()
    x = torch.from_numpy(fx["x"])
    B = x.shape[0]
    chans = [d.block[0].out_channels for d in list(net.down_path)[:-1]] + [u.conv_block.block[0].out_channels for u in net.up_path]
    g = torch.Generator().manual_seed(9)
    draws = [torch.rand(B, c, generator=g) for c in chans]
    assert any(bool((d < 0.3).any()) for d in draws)
    flat = torch.cat([d.reshape(-1) for d in draws]).cuda()

    class Pool(U._DrawPool):                                   # the forward's one torch.rand launch replaced by the given draws
        def __init__(self, n, device):
            assert n == flat.numel()
            self.u, self.pos = flat, 0

    orig = U._DrawPool
    U._DrawPool = Pool
    ops.set_precision(prec)
    try:
        with torch.no_grad():
            y = net(x.cuda())
    finally:
        U._DrawPool = orig
        ops.set_precision("fp32")
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()} if mode == "eval" else sd_of(fx)
    with torch.no_grad():
        ref = O.unet(sd, x, depth=3, train=(mode == "train"), drop_u=draws, drop_p=0.3)
    assert_close(y, ref, TOL, f"UNet with dropout2d, {mode}, {prec}")


def test_drop_path_is_per_sample_bernoulli_scaling():
    """networks.drop_path (networks.py:370-385: x / keep * floor(keep + U) per sample): every sample is either dropped or scaled by
    exactly 1 / keep, the kept fraction is Bernoulli(keep), eval mode / p = 0 return the input itself."""
    from cwfa_amd import networks as N
    torch.manual_seed(4)
    x = torch.randn(400, 3, 4, 8).cuda()
    y = N.drop_path(x, 0.25, True)
    ratio = (y / x).reshape(400, -1)
    kept = ratio[:, 0] != 0
    assert torch.equal(y[~kept], torch.zeros_like(y[~kept]))
    assert torch.equal(y[kept], x[kept] * torch.tensor(1.0, device="cuda").div(0.75))
    assert 0.65 < float(kept.float().mean()) < 0.85                       # Bernoulli(0.75), n = 400: 6 sigma = 0.13
    assert N.drop_path(x, 0.25, False) is x and N.drop_path(x, 0.0, True) is x


def test_convnext_attention_golden():
    from cwfa_amd import networks as N
    fx = load_golden("g11_convnext")
    m = N.ConvNeXt(6, 10, drop_prob=0.05, size=16)
    m.load_state_dict(sd_of(fx))
    assert_close(m.eval().cuda()(cu(fx["x"])), fx["y"], TOL)
    fx = load_golden("g11_attention")
    a = N.GlobalAttention(6)
    a.load_state_dict(sd_of(fx))
    assert_close(a.cuda()(cu(fx["x"])), fx["y"], 1e-5)


def test_lrnn_small_and_full_golden():
    """Weights are regenerated from the seed (63.7 M parameters are not shipped; test_host_logic pins the RNG stream)."""
    from cwfa_amd import networks as N
    fs, ff = load_golden("g11_lrnn_small"), load_golden("g11_lrnn_full")
    torch.manual_seed(int(fs["seed_init"]))
    enc = N.Encoder(29, 6, 5, 64, True)
    enc.net.deconv[1].drop_out = 0            # dropout2d is stochastic in the reference (unet.py:80,86)
    gi = torch.Generator().manual_seed(int(fs["seed_input"]))
    x_small = torch.randn(2, 29, 16, 16, generator=gi)
    enc = enc.eval().cuda()
    assert_close(enc(x_small.cuda())[-1], fs["y_eval"], TOL, "small, eval")
    gi = torch.Generator().manual_seed(int(ff["seed_input"]))
    x_full = torch.randn(1, 29, 512, 512, generator=gi)
    mean_full = torch.randn(1, 6, 512, 512, generator=gi) * 0.1
    gl = torch.Generator().manual_seed(int(ff["seed_ln"]))
    with torch.no_grad():
        for cn in enc.net.conv3d:
            cn.m[1].weight.copy_((1 + 0.1 * torch.randn(cn.m[1].weight.shape, generator=gl)).cuda())
            cn.m[1].bias.copy_((0.1 * torch.randn(cn.m[1].bias.shape, generator=gl)).cuda())
    y = enc(x_full.cuda(), mean_full.cuda())[-1]
    assert_close(y[:, :, ::23, ::29], ff["y_sub"], TOL, "full, with mean volume (subsampled)")
    assert abs(float(y.double().sum()) - float(ff["y_sum"])) <= 1e-4 * float(ff["y_abs"])
    assert_close(enc(x_full.cuda())[-1][:, :, ::23, ::29], ff["y_nomean_sub"], TOL, "full, no mean volume")


@pytest.mark.parametrize("hw", [(40, 96), (24, 300)])
def test_cat_step_row_staged_chain_vs_oracle(hw):
    """Wide images take the row-staged chain kernels (W >= 64): compare a CAT step, both directions, with the oracle."""
    from cwfa_amd import CWFA, networks as N
    from oracle import cwfa_oracle as O
    H, W = hw
    torch.manual_seed(3)
    np.random.seed(3)
    cn, inns = N.conditional_wavelet_flow([12, H, W], [1, 29, H, W], N.wavelet_flow_subnetwork2D,
                                          lambda: N.cond_network(29, 6, 1, 3, [], 4), n_internal_ch=8, n_down_steps=1,
                                          use_permutations=True, block_type="CAT", n_blocks=4)
    g = inns[0].eval()
    with torch.no_grad():
        for p_ in g.parameters():
            if p_.requires_grad:
                p_.add_(0.05 * torch.randn(p_.shape))
    sd = {k: v.detach().clone() for k, v in g.state_dict().items()}
    axes = {i: (m.axis if hasattr(m, "axis") else 1) for i, m in enumerate(g.module_list) if hasattr(m, "perm")}
    gen = torch.Generator().manual_seed(W)
    x = torch.randn(2, 12, H, W, generator=gen)
    c = [torch.randn(2, 6, H, W, generator=gen), 0.3 * torch.randn(2, 6, H, W, generator=gen)]
    (z_ref, low_ref), j_ref = O.flow_step(sd, x, c, False, axes)
    x_ref, jr_ref = O.flow_step(sd, (z_ref, low_ref), c, True, axes)
    g = g.cuda()
    cc = [t.cuda() for t in c]
    sumsq = torch.zeros(1, dtype=torch.float64, device="cuda")
    (z, low), j = g(x.cuda(), c=cc, sumsq=sumsq)
    assert_close(z, z_ref, TOL, "z")
    assert torch.equal(low.cpu(), low_ref)
    assert_close(j, j_ref, TOL, "logdet")
    assert abs(float(sumsq) - float(z_ref.double().pow(2).sum())) <= 1e-5 * float(z_ref.double().pow(2).sum())
    xr, jr = g([z_ref.cuda(), low_ref.cuda()], c=cc, rev=True)
    assert_close(xr, x_ref, TOL, "x_rev")
    assert_close(jr, jr_ref, TOL, "logdet rev")
    x0, j0 = g([None, low_ref.cuda()], c=cc, rev=True, jac=False)
    assert j0 is None
    assert_close(x0, O.flow_step(sd, (torch.zeros_like(z_ref), low_ref), c, True, axes)[0], TOL, "x_rev z=0")


def test_full_size_step_roundtrip():
    """Config-3 finest step (C=48, 512x512): forward then inverse recovers x; log-dets are antisymmetric."""
    from cwfa_amd import CWFA
    torch.manual_seed(0)
    np.random.seed(0)
    conv_inn, cond_nets = CWFA.build_networks(96, 512, 2, with_lrnn=False)
    g = conv_inn[0]
    x = torch.randn(1, 96, 512, 512, device="cuda")
    c = [0.3 * torch.randn(1, 48, 512, 512, device="cuda"), 0.1 * torch.randn(1, 48, 512, 512, device="cuda")]
    (z, low), jf = g(x, c=c)
    xr, jr = g([z, low], c=c, rev=True)
    assert float((xr - x).abs().max()) <= 1e-4 * float(x.abs().max())
    assert abs(float(jf + jr)) <= 1e-4 * abs(float(jf))


# ------------------------------------------------------------------------------------------------ lenslet views (8f row 2)
@pytest.mark.parametrize("name", ["even", "odd", "wide"])
def test_extract_views_golden(name):
    from cwfa_amd import ops
    from cwfa_amd.XLFMDataset import XLFMDatasetFull
    fx = load_golden(f"g12_extract_views_{name}")
    img, coords, sub = torch.from_numpy(fx["image"]).cuda(), fx["coords"].tolist(), fx["sub"].tolist()
    assert np.array_equal(XLFMDatasetFull.extract_views(img, coords, sub).cpu().numpy(), fx["views"])
    got = XLFMDatasetFull.extract_views_normalized(img, coords, sub, float(fx["mean"]), float(fx["std"]))
    assert np.array_equal(got.cpu().numpy(), fx["normalized"])
    with pytest.raises(ValueError, match="outside the image"):
        ops.extract_views(img, [(-40, 3)], sub)


def test_extract_views_full_frame_vs_oracle():
    """The real geometry: 29 views of 512x512 out of a 2160x2160 frame (30 MB out), against the CPU oracle."""
    from cwfa_amd import ops
    from oracle import cwfa_oracle as O
    g = torch.Generator().manual_seed(29)
    img = torch.randn(1, 1, 2160, 2160, generator=g)
    coords = torch.randint(100, 2060, (29, 2), generator=g).tolist()
    coords[0], coords[1] = (120, 2100), (2150, 300)               # clipped at two borders
    ref = O.extract_views(img, coords, (512, 512), 0.1, 2.5)
    got = ops.extract_views(img.cuda(), coords, (512, 512), 0.1, 2.5)
    assert np.array_equal(got.cpu().numpy(), ref.numpy())


@pytest.mark.parametrize("shape", [(2, 5, 7, 9), (1, 3, 64, 64), (2, 4, 33, 100), (1, 2, 300, 300)])
def test_channel_stats_vs_torch(shape):
    """BatchNorm batch statistics (sum, sum of squares per channel, float64), contiguous and channel-sliced inputs."""
    from cwfa_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g) * 3 + 1
    for xv in (x, torch.cat([x, x], 1)[:, :shape[1]]):
        st = ops.channel_stats(xv.cuda()).cpu().view(shape[1], 2)
        ref = torch.stack([xv.double().sum((0, 2, 3)), (xv.double() ** 2).sum((0, 2, 3))], 1)
        assert float((st - ref).abs().max() / ref.abs().max()) < 1e-12


def test_bn_running_update_matches_torch_batchnorm():
    """Train-mode buffer bookkeeping (running mean / unbiased running var / num_batches_tracked) vs nn.BatchNorm2d on CPU."""
    from cwfa_amd import ops
    g = torch.Generator().manual_seed(77)
    x = torch.randn(3, 6, 9, 11, generator=g) * 2 + 0.5
    ref = torch.nn.BatchNorm2d(6, momentum=0.1)
    ref.running_mean.copy_(torch.randn(6, generator=g))
    ref.running_var.copy_(torch.rand(6, generator=g) + 0.5)
    rm, rv, nbt = ref.running_mean.clone().cuda(), ref.running_var.clone().cuda(), ref.num_batches_tracked.clone().cuda()
    ref.train()(x)
    st = ops.channel_stats(x.cuda())
    ops.bn_running_update(st, x.numel() // 6, 0.1, rm, rv, nbt)
    assert_close(rm, ref.running_mean, 1e-6)
    assert_close(rv, ref.running_var, 1e-6)
    assert int(nbt) == int(ref.num_batches_tracked) == 1


@pytest.mark.parametrize("cfg", [(16, 64, 3, "configs[0]: 64x64x16, 2-scale"), (48, 256, 4, "configs[1]: 256x256x48, 3-scale")])
def test_baseline_small_configs_inverse_vs_oracle(cfg):
    """BASELINE.json configs[0] and configs[1]: the k-scale inverse on a synthetic lowest-resolution volume (no LRNN: its
    mean branch is hard-wired to 512^2, networks.py:472,528), default-init weights, against the CPU oracle."""
    from cwfa_amd import CWFA
    from oracle import cwfa_oracle as O
    D, side, S, what = cfg
    torch.manual_seed(0)
    np.random.seed(0)
    conv_inn, cond_nets = CWFA.build_networks(D, side, S, with_lrnn=False, device="cuda")
    g = torch.Generator().manual_seed(1)
    cond_input = torch.randn(1, 29, side, side, generator=g)
    mean_cache = [0.1 * torch.randn(1, D // 2 ** (n + 1), side, side, generator=g) for n in range(S - 1)]
    low = torch.randn(1, D // 2 ** (S - 1), side, side, generator=g)
    from cwfa_amd import ops
    runs = {}
    for prec in ("fp32", "split_bf16"):                # the plain fp32 MFMA kernels, and the benchmark's arithmetic
        ops.set_precision(prec)
        try:
            with torch.no_grad():
                runs[prec] = CWFA.inverse_pass(conv_inn, cond_nets, cond_input.cuda(), [m.cuda() for m in mean_cache], low=low.cuda(),
                                               keep_all=True)
        finally:
            ops.set_precision("fp32")
    cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}   # noqa: E731
    steps = []
    for n, gi in enumerate(conv_inn):
        axes = {i: (m.axis if hasattr(m, "axis") else 1) for i, m in enumerate(gi.module_list) if hasattr(m, "perm")}
        steps.append({"inn": cpu(gi.state_dict()), "omega": cpu(cond_nets[n].state_dict()), "axes": axes})
    with torch.no_grad():
        ref = O.inverse_pass(steps, low, cond_input, mean_cache)
    for prec, vols in runs.items():
        assert vols[-1].shape == (1, D, side, side)
        for i, (a, b) in enumerate(zip(vols, ref)):
            assert_close(a, b, TOL, f"{what}, {prec}, level {i}")


def test_per_channel_prelu_slopes_in_the_split_3x3_epilogue():
    """cwfa_conv_opts.prelu_per_channel: one slope per output channel (1.0 = identity) against float64 torch -- the form in which
    conv1 (+ PReLU) and downsample (plain) of several condition nets run as one convolution."""
    from cwfa_amd import ops
    F = torch.nn.functional
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 29, 37, 45, generator=g)
    w, b = torch.randn(180, 29, 3, 3, generator=g) / 16, torch.randn(180, generator=g) * 0.1
    slopes = torch.where(torch.arange(180) % 24 < 12, torch.tensor(0.25), torch.tensor(1.0))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = torch.where(ref > 0, ref, ref * slopes.double().view(1, -1, 1, 1))
    ops.set_precision("split_bf16")
    try:
        pc = ops.pack_conv_weight(w.cuda())
        assert pc.split
        y = ops.conv2d(x.cuda(), pc, bias=b.cuda(), act="prelu", prelu_alpha=slopes.cuda())
        with pytest.raises(ValueError):
            ops.conv2d(x.cuda(), pc, bias=b.cuda(), act="prelu", prelu_alpha=slopes[:7].cuda())
    finally:
        ops.set_precision("fp32")
    assert_close(y, ref, 3e-6, "per-channel PReLU")


def test_condition_nets_with_merged_first_convolutions():
    """networks.omega_first_scope: conv1 / downsample of the four steps' condition nets as one launch (split precision) against the
    blocks' own launches -- same values up to the kernels' rounding (the separate launches take other tilings / the fp32 Winograd
    kernel for the small banks), and the scope leaves train-mode blocks and other inputs alone."""
    from cwfa_amd import networks as N, ops
    torch.manual_seed(3)
    nets = [N.cond_network(29, c, 1).cuda().eval() for c in (48, 24, 12, 6)]
    x = torch.randn(1, 29, 96, 128).cuda()
    ops.set_precision("split_bf16")
    try:
        with torch.no_grad():
            ref = [net(x)[-1] for net in nets]
            rec = []
            with N.omega_first_scope(nets, x) as sc:
                assert len(sc.maps) == 4
                got = [net(x)[-1] for net in nets]
                other = nets[0](x.clone())[-1]              # another tensor object: the block runs its own launches
            with N.omega_first_scope(nets[:1], x) as sc1:   # a single net: nothing to merge
                assert not sc1.maps
            nets[1].train()
            with N.omega_first_scope(nets, x) as sc2:       # a train-mode block (Dropout3d live): the scope stays out
                assert not sc2.maps
            nets[1].eval()
    finally:
        ops.set_precision("fp32")
    assert N._omega_first is None
    for a, b in zip(got, ref):
        assert_close(a, b, 1e-5, "merged first convolutions of the condition nets")
    assert torch.equal(other, ref[0])


@pytest.mark.parametrize("block_type", ["GLOW", "AI1", "RNVP", "GIN"])
def test_full_size_block_types_vs_oracle(block_type):
    """The block types ``--INN_block_type`` selects besides CAT (networks.py:289-297; north_star names GLOW and AI1), at
    FULL size: the finest flow step of the 512x512x96 configuration (48 flow channels, its condition net, 4 blocks,
    permutations) forward (latent, log-det) and inverse against the CPU oracle, in fp32 and in the benchmark's split-bf16
    precision.  These blocks' coefficients depend on the data: the step runs on the mixed plan (Haar + Split + first CAT +
    permutation in one chain launch; then block by block, halves written in place, in split precision with the coupling in
    the epilogue of each sub-network's last convolution)."""
    from cwfa_amd import CWFA, ops
    from oracle import cwfa_oracle as O
    torch.manual_seed(0)
    np.random.seed(0)
    conv_inn, cond_nets = CWFA.build_networks(96, 512, 2, block_type=block_type, with_lrnn=False, device="cuda")
    gi, cn = conv_inn[0], cond_nets[0]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 96, 512, 512, generator=g)
    views = torch.randn(1, 29, 512, 512, generator=g)
    mean = 0.1 * torch.randn(1, 48, 512, 512, generator=g)
    low = torch.randn(1, 48, 512, 512, generator=g)
    cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}   # noqa: E731
    axes = {i: (m.axis if hasattr(m, "axis") else 1) for i, m in enumerate(gi.module_list) if hasattr(m, "perm")}
    with torch.no_grad():
        om_r = O.omega_net(cpu(cn.state_dict()), views)
        (z_r, low_r), ld_r = O.flow_step(cpu(gi.state_dict()), x, [om_r, mean], False, axes, block_type)
        x_r, _ = O.flow_step(cpu(gi.state_dict()), (torch.zeros_like(low), low), [om_r, mean], True, axes, block_type)
    for prec in ("fp32", "split_bf16"):
        ops.set_precision(prec)
        try:
            with torch.no_grad():
                om = cn(views.cuda())[-1]
                Z, logdet = gi(x.cuda(), c=[om, mean.cuda()])
                xi, _ = gi([torch.zeros_like(low).cuda(), low.cuda()], c=[om, mean.cuda()], rev=True)
        finally:
            ops.set_precision("fp32")
        assert_close(Z[0], z_r, TOL, f"{block_type} {prec}: latent")
        assert_close(Z[1], low_r, 5e-6, f"{block_type} {prec}: low band")
        assert_close(logdet, ld_r, TOL, f"{block_type} {prec}: log-det")
        assert_close(xi, x_r, TOL, f"{block_type} {prec}: inverse")


def test_full_config3_inverse_vs_oracle():
    """BASELINE.json configs[2] at FULL size -- 512x512x96, LRNN (train-mode BatchNorm as CWFA.py:532) + 4 flow steps with
    their condition nets, default-init weights -- against the CPU oracle on identical inputs.  The stochastic layers
    (dropout2d, drop_path) are switched off on both sides; everything else is the benchmark's path.  ~20-60 s of CPU."""
    from cwfa_amd import CWFA
    from oracle import cwfa_oracle as O
    torch.manual_seed(0)
    np.random.seed(0)
    conv_inn, cond_nets = CWFA.build_networks(96, 512, 5, with_lrnn=True, device="cuda")
    enc = cond_nets[-1]
    enc.net.deconv[1].drop_out = 0
    for cn in enc.net.conv3d:
        cn.drop_prob = 0.0
    g = torch.Generator().manual_seed(1)
    cond_input = torch.randn(1, 29, 512, 512, generator=g)
    mean_cache = [0.1 * torch.randn(1, 96 // 2 ** (n + 1), 512, 512, generator=g) for n in range(4)]
    with torch.no_grad():
        out = CWFA.inverse_pass(conv_inn, cond_nets, cond_input.cuda(), [m.cuda() for m in mean_cache])
    torch.cuda.synchronize()
    cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}   # noqa: E731
    steps = []
    for n, gi in enumerate(conv_inn):
        axes = {i: (m.axis if hasattr(m, "axis") else 1) for i, m in enumerate(gi.module_list) if hasattr(m, "perm")}
        steps.append({"inn": cpu(gi.state_dict()), "omega": cpu(cond_nets[n].state_dict()), "axes": axes})
    with torch.no_grad():
        ref = O.inverse_pass(steps, None, cond_input, mean_cache, lrnn_sd=cpu(enc.state_dict()), lrnn_train=True)[-1]
    assert out.shape == (1, 96, 512, 512)
    assert_close(out, ref, TOL, "full-size config-3 inverse")
    from cwfa_amd import ops
    from conftest import rel_err
    # bench.py's headline precision: every heavy convolution (incl. the 64-channel fused layers, both convs) from an exact
    # three-way bf16 split of both operands, six products, fp32 accumulation -- held to the SAME fp32 bound (1e-4)
    ops.set_precision("split_bf16")
    try:
        with torch.no_grad():
            outs = CWFA.inverse_pass(conv_inn, cond_nets, cond_input.cuda(), [m.cuda() for m in mean_cache])
        torch.cuda.synchronize()
    finally:
        ops.set_precision("fp32")
    assert_close(outs, ref, TOL, "full-size config-3 inverse, split-bf16 (fp32-equivalent) precision")
    assert float((outs - out).abs().max()) > 0.0, "the split mode must actually run the split kernels"
    # BASELINE.json configs[4]: the same inverse with bf16 operands in the heavy convolutions (fp32 accumulation; wavelets,
    # couplings, permutations in fp32).  Tolerance re-stated for bf16 as SURVEY.md 8(d) measured on the reference's own
    # CPU autocast: max|d|/max|ref| <= 1e-2 and L2-relative <= 5e-3 against the fp32 oracle.
    ops.set_precision("bf16")
    try:
        with torch.no_grad():
            out16 = CWFA.inverse_pass(conv_inn, cond_nets, cond_input.cuda(), [m.cuda() for m in mean_cache])
        torch.cuda.synchronize()
    finally:
        ops.set_precision("fp32")
    m16, l16 = rel_err(out16, ref)
    assert m16 <= 1e-2 and l16 <= 5e-3, f"bf16 config: max-rel {m16:.3e}, l2-rel {l16:.3e}"
    assert max(rel_err(out16, out)) > 1e-6, "the bf16 mode must actually run bf16 kernels"


def test_full_size_forward_nll_vs_oracle():
    """BASELINE.json configs[3] per rank: the forward / NLL step of the finest flow (512x512x96 volumes, batch 2) with its
    condition net, GPU (fused forward chain, float64 shard sums) vs the CPU oracle: latent, low band, log-det, NLL."""
    from cwfa_amd import CWFA
    from oracle import cwfa_oracle as O
    torch.manual_seed(0)
    np.random.seed(0)
    conv_inn, cond_nets = CWFA.build_networks(96, 512, 2, with_lrnn=False, device="cuda")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 96, 512, 512, generator=g)
    views = torch.randn(2, 29, 512, 512, generator=g)
    mean = 0.1 * torch.randn(2, 48, 512, 512, generator=g)
    with torch.no_grad():
        om = cond_nets[0](views.cuda())[-1]
        nll, Z, logdet = CWFA.nll_step(conv_inn[0], x.cuda(), [om, mean.cuda()])
    cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}   # noqa: E731
    gi = conv_inn[0]
    axes = {i: (m.axis if hasattr(m, "axis") else 1) for i, m in enumerate(gi.module_list) if hasattr(m, "perm")}
    with torch.no_grad():
        omr = O.omega_net(cpu(cond_nets[0].state_dict()), views)
        (zr, lowr), ldr = O.flow_step(cpu(gi.state_dict()), x, [omr, mean], False, axes)
    assert_close(Z[0], zr, TOL, "latent")
    assert_close(Z[1], lowr, 2e-6, "low band")
    assert_close(logdet, ldr, TOL, "log-det")
    ss, sl, B = O.nll_terms(zr, ldr)
    ref = O.nll_from_terms(ss, sl, B, x.numel())
    assert abs(float(nll) - ref) <= 1e-5 * abs(ref)


@pytest.mark.parametrize("prec", ["fp32", "split_bf16"])
def test_full_size_step_with_saturated_couplings_vs_oracle(prec):
    """Default-init weights keep |s| small (the `_first` last convolution has gain 0.01).  Trained weights may drive the
    couplings to their clamp: here the output convolutions of all five sub-networks of the finest step (512x512x96) are
    scaled up until s spans the whole +-2 * 0.636 * pi/2 range of the ATAN clamp (every block then amplifies by up to e^2),
    and the inverse and the forward pass are compared with the CPU oracle under the same bound."""
    from cwfa_amd import CWFA, ops
    from oracle import cwfa_oracle as O
    torch.manual_seed(0)
    np.random.seed(0)
    conv_inn, cond_nets = CWFA.build_networks(96, 512, 2, with_lrnn=False, device="cuda")
    gi, cn = conv_inn[0], cond_nets[0]
    with torch.no_grad():
        for m in gi.module_list:
            net = getattr(m, "subnet", None)
            if net is not None:
                last = net.block72[1] if net.normal else net.block7[1]
                last.weight.mul_(1000.0 if not net.normal else 20.0)
                last.bias.mul_(10.0)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 96, 512, 512, generator=g)
    views = torch.randn(1, 29, 512, 512, generator=g)
    mean = 0.1 * torch.randn(1, 48, 512, 512, generator=g)
    low = torch.randn(1, 48, 512, 512, generator=g)
    cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}   # noqa: E731
    axes = {i: (m.axis if hasattr(m, "axis") else 1) for i, m in enumerate(gi.module_list) if hasattr(m, "perm")}
    with torch.no_grad():
        om_r = O.omega_net(cpu(cn.state_dict()), views)
        (z_r, low_r), ld_r = O.flow_step(cpu(gi.state_dict()), x, [om_r, mean], False, axes)
        x_r, ldi_r = O.flow_step(cpu(gi.state_dict()), (torch.zeros_like(low), low), [om_r, mean], True, axes)
    # the couplings really are driven hard: five blocks of e^{+-s}, |s| up to 2, spread a unit-variance input over orders of magnitude
    assert float(z_r.abs().max()) > 100.0 and float(z_r.std()) > 3.0, (float(z_r.abs().max()), float(z_r.std()))
    ops.set_precision(prec)
    try:
        with torch.no_grad():
            om = cn(views.cuda())[-1]
            Z, logdet = gi(x.cuda(), c=[om, mean.cuda()])
            xi, ldi = gi([torch.zeros_like(low).cuda(), low.cuda()], c=[om, mean.cuda()], rev=True)
    finally:
        ops.set_precision("fp32")
    assert_close(Z[0], z_r, TOL, f"{prec}: latent")
    assert_close(logdet, ld_r, TOL, f"{prec}: log-det")
    assert_close(xi, x_r, TOL, f"{prec}: inverse")
    assert_close(ldi, ldi_r, TOL, f"{prec}: log-det of the inverse")


def test_config4_forward_nll_pass_batch4_vs_oracle():
    """BASELINE.json configs[3] at its per-GPU shape: batch 4 of 512x512x96 volumes, the forward / NLL pass over ALL four
    flow steps with their condition nets (``CWFA.forward_nll_pass``: what bench.py's forward_nll leg times) against the CPU
    oracle step by step: latent, low band (the next level's input), log-det and the NLL of CWFA.py:978.  ~1-2 min of CPU."""
    from cwfa_amd import CWFA
    from oracle import cwfa_oracle as O
    torch.manual_seed(0)
    np.random.seed(0)
    B = 4
    conv_inn, cond_nets = CWFA.build_networks(96, 512, 5, with_lrnn=False, device="cuda")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 96, 512, 512, generator=g)
    views = torch.randn(B, 29, 512, 512, generator=g)
    means = [0.1 * torch.randn(B, 96 // 2 ** (n + 1), 512, 512, generator=g) for n in range(4)]
    xg, vg, mg = x.cuda(), views.cuda(), [m.cuda() for m in means]
    with torch.no_grad():
        nll, low = CWFA.forward_nll_pass(conv_inn, cond_nets, xg, vg, mg)
        per_step = []
        gt = xg
        for n, gi in enumerate(conv_inn):
            Z, logdet, _ = CWFA.nll_terms(gi, gt, [cond_nets[n](vg)[-1], mg[n]])
            per_step.append((Z[0].cpu(), Z[1].cpu(), logdet.cpu()))
            gt = Z[1]
    cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}   # noqa: E731
    gt = x
    for n, gi in enumerate(conv_inn):
        axes = {i: (m.axis if hasattr(m, "axis") else 1) for i, m in enumerate(gi.module_list) if hasattr(m, "perm")}
        with torch.no_grad():
            om = O.omega_net(cpu(cond_nets[n].state_dict()), views)
            (zr, lowr), ldr = O.flow_step(cpu(gi.state_dict()), gt, [om, means[n]], False, axes)
        assert_close(per_step[n][0], zr, TOL, f"latent, step {n}")
        assert_close(per_step[n][1], lowr, 5e-6, f"low band, step {n}")
        assert_close(per_step[n][2], ldr, TOL, f"log-det, step {n}")
        ss, sl, Bn = O.nll_terms(zr, ldr)
        ref = O.nll_from_terms(ss, sl, Bn, gt.numel())
        assert abs(float(nll[n]) - ref) <= 1e-5 * abs(ref), (n, float(nll[n]), ref)
        gt = lowr
    assert_close(low, gt, 1e-5, "lowest-resolution volume")


@pytest.mark.parametrize("cfg", [(1, 70, 16, 32, 200), (2, 33, 9, 37, 130), (1, 256, 24, 64, 512)])
def test_split_bf16_1x1_is_fp32_accurate(cfg):
    """Opt-in split-bf16 GEMM (three bf16 pieces per operand, six products, fp32 accumulate): same tolerance as the fp32
    matrix-core path against an fp64 reference, with prologue, epilogues and ragged tiles."""
    from cwfa_amd import ops
    B, Cin, H, W, Cout = cfg
    F = torch.nn.functional
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    alpha = torch.tensor([0.2])
    sc, sh = torch.rand(B, Cin, generator=g) + 0.5, torch.randn(B, Cin, generator=g)
    add = torch.randn(B, Cin, H, W, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double())
    xin = x.double() * sc.double().view(B, -1, 1, 1) + sh.double().view(B, -1, 1, 1) + add.double()
    ref_pro = F.prelu(F.conv2d(xin, w.double(), b.double()), alpha.double())
    ops.set_option("split_bf16", 1)
    try:
        pc = ops.pack_conv_weight(w.cuda())
        assert pc.split
        y = ops.conv2d(x.cuda(), pc, bias=b.cuda())
        y2 = ops.conv2d(x.cuda(), pc, bias=b.cuda(), act="elu", residual=res.cuda(), act2="elu")
        y3 = ops.conv2d(x.cuda(), pc, bias=b.cuda(), act="prelu", prelu_alpha=alpha.cuda(), in_scale=sc.cuda(), in_shift=sh.cuda(),
                        in_add=add.cuda())
    finally:
        ops.set_option("split_bf16", 0)
    assert_close(y, ref, 2e-6, "plain")
    assert_close(y2, F.elu(F.elu(ref) + res.double()), 3e-6, "elu+res+elu")
    assert_close(y3, ref_pro, 3e-6, "prologue + prelu")


def test_split_bf16_conv_transpose():
    from cwfa_amd import ops
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 40, 9, 16, generator=g)
    w = torch.randn(40, 48, 2, 2, generator=g) * 0.2
    b = torch.randn(48, generator=g)
    ops.set_option("split_bf16", 1)
    try:
        pc = ops.pack_conv_weight(w.cuda(), transposed=True)
        assert pc.split
        y = ops.conv2d(x.cuda(), pc, bias=b.cuda())
    finally:
        ops.set_option("split_bf16", 0)
    assert_close(y, torch.nn.functional.conv_transpose2d(x.double(), w.double(), b.double(), stride=2), 2e-6)


@pytest.mark.parametrize("cfg", [(1, 70, 7, 63, 200), (1, 256, 8, 64, 256), (2, 20, 12, 37, 320), (1, 6, 19, 40, 256),
                                 (1, 48, 9, 33, 130), (2, 33, 17, 50, 520), (1, 64, 16, 32, 96), (1, 64, 5, 20, 12),
                                 # the narrow tilings: 32 / 16 output channels per block on the 16-row tile (<= 32 / <= 16 outputs)
                                 (1, 64, 33, 50, 24), (2, 29, 20, 40, 6), (1, 24, 16, 32, 32), (1, 64, 40, 64, 17), (1, 29, 9, 12, 16)])
def test_split_bf16_3x3_is_fp32_accurate(cfg):
    """The split-bf16 3x3 kernel (K = 32 tap pairs on v_mfma_f32_16x16x32_bf16; tilings of 256 / 128 / 64 output channels):
    zero padding and the channel tail through the buffer range check, odd and even chunk counts, the load-side prologue
    applied by the kernel itself, specialised and run-time epilogues, per-sample affine tables."""
    level = 2
    from cwfa_amd import ops
    B, Cin, H, W, Cout = cfg
    F = torch.nn.functional
    g = torch.Generator().manual_seed(Cin * 3 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    alpha = torch.tensor([0.2])
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)
    add = torch.randn(B, Cin, H, W, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xin = x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + add.double()
    lin_pro = F.conv2d(xin, w.double(), b.double(), padding=1)
    ops.set_option("split_bf16", level)
    keep_min, keep_nar = ops.SPLIT_3X3_MIN_COUT, ops.SPLIT_3X3_NARROW_MAX
    ops.SPLIT_3X3_MIN_COUT = 1                 # also the narrow tilings ...
    ops.SPLIT_3X3_NARROW_MAX = 32              # ... for every bank with <= 32 outputs
    try:
        pc = ops.pack_conv_weight(w.cuda()) if Cin >= 29 or Cout > 32 else None
        if pc is None:                         # (the selection rule wants >= 29 inputs for the narrow tilings: pack directly)
            L_ = ops._lib.lib()
            packed = torch.empty(L_.cwfa_conv3x3_split_packed_bytes(Cout, Cin), dtype=torch.uint8, device="cuda")
            wc = w.cuda().contiguous()
            ops.check(L_.cwfa_conv3x3_split_pack_f32(ops._p(wc), ops._p(packed), Cout, Cin, ops._stream()), "pack")
            pc = ops.PackedConv(packed, Cout, Cin, 3, False, wc._version, wc.data_ptr(), split=True)
        assert pc.split
        got = {"plain": ops.conv2d(x.cuda(), pc, bias=b.cuda()),
               "prelu": ops.conv2d(x.cuda(), pc, bias=b.cuda(), act="prelu", prelu_alpha=alpha.cuda()),
               "generic": ops.conv2d(x.cuda(), pc, bias=b.cuda(), act="gelu", residual=res.cuda(), act2="relu"),
               "pro_prelu": ops.conv2d(x.cuda(), pc, bias=b.cuda(), act="prelu", prelu_alpha=alpha.cuda(), in_scale=sc.cuda(),
                                       in_shift=sh.cuda(), in_add=add.cuda()),
               "aff_prelu": ops.conv2d(x.cuda(), pc, bias=b.cuda(), act="prelu", prelu_alpha=alpha.cuda(), in_scale=sc.cuda(),
                                       in_shift=sh.cuda())}
    finally:
        ops.SPLIT_3X3_MIN_COUT, ops.SPLIT_3X3_NARROW_MAX = keep_min, keep_nar
        ops.set_option("split_bf16", 0)
    lin_aff = F.conv2d(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1), w.double(), b.double(), padding=1)
    want = {"plain": ref, "prelu": F.prelu(ref, alpha.double()), "generic": F.relu(F.gelu(ref) + res.double()),
            "pro_prelu": F.prelu(lin_pro, alpha.double()), "aff_prelu": F.prelu(lin_aff, alpha.double())}
    for k in want:
        assert_close(got[k], want[k], 5e-6, f"split 3x3 {k}")


@pytest.mark.parametrize("shape", [(2, 5, 16, 24, 8, 12), (1, 3, 10, 14, 5, 7), (1, 4, 9, 13, 4, 6), (1, 64, 128, 128, 64, 64)])
def test_maxpool_with_affine_vs_torch(shape):
    """cwfa_maxpool_f32 (adaptive max-pool of the BatchNorm-normalised map + the normalised full map, unet.py:79): the 16-byte
    2x2 fast path (W % 4 == 0), the generic window kernel (W % 4 != 0, overlapping adaptive windows) -- against ATen, bit-exact."""
    from cwfa_amd import ops
    B, Cc, H, W, Ho, Wo = shape
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(B, Cc, H, W, generator=g)
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    norm = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    ref = torch.nn.functional.adaptive_max_pool2d(norm, (Ho, Wo))
    y, full = ops.maxpool(x.cuda(), Ho, Wo, sc.cuda(), sh.cuda(), want_full=True)
    assert torch.equal(full.cpu(), norm) and torch.equal(y.cpu(), ref)
    y2 = ops.maxpool(x.cuda(), Ho, Wo)
    y2 = y2[0] if isinstance(y2, tuple) else y2
    assert torch.equal(y2.cpu(), torch.nn.functional.adaptive_max_pool2d(x, (Ho, Wo)))


@pytest.mark.parametrize("cfg", [(2, 3, 5, 9, 31, 8), (1, 24, 48, 40, 64, 64), (1, 16, 6, 17, 20, 64), (2, 7, 29, 16, 36, 40)])
def test_conv1x1_two_source_input_equals_the_concatenation(cfg):
    """cwfa_conv_opts.in_cat: the 1x1 convolution over cat(x, x2) read from the two tensors (the input of a coupling
    sub-network, coupling_layers.py:74-87) == the same convolution over the materialised concatenation, bit for bit (the zero
    columns the bank is padded with add +0.0 in place), both staging forms (W % 4 == 0 or not), channel-sliced views."""
    from cwfa_amd import ops
    B, c1, c2, H, W, cout = cfg
    g = torch.Generator().manual_seed(c1 + c2)
    w = (torch.randn(cout, c1 + c2, 1, 1, generator=g) / (c1 + c2) ** 0.5).cuda()
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    big = torch.randn(B, c1 + 3, H, W, generator=g).cuda()
    x, x2 = big[:, 3:], torch.randn(B, c2, H, W, generator=g).cuda()
    ref = ops.conv2d(ops.concat_channels([x, x2]), ops.pack_conv_weight(w), bias=bias)
    pc = ops.pack_conv_weight_cat(w, c1)
    assert torch.equal(ops.conv2d(x, pc, bias=bias, cat=x2), ref)
    assert_close(ref, torch.nn.functional.conv2d(torch.cat([x, x2], 1).double().cpu(), w.double().cpu(), bias.double().cpu()), 2e-6)
    with pytest.raises(ValueError):
        ops.conv2d(x2, pc, bias=bias, cat=x)


@pytest.mark.parametrize("cfg", [(1, 64, 256, 64, 64, "prelu"), (2, 24, 100, 19, 45, "prelu"), (2, 40, 40, 33, 31, None), (1, 16, 300, 8, 32, "prelu"),
                                 (1, 64, 64, 40, 64, None)])
def test_conv_epilogue_statistics_equal_a_pass_over_the_output(cfg):
    """cwfa_conv_opts.out_stats (unet.py:99-107: conv -> PReLU -> train-mode BatchNorm): the (sum, sum of squares) the split-bf16
    3x3 kernel takes from its accumulators against ops.channel_stats of the tensor it wrote -- every tiling (64 / 128 / 256
    output channels, 16-row tiles), ragged image borders, channels past Cout, batch > 1, accumulation into a used buffer."""
    from cwfa_amd import ops
    B, Cin, Cout, H, W, act = cfg
    g = torch.Generator().manual_seed(Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    a = torch.tensor([0.2]).cuda()
    ops.set_precision("split_bf16")
    try:
        pc = ops.pack_conv_weight(w)
        assert ops.conv_writes_stats(pc, act)
        st = torch.zeros(2 * Cout, dtype=torch.float64, device="cuda")
        y = ops.conv2d(x, pc, bias=b, act=act, prelu_alpha=a if act else None, out_stats=st)
        y0 = ops.conv2d(x, pc, bias=b, act=act, prelu_alpha=a if act else None)
        ops.conv2d(x, pc, bias=b, act=act, prelu_alpha=a if act else None, out_stats=st)      # adds: twice the sums
    finally:
        ops.set_precision("fp32")
    assert torch.equal(y, y0)
    ref = ops.channel_stats(y).cpu()
    ref64 = torch.stack([y.double().sum((0, 2, 3)), (y.double() ** 2).sum((0, 2, 3))], 1).reshape(-1).cpu()
    scale = ref64.view(-1, 2)[:, 1].max().sqrt() * (B * H * W) ** 0.5     # |sum| <= sqrt(n * sumsq)
    assert float((ref - ref64).abs().max()) <= 1e-6 * float(scale)
    assert float((st.cpu() / 2 - ref64).view(-1, 2)[:, 0].abs().max()) <= 2e-6 * float(scale)
    assert float(((st.cpu() / 2 - ref64).view(-1, 2)[:, 1] / ref64.view(-1, 2)[:, 1]).abs().max()) <= 2e-6
    with pytest.raises(ValueError):
        ops.conv2d(x, ops.pack_conv_weight(w), bias=b, out_stats=st)          # fp32 precision: no such epilogue


@pytest.mark.parametrize("cfg", [(1, 64, 64, 40, 70), (2, 40, 33, 19, 33), (1, 64, 64, 128, 128), (1, 32, 64, 8, 32)])
def test_split_bf16_7x7_is_fp32_accurate(cfg):
    """cwfa_conv7x7_split_f32 (the split 3x3 kernel with a 3-pixel halo and 49 taps: the ConvNeXt convolution, networks.py:488)
    against fp64 torch at the fp32 kernels' bound, ragged tiles, odd channel counts, batch > 1; and against the fp32 MFMA kernel."""
    from cwfa_amd import ops
    B, cin, cout, H, W = cfg
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 7, 7, generator=g) / (7 * cin ** 0.5)
    bias = torch.randn(cout, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), padding=3)
    y32 = ops.conv2d(x.cuda(), ops.pack_conv_weight(w.cuda()), bias=bias.cuda())
    ops.set_precision("split_bf16")
    try:
        pc = ops.pack_conv_weight(w.cuda())
        assert pc.split and pc.ks == 7
        y = ops.conv2d(x.cuda(), pc, bias=bias.cuda())
        with pytest.raises(ValueError):
            ops.conv2d(x.cuda(), pc, bias=bias.cuda(), act="elu")
    finally:
        ops.set_precision("fp32")
    assert_close(y, ref, 3e-6, "split 7x7 vs fp64")
    assert_close(y, y32, 5e-6, "split 7x7 vs fp32 MFMA kernel")


def test_bn_finish_equals_fold_plus_running_update_plus_mask():
    """cwfa_bn_finish_f32 = cwfa_bn_fold_f32 + cwfa_bn_running_update_f32 + the dropout factor (u >= p) / (1 - p) formed in the
    kernel + re-zeroing of the statistics buffer: bit-identical to the separate launches."""
    from cwfa_amd import ops
    g = torch.Generator().manual_seed(9)
    B, Cc, H, W = 3, 40, 12, 20
    y = torch.randn(B, Cc, H, W, generator=g).cuda()
    w, b = (torch.rand(Cc, generator=g) + 0.5).cuda(), torch.randn(Cc, generator=g).cuda()
    u = torch.rand(B, Cc, generator=g).cuda()
    p = 0.3
    rm0, rv0 = torch.randn(Cc, generator=g).cuda(), (torch.rand(Cc, generator=g) + 0.5).cuda()
    nbt0 = torch.tensor(7, dtype=torch.int64, device="cuda")
    n = float(B * H * W)
    st = ops.channel_stats(y)
    mask = ((u >= p).to(torch.float32) / (1.0 - p)).contiguous()
    rm1, rv1, nbt1 = rm0.clone(), rv0.clone(), nbt0.clone()
    ops.bn_running_update(st, n, 0.1, rm1, rv1, nbt1)
    sc_ref, sh_ref = ops.bn_fold(Cc, w, b, 1e-5, stats=st, count=n, mask_bc=mask)
    buf = torch.zeros(2 * Cc, dtype=torch.float64, device="cuda")
    ops.channel_stats(y, out=buf)                               # adds into the given (zeroed) buffer; float64 atomics: order-dependent
    assert_close(buf, st, 1e-13, "statistics into a given buffer")     # in the last bits only
    buf.copy_(st)                                               # the comparison below is on identical statistics
    rm2, rv2, nbt2 = rm0.clone(), rv0.clone(), nbt0.clone()
    sc, sh = ops.bn_finish(Cc, w, b, 1e-5, stats=buf, count=n, running_mean=rm2, running_var=rv2, momentum=0.1, num_batches_tracked=nbt2,
                           mask_u=u, drop_p=p, zero_stats=True)
    assert torch.equal(sc, sc_ref) and torch.equal(sh, sh_ref)
    assert torch.equal(rm2, rm1) and torch.equal(rv2, rv1) and int(nbt2) == int(nbt1) == 8
    assert float(buf.abs().max()) == 0.0
    sc3, sh3 = ops.bn_finish(Cc, w, b, 1e-5, running_mean=rm1, running_var=rv1)              # eval mode: running statistics, no mask
    sc4, sh4 = ops.bn_fold(Cc, w, b, 1e-5, running_mean=rm1, running_var=rv1)
    assert torch.equal(sc3, sc4) and torch.equal(sh3, sh4)
    sc5, sh5 = ops.bn_finish(Cc, mask_u=u, drop_p=p)                                           # a bare dropout factor
    assert torch.equal(sc5, mask.reshape(-1)) and float(sh5.abs().max()) == 0.0


def _to_blocked(t):
    B, Cc, H, W = t.shape
    return t.view(B, Cc // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous().view(B, Cc, H, W)


def _from_blocked(t):
    B, Cc, H, W = t.shape
    return t.view(B, Cc // 8, H, W, 8).permute(0, 1, 4, 2, 3).contiguous().view(B, Cc, H, W)


@pytest.mark.parametrize("shape", [(1, 16, 32), (2, 50, 70), (1, 33, 64), (1, 128, 96), (1, 20, 41)])
def test_split_layer_channel_blocked_layouts(shape):
    """cwfa_subnet_layer_split_f32 with channel-blocked input / output ([8][H][W][8]): the same arithmetic, only the memory
    order of the maps differs -> bit-identical to the NCHW form in all four combinations; and the split 3x3 convolution
    (plain and with the coupling epilogue) reading a blocked map."""
    from cwfa_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(H)
    x = torch.randn(B, 64, H, W, generator=g).cuda()
    w3, b3 = (torch.randn(64, 64, 3, 3, generator=g) / 24).cuda(), (torch.randn(64, generator=g) * 0.1).cuda()
    w1, b1 = (torch.randn(64, 64, 1, 1, generator=g) / 8).cuda(), (torch.randn(64, generator=g) * 0.1).cuda()
    ops.set_precision("split_bf16")
    try:
        pc = ops.pack_split_layer_weight(w3, w1)
        y0 = ops.subnet_layer(x, pc, b3, None, b1)
        xb = _to_blocked(x)
        assert torch.equal(_from_blocked(xb), x)
        assert torch.equal(_from_blocked(ops.subnet_layer(x, pc, b3, None, b1, layout=2)), y0)
        assert torch.equal(ops.subnet_layer(xb, pc, b3, None, b1, layout=1), y0)
        assert torch.equal(_from_blocked(ops.subnet_layer(xb, pc, b3, None, b1, layout=3)), y0)
        for cout in (96, 48):
            wl, bl = (torch.randn(cout, 64, 3, 3, generator=g) / 24).cuda(), (torch.randn(cout, generator=g) * 0.1).cuda()
            pl = ops.pack_conv_weight(wl)
            assert pl.split
            assert torch.equal(ops.conv2d(xb, pl, bias=bl, in_blocked=True), ops.conv2d(x, pl, bias=bl))
        n = 24
        wl, bl = (torch.randn(2 * n, 64, 3, 3, generator=g) / 24).cuda(), (torch.randn(2 * n, generator=g) * 0.1).cuda()
        bank = ops.pack_couple_weight(wl, bl)
        xa = torch.randn(B, n, H, W, generator=g).cuda()
        o0, o1 = torch.empty_like(xa), torch.empty_like(xa)
        ops.conv3x3_couple(x, bank, xa, o0, "ATAN", 2.0, 1.0, True)
        ops.conv3x3_couple(xb, bank, xa, o1, "ATAN", 2.0, 1.0, True, in_blocked=True)
        assert torch.equal(o0, o1)
        with pytest.raises(ValueError):
            ops.conv2d(xb, ops.pack_1x1_panel(w1), in_blocked=True)
        # the producer of the first map: the direct 1x1 kernel writing channel-blocked (both staging forms: W % 4 == 0 or not)
        for cin in (48, 6):
            wi, bi = (torch.randn(64, cin, 1, 1, generator=g) / cin ** 0.5).cuda(), (torch.randn(64, generator=g) * 0.1).cuda()
            pi = ops.pack_conv_weight(wi)
            ui = torch.randn(B, cin, H, W, generator=g).cuda()
            assert torch.equal(_from_blocked(ops.conv2d(ui, pi, bias=bi, out_blocked=True)), ops.conv2d(ui, pi, bias=bi))
    finally:
        ops.set_precision("fp32")


# ((6, *, 160, 160): 300 tiles on 256 persistent workgroups -- the next-tile staging of every form runs)
@pytest.mark.parametrize("cfg", [(1, 6, 40, 70), (2, 12, 33, 50), (1, 24, 64, 64), (1, 31, 17, 33), (1, 1, 16, 32), (6, 13, 160, 160), (6, 25, 160, 160)])
def test_first_layer_composed_form_is_fp32_accurate(cfg):
    """cwfa_subnet_layer_first_f32: the first layer of a sub-network with its 3x3 composed with the 1x1 in front
    (networks.py:621-631,641-665: conv3x3(conv1x1(u) + b0) = conv3x3'(u | 1), K = 9 x 32) against float64 torch evaluating the two
    convolutions one after the other -- borders (the ones channel is zero-padded like the map it stands for), ragged tiles, batch,
    both residual layouts; then a whole sub-network with and without the form."""
    from cwfa_amd import networks as N, ops
    B, cin, H, W = cfg
    F = torch.nn.functional
    g = torch.Generator().manual_seed(cin + H)
    u = torch.randn(B, cin, H, W, generator=g)
    w0, b0 = torch.randn(64, cin, 1, 1, generator=g) / cin ** 0.5, torch.randn(64, generator=g) * 0.3
    w3, b3 = torch.randn(64, 64, 3, 3, generator=g) / 24, torch.randn(64, generator=g) * 0.1
    w1, b1 = torch.randn(64, 64, 1, 1, generator=g) / 8, torch.randn(64, generator=g) * 0.1
    x64 = F.conv2d(u.double(), w0.double(), b0.double())
    ref = F.elu(F.conv2d(F.elu(F.conv2d(x64, w3.double(), b3.double(), padding=1)), w1.double(), b1.double()) + x64)
    ops.set_precision("split_bf16")
    try:
        pc = ops.pack_first_layer_weight(w0.cuda(), b0.cuda(), w3.cuda(), w1.cuda(), short=False)     # nine conv steps, either residual form
        pcx = ops.pack_first_layer_weight(w0.cuda(), b0.cuda(), w3.cuda(), w1.cuda())                 # the default: short where cin + 1 <= 16
        assert pcx.short == (cin + 1 <= 16)
        x = ops.conv2d(u.cuda(), ops.pack_conv_weight(w0.cuda()), bias=b0.cuda())
        u1 = ops.with_ones(u.cuda())
        assert u1.shape[1] == cin + 1 and float(u1[:, -1].min()) == 1.0
        y = ops.subnet_layer_first(u1, x, pc, b3.cuda(), b1.cuda())
        assert_close(y, ref, 5e-6, "composed first layer")
        yb = ops.subnet_layer_first(u1, _to_blocked(x), pc, b3.cuda(), b1.cuda(), layout=3)
        assert torch.equal(_from_blocked(yb), y)
        # x = None: the launch forms the first map itself (third k step of its 1x1 phase from u1 and the packed [W0 | b0])
        yx = ops.subnet_layer_first(u1, None, pc, b3.cuda(), b1.cuda())
        assert_close(yx, ref, 5e-6, "composed first layer, fused first map")
        yxb = ops.subnet_layer_first(u1, None, pc, b3.cuda(), b1.cuda(), layout=2)
        assert torch.equal(_from_blocked(yxb), yx)
        # ... and in its short form where u is one 16-channel chunk (five conv steps): the same sums in the same order
        ys = ops.subnet_layer_first(u1, None, pcx, b3.cuda(), b1.cuda())
        assert_close(ys, ref, 5e-6, "composed first layer, fused first map, default image")
        if pcx.short:
            assert torch.equal(ys, yx)
            assert torch.equal(_from_blocked(ops.subnet_layer_first(u1, None, pcx, b3.cuda(), b1.cuda(), layout=2)), ys)
            with pytest.raises(ValueError):
                ops.subnet_layer_first(u1, x, pcx, b3.cuda(), b1.cuda())                  # a short image forms its first map itself
        # the plain layer on the same maps: the same function up to the rounding of x between the two convolutions
        y2 = ops.subnet_layer(x, ops.pack_split_layer_weight(w3.cuda(), w1.cuda()), b3.cuda(), None, b1.cuda())
        assert_close(y2, y, 5e-6, "composed vs two-step")
        with pytest.raises(ValueError):
            ops.subnet_layer_first(u.cuda(), x, pc, b3.cuda(), b1.cuda())            # the ones channel is part of the contract
        if cin == 12:
            N.networks_n_chans = 64
            torch.manual_seed(cin)
            net = N.wavelet_flow_subnetwork2D(cin, 2 * cin).cuda()
            with torch.no_grad():
                a1 = net(u.cuda())
                ops.FIRST_LAYER_FUSED_X = False
                a2 = net(u.cuda())
                ops.FIRST_LAYER_COMPOSED = False
                a0 = net(u.cuda())
            assert not torch.equal(a0, a1) and not torch.equal(a2, a1)
            assert_close(a1, a0, 5e-6, "sub-network with / without the composed first layer")
            assert_close(a2, a0, 5e-6, "sub-network with the composed first layer reading its first map")
    finally:
        ops.FIRST_LAYER_COMPOSED = True
        ops.FIRST_LAYER_FUSED_X = True
        ops.set_precision("fp32")


def test_cat_step_with_merged_first_maps_equals_the_separate_launches():
    """ops.first_map_scope / networks.merged_first_maps: the first 1x1 convolutions of the five sub-networks of a CAT step (they all
    read the condition, coupling_layers.py:475-500) as ONE launch with stacked banks -- bit-identical to one launch per sub-network
    (same kernel, same arithmetic per output channel), both directions, batch 2, a ragged size."""
    from cwfa_amd import CWFA, networks as N, ops
    torch.manual_seed(21)
    np.random.seed(21)
    conv_inn, _ = CWFA.build_networks(16, 40, 2, with_lrnn=False, cond_chans=4, device="cuda")    # 64 internal channels
    gi = conv_inn[0]
    g = torch.Generator().manual_seed(22)
    x = torch.randn(2, 16, 40, 40, generator=g).cuda()
    c = [torch.randn(2, 8, 40, 40, generator=g).cuda(), 0.1 * torch.randn(2, 8, 40, 40, generator=g).cuda()]
    low = torch.randn(2, 8, 40, 40, generator=g).cuda()
    calls = []
    real = N.merged_first_maps
    ops.set_precision("split_bf16")
    ops.FIRST_LAYER_FUSED_X = False        # (with it the first layers form their first maps themselves and nothing is left to merge: below)
    try:
        outs = []
        for flag in (True, False):
            N.MERGE_FIRST_MAPS = flag
            N.merged_first_maps = lambda jobs: (calls.append(len(real(jobs))), real(jobs))[1]
            with torch.no_grad():
                (z, lo), j = gi(x, c=c)
                xi, _ = gi([None, low], c=c, rev=True)
            outs.append((z, lo, j, xi))
        assert calls == [5, 5, 0, 0], calls                    # all five sub-networks merged when on, none when off
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        # the default: every first layer takes its first map as a third k step of its own 1x1 phase -- no first 1x1 launch at all
        ops.FIRST_LAYER_FUSED_X = True
        N.MERGE_FIRST_MAPS = True
        del calls[:]
        with torch.no_grad():
            (z, lo), j = gi(x, c=c)
            xi, _ = gi([None, low], c=c, rev=True)
        assert calls == [0, 0], calls
        for a, b in zip((z, lo, j, xi), outs[0]):
            assert_close(a, b, 1e-5, "fused first maps vs first maps from memory")
    finally:
        N.merged_first_maps = real
        N.MERGE_FIRST_MAPS = True
        ops.FIRST_LAYER_FUSED_X = True
        ops.set_precision("fp32")


@pytest.mark.parametrize("block_type", ["GLOW", "AI1", "RNVP"])
def test_data_dependent_blocks_with_the_fused_first_layer(block_type):
    """Coupling blocks whose sub-network input cat(half, condition) has <= 31 channels (the coarse steps): the composed first layer
    with its fused first map takes cat(half, condition, 1) as one small tensor.  Against the same graph with the form switched off
    (two-source 1x1 + full first layer), both directions, split precision; and the form must really be taken."""
    from cwfa_amd import CWFA, ops
    torch.manual_seed(31)
    np.random.seed(31)
    conv_inn, _ = CWFA.build_networks(16, 48, 2, block_type=block_type, with_lrnn=False, cond_chans=4, device="cuda")
    gi = conv_inn[0]
    g = torch.Generator().manual_seed(32)
    x = torch.randn(2, 16, 48, 48, generator=g).cuda()
    c = [torch.randn(2, 8, 48, 48, generator=g).cuda(), 0.1 * torch.randn(2, 8, 48, 48, generator=g).cuda()]
    z0 = torch.randn(2, 8, 48, 48, generator=g).cuda()
    low = torch.randn(2, 8, 48, 48, generator=g).cuda()
    seen = []
    real = ops.subnet_layer_first
    ops.set_precision("split_bf16")
    try:
        outs = []
        for flag in (True, False):
            ops.FIRST_LAYER_FUSED_X = flag
            ops.subnet_layer_first = lambda u1, xx, *a, **k: (seen.append((flag, xx is None, u1.shape[1])), real(u1, xx, *a, **k))[1]
            with torch.no_grad():
                (z, lo), j = gi(x, c=c)
                xi, ji = gi([z0, low], c=c, rev=True)
            outs.append((z, lo, j, xi, ji))
        assert any(f and fused and ch > 9 for f, fused, ch in seen), seen      # a two-tensor input went through the fused form
        for a, b in zip(*outs):
            assert_close(a, b, 2e-5, f"{block_type}: fused first layer vs the two-source 1x1")
    finally:
        ops.subnet_layer_first = real
        ops.FIRST_LAYER_FUSED_X = True
        ops.set_precision("fp32")


def test_split_bf16_subnetwork_matches_the_fp32_path():
    """A whole coupling sub-network (networks.py:586-671) with the opt-in split level 2 -- its three fused layers on
    split_layer_kernel -- against the default fp32 path on the same input, at a size with ragged 32x32 tiles."""
    from cwfa_amd import networks as N, ops
    torch.manual_seed(4)
    net = N.wavelet_flow_subnetwork2D(58, 24).cuda()
    x = torch.randn(2, 58, 75, 41, device="cuda")
    with torch.no_grad():
        ref = net(x)
        ops.set_option("split_bf16", 2)
        try:
            got = net(x)
        finally:
            ops.set_option("split_bf16", 0)
    assert_close(got, ref, 5e-6, "sub-network, split level 2")


# ------------------------------------------------------------------------------------------------ sharded NLL on the GPU path
def _nll_rank(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)      # one card here: gloo (RCCL needs one GPU per rank)
    try:
        fx = load_golden("g10_pipeline")
        CWFA, conv_inn, cond_nets, S = _pipeline(fx)
        x = torch.from_numpy(fx["gt"]).cuda()
        g0 = torch.Generator().manual_seed(5)
        c = [torch.randn(x.shape[0], 8, *x.shape[2:], generator=g0).cuda(), 0.1 * torch.randn(x.shape[0], 8, *x.shape[2:], generator=g0).cuda()]
        B = x.shape[0]
        lo, hi = rank * B // world, (rank + 1) * B // world
        with torch.no_grad():
            nll, _, _ = CWFA.nll_step(conv_inn[0], x[lo:hi].contiguous(), [t[lo:hi].contiguous() for t in c])
            dist.barrier()
        q.put((rank, float(nll)))
    finally:
        dist.destroy_process_group()


def test_sharded_nll_two_ranks_on_the_gpu_path():
    """SURVEY.md 8(e) through the product: two processes share this card, each runs the fused forward chain on its half
    of the batch, one all-reduce of the float64[3] shard sums, both obtain the single-process NLL."""
    import socket
    import torch.multiprocessing as mp
    fx = load_golden("g10_pipeline")
    if fx["gt"].shape[0] < 2:
        pytest.skip("fixture batch too small to shard")
    CWFA, conv_inn, cond_nets, S = _pipeline(fx)
    x = torch.from_numpy(fx["gt"]).cuda()
    g0 = torch.Generator().manual_seed(5)
    c = [torch.randn(x.shape[0], 8, *x.shape[2:], generator=g0).cuda(), 0.1 * torch.randn(x.shape[0], 8, *x.shape[2:], generator=g0).cuda()]
    with torch.no_grad():
        ref, _, _ = CWFA.nll_step(conv_inn[0], x, c)
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nll_rank, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=300) for _ in procs]
    for p_ in procs:
        p_.join(timeout=120)
        assert p_.exitcode == 0
    for _, nll in res:
        assert abs(nll - float(ref)) <= 1e-6 * abs(float(ref)), (nll, float(ref))


@pytest.mark.parametrize("shape", [(1, 32, 32), (2, 17, 45), (1, 70, 96), (3, 16, 32), (1, 5, 3), (2, 130, 200), (6, 160, 160)])
def test_split_fused_layer_is_fp32_accurate(shape):
    """The fused sub-network layer with both convolutions on the split-bf16 core (persistent kernel, several tiles per
    workgroup at the larger shapes, ragged borders): same reference, same tolerance as the fp32 MFMA layer."""
    from cwfa_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(H * W + 1)
    x = torch.randn(B, 64, H, W, generator=g)
    w3, b3 = torch.randn(64, 64, 3, 3, generator=g) / 24, torch.randn(64, generator=g) * 0.1
    w1, b1 = torch.randn(64, 64, 1, 1, generator=g) / 8, torch.randn(64, generator=g) * 0.1
    F = torch.nn.functional
    xd = x.double()
    ref = F.elu(F.conv2d(F.elu(F.conv2d(xd, w3.double(), b3.double(), padding=1)), w1.double(), b1.double()) + xd)
    y = ops.subnet_layer(x.cuda(), ops.pack_split_layer_weight(w3.cuda(), w1.cuda()), b3.cuda(), None, b1.cuda())
    assert_close(y, ref, 3e-6, "split fused layer")


@pytest.mark.parametrize("shape", [(1, 32, 32), (2, 17, 45), (1, 70, 96)])
@pytest.mark.parametrize("products", [6, 1])
def test_split_fused_layer_tape_form(shape, products):
    """cwfa_subnet_layer_split_tape_f32 (training forward): the output is bit-identical to the plain launch and the hidden map
    h = ELU(conv3x3(x) + b3) it also writes matches the float64 reference (bf16 operands: the restated bound)."""
    from cwfa_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(H * W + 7)
    x = torch.randn(B, 64, H, W, generator=g)
    w3, b3 = torch.randn(64, 64, 3, 3, generator=g) / 24, torch.randn(64, generator=g) * 0.1
    w1, b1 = torch.randn(64, 64, 1, 1, generator=g) / 8, torch.randn(64, generator=g) * 0.1
    F = torch.nn.functional
    href = F.elu(F.conv2d(x.double(), w3.double(), b3.double(), padding=1))
    ops.set_option("split_products", products)
    try:
        pc = ops.pack_split_layer_weight(w3.cuda(), w1.cuda())
        y0 = ops.subnet_layer(x.cuda(), pc, b3.cuda(), None, b1.cuda())
        y, h = ops.subnet_layer(x.cuda(), pc, b3.cuda(), None, b1.cuda(), want_hidden=True)
    finally:
        ops.set_option("split_products", 6)
    assert torch.equal(y, y0)
    assert_close(h, href, 3e-6 if products == 6 else 1e-2, "hidden map of the tape form")


@pytest.mark.parametrize("extra", [[], ["--block-type", "GLOW"], ["--precision", "fp32"]])
def test_bench_line_contract_at_a_small_size(extra):
    """bench.py end to end on a small workload (flows + condition nets only: the LRNN's mean branch is hard-wired to 512^2): ONE
    JSON line with the contract's keys, a roofline fraction <= 1 on a kernel family that is named like the rocprofv3 kernel, and
    an in-path DWT figure."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-lrnn", "--side", "128", "--depths", "32", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-experiment", *extra], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "roofline_dwt"):
        assert k in d, k
    assert d["unit"] == "volumes/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] <= 1.0 and rf["unit"] == "TFLOP/s" and rf["achieved"] / rf["peak"] == pytest.approx(rf["frac"])
    assert d["roofline_dwt"]["bound"] == "hbm" and 0 < d["roofline_dwt"]["frac"] <= 1.0


# ------------------------------------------------------------------------------------------------ API surface not on the plan path
def test_sequence_inn_forward_and_inverse():
    """SequenceINN.forward (sequence_inn.py:65-101): a conditioned chain ActNorm -> PermuteRandom -> GLOW -> PermuteDim against
    the same modules applied one by one with the oracle's arithmetic -- outputs, summed log-det, the inverse and the tuple form."""
    from cwfa_amd import networks as N, INN_utils
    from cwfa_amd.FrEIA import framework as Ff, modules as Fm
    from oracle import cwfa_oracle as O
    N.networks_n_chans = 8
    torch.manual_seed(5)
    np.random.seed(5)
    C_, H, W, B = 6, 8, 16, 2
    seq = Ff.SequenceINN(C_, H, W)
    seq.append(Fm.ActNorm)
    seq.append(Fm.PermuteRandom, seed=3)
    seq.append(Fm.GLOWCouplingBlock, cond=0, cond_shape=(4, H, W), subnet_constructor=N.wavelet_flow_subnetwork2D, clamp=1.5,
               clamp_activation="ATAN")
    seq.append(INN_utils.PermuteDim, seed=4)
    seq = seq.cuda().eval()
    assert [tuple(s) for s in seq.shapes] == [(C_, H, W)] * 5
    gen = torch.Generator().manual_seed(6)
    x = (1.5 * torch.randn(B, C_, H, W, generator=gen) - 0.3)
    c = torch.randn(B, 4, H, W, generator=gen)
    with torch.no_grad():
        z, j = seq(x.cuda(), c=[c.cuda()])
        xr, jr = seq(z, c=[c.cuda()], rev=True)
    an, pr, glow, pd = list(seq.module_list)
    assert an.init_on_next_batch is False
    sc, bi = O.actnorm_init(x)
    v, j0 = O.actnorm(sc, bi, x, False)
    v = O.gather_axis(v, pr.perm.cpu(), 1)
    sd = {"b." + k: t.detach().cpu() for k, t in glow.state_dict().items()}
    v, j1 = O.block_two_sided(sd, "b.", v, [c], False, "GLOW", clamp=1.5, kind="ATAN")
    v = O.gather_axis(v, pd.perm.cpu(), pd.axis)
    assert_close(z, v, 1e-5, "SequenceINN forward")
    assert_close(j, j0 + j1, 1e-5, "SequenceINN log-det")
    assert_close(xr, x, 1e-5, "SequenceINN inverse")
    assert_close(jr, -(j0 + j1), 1e-5)
    seq.force_tuple_output = True
    with torch.no_grad():
        zt, _ = seq((x.cuda(),), c=[c.cuda()])
    assert isinstance(zt, (tuple, list)) and torch.equal(zt[0], z)


def test_reset_actnorm_rearms_the_data_dependent_init():
    """reset_ActNorm (networks.py:137-151): the first n ActNorm layers are re-armed; the next pass -- here the DEFAULT inverse
    pass with an all-zero latent given as None, the direction the reference's ActNorm initialises from just as well
    (invertible_resnet.py:68-72) -- re-initialises them from its batch, after which the fused plan runs again."""
    from cwfa_amd import networks as N, INN_utils
    from cwfa_amd.FrEIA import framework as Ff, modules as Fm
    from oracle import cwfa_oracle as O
    N.networks_n_chans = 8
    torch.manual_seed(7)
    np.random.seed(7)
    D, H, W, B = 8, 8, 64, 2
    C_ = D // 2
    nodes = [Ff.InputNode(D, H, W, name="input")]
    nodes.append(Ff.Node(nodes[-1], INN_utils.HaarTransform1D, {"order_by_wavelet": True}, name="haar"))
    split = Ff.Node(nodes[-1], Fm.Split, {"section_sizes": (C_, C_), "dim": 0}, name="split")
    nodes.append(split)
    cond = Ff.ConditionNode(C_, H, W, name="cond")
    nodes.append(cond)
    nodes.append(Ff.Node(split.out1, Fm.ConditionalAffineTransform, {"subnet_constructor": N.wavelet_flow_subnetwork2D},
                         conditions=[cond], name="cat0"))
    nodes.append(Ff.Node(nodes[-1], Fm.ActNorm, {}, name="an0"))
    nodes.append(Ff.Node(nodes[-1], Fm.ActNorm, {}, name="an1"))
    nodes.append(Ff.OutputNode(nodes[-1], name="z"))
    nodes.append(Ff.OutputNode(split.out0, name="low"))
    g = Ff.GraphINN(nodes).cuda()
    gen = torch.Generator().manual_seed(8)
    x = (2.0 * torch.randn(B, D, H, W, generator=gen) + 0.5).cuda()
    c = [torch.randn(B, C_, H, W, generator=gen).cuda()]
    low = torch.randn(B, C_, H, W, generator=gen).cuda()
    an = [m for m in g.module_list if isinstance(m, Fm.ActNorm)]
    # a fresh graph: the default inverse pass (z = None, an all-zero latent) must RUN although both ActNorms are still waiting for
    # their batch (round 2 raised ValueError here).  Like the reference's, an ActNorm that initialises itself on an all-zero batch
    # gets scale = log(1 / 0) = inf -- the reference's own behaviour (invertible_resnet.py:59-60), nothing to "fix"
    assert g._plan is not None and g._plan.needs_walk()
    with torch.no_grad():
        x0, _ = g([None, low], c=c, rev=True)
    assert not g._plan.needs_walk() and all(not m.init_on_next_batch for m in an)
    assert x0.shape == x.shape and torch.isinf(an[1].scale).all()
    # initialisation from the inverse direction on a real latent: the LAST ActNorm sees z first (invertible_resnet.py:68-72)
    N.reset_ActNorm(g)
    zin = (1.7 * torch.randn(B, C_, H, W, generator=gen) + 0.2).cuda()
    with torch.no_grad():
        x1, _ = g([zin, low], c=c, rev=True)
    sc, bi = O.actnorm_init(zin.cpu())
    assert_close(an[1].scale, sc, 1e-5, "ActNorm initialised from the inverse direction")
    assert_close(an[1].bias, bi, 1e-5)
    assert torch.isfinite(x1).all()
    with torch.no_grad():
        N.reset_ActNorm(g)
        g(x, c=c)
    s_old = [m.scale.detach().clone() for m in an]
    _, n = N.reset_ActNorm(g, n_to_reset=1)
    assert n == 1 and an[0].init_on_next_batch and not an[1].init_on_next_batch and g._plan.needs_walk()
    _, n = N.reset_ActNorm(g)
    assert n == 2 and all(m.init_on_next_batch for m in an)
    with torch.no_grad():
        (z, lo), j = g(3.0 * x, c=c)                                    # re-initialised on this batch: node walk
    assert not g._plan.needs_walk()
    assert not torch.equal(an[0].scale, s_old[0])
    # after the init the output of an ActNorm has zero mean and unit (unbiased) std per channel; the last one feeds z
    flat = z.transpose(0, 1).reshape(C_, -1)
    assert float(flat.mean(1).abs().max()) < 1e-4 and float((flat.std(1) - 1).abs().max()) < 1e-4
    with torch.no_grad():
        (z2, _), j2 = g(3.0 * x, c=c)                                   # the fused plan on the new parameters
    assert_close(z2, z, 1e-5, "plan after reset vs the initialising walk")
    assert_close(j2, j, 1e-5)


def test_log_jacobian_numerical_matches_the_analytic_logdet():
    """GraphINN.log_jacobian_numerical (graph_inn.py:369-407) on a tiny conditioned graph: central differences of the HIP
    forward against the log-det the blocks report, forward and inverse."""
    from cwfa_amd import networks as N
    from cwfa_amd.FrEIA import framework as Ff, modules as Fm
    N.networks_n_chans = 4
    torch.manual_seed(9)
    np.random.seed(9)
    C_, H, W, B = 2, 2, 2, 2
    inp = Ff.InputNode(C_, H, W, name="in")
    cond = Ff.ConditionNode(1, H, W, name="c")
    b1 = Ff.Node(inp, Fm.GLOWCouplingBlock, {"subnet_constructor": N.wavelet_flow_subnetwork2D, "clamp": 1.0}, conditions=[cond], name="g1")
    p1 = Ff.Node(b1, Fm.PermuteRandom, {"seed": 1}, name="p1")
    b2 = Ff.Node(p1, Fm.ConditionalAffineTransform, {"subnet_constructor": N.wavelet_flow_subnetwork2D}, conditions=[cond], name="c2")
    g = Ff.GraphINN([inp, cond, b1, p1, b2, Ff.OutputNode(b2, name="out")]).cuda().eval()
    for m in g.modules():                                 # default-init sub-networks give s ~ 1e-2: scale them up to a visible log-det
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data *= 2.0
    from cwfa_amd import ops
    ops.invalidate_packs()
    gen = torch.Generator().manual_seed(10)
    x = torch.randn(B, C_, H, W, generator=gen).cuda()
    c = [torch.randn(B, 1, H, W, generator=gen).cuda()]
    with torch.no_grad():
        z, j = g(x, c=c)
        jn = g.log_jacobian_numerical(x, c=c, h=4e-3)
        _, jr = g(z, c=c, rev=True)
        jrn = g.log_jacobian_numerical(z, c=c, rev=True, h=1e-3)
    assert float(j.abs().min()) > 1e-3, "the test needs a non-trivial log-det"
    # central differences in fp32: truncation ~h^2, round-off ~1e-7 / h per Jacobian entry
    assert torch.allclose(jn, j, atol=1e-2, rtol=5e-3), (jn, j)
    assert torch.allclose(jrn, jr, atol=5e-2, rtol=1e-2), (jrn, jr)      # (the inverse map is the more curved one: smaller h, more round-off)
    assert torch.allclose(jr, -j, atol=1e-5)


@pytest.mark.parametrize("temperature", [0.7, 2.0])
def test_truncated_normal_sampler_moments(temperature):
    """sample_z_truncated with T != 0 (CWFA.py:47-64 via utils.py:42-82: N(0,1) truncated to [-T, T]) -- a path that raises
    NameError in the reference itself, so there is nothing to match but the distribution: support, mean, variance and the
    central-interval mass of 4 M draws against the closed forms."""
    import math
    from cwfa_amd import CWFA
    torch.manual_seed(11)
    z = CWFA.sample_z_truncated((4, 4, 512, 512), device="cuda", temperature=temperature)
    assert z.shape == (4, 4, 512, 512) and z.is_cuda
    assert float(z.abs().max()) <= temperature
    T_ = temperature
    phi = math.exp(-T_ * T_ / 2) / math.sqrt(2 * math.pi)
    mass = math.erf(T_ / math.sqrt(2))
    var = 1 - 2 * T_ * phi / mass                         # variance of the symmetric truncation
    n = z.numel()
    assert abs(float(z.mean())) < 5 * math.sqrt(var / n)
    assert abs(float(z.var()) - var) < 0.01 * var
    inner = math.erf(0.5 * T_ / math.sqrt(2)) / mass      # P(|z| < T/2)
    assert abs(float((z.abs() < 0.5 * T_).float().mean()) - inner) < 5 * math.sqrt(inner * (1 - inner) / n)
    assert torch.equal(CWFA.sample_z_truncated((2, 3), device="cuda", temperature=0), torch.zeros(2, 3, device="cuda"))


def test_tanh_clamp_small_arguments_keep_their_logdet():
    """the TANH soft clamp of AllInOneBlock-style stages near s_raw = 0 (freshly initialised sub-networks):
    values and the summed log-det of ops.affine against float64, relative to the log-det's own size (the hardware-exp form
    1 - 2 / (e^{2x} + 1) alone cancels there: ~1e-3 relative at |x| = 1e-4)."""
    from cwfa_amd import ops
    gen = torch.Generator().manual_seed(12)
    B, C_, H, W = 2, 6, 16, 32
    x = torch.randn(B, C_, H, W, generator=gen)
    t = torch.randn(B, C_, H, W, generator=gen)
    for mag in (1e-4, 1e-3, 1e-2, 0.2, 0.5, 3.0):
        s_raw = mag * (torch.rand(B, C_, H, W, generator=gen) + 0.5)          # one sign: the sum does not cancel
        s64 = 2.0 * torch.tanh(0.1 * s_raw.double())
        ref = x.double() * s64.exp() + 0.1 * t.double()                 # (AllInOneBlock scales s AND t: all_in_one_block.py:215)
        ld = torch.zeros(B, dtype=torch.float64, device="cuda")
        y = ops.affine(x.cuda(), ops.stage(s_raw.cuda(), t.cuda(), "TANH", 2.0, 0.1), False, logdet=ld)
        assert_close(y, ref, 1e-6, f"TANH stage at |s_raw| ~ {mag}")
        want = s64.sum((1, 2, 3))
        assert float(((ld.cpu() - want) / want).abs().max()) < 2e-6, (mag, ld.cpu(), want)


def test_in_place_parameter_edits_need_invalidate_packs():
    """packed filter banks / bias rows are keyed on (_version, data_ptr), which `.data` edits do not change.
    The initialisers of cwfa_amd.networks (which edit through `.data`, like the reference's: networks.py:19-62) call
    ops.invalidate_packs(); after them a GLOW block on the coupling-epilogue path must see the new weights AND biases."""
    from cwfa_amd import networks as N, ops
    from cwfa_amd.FrEIA import modules as Fm
    N.networks_n_chans = 64
    torch.manual_seed(13)
    blk = Fm.GLOWCouplingBlock([(8, 16, 32)], dims_c=[(4, 16, 32)], subnet_constructor=N.wavelet_flow_subnetwork2D).cuda().eval()
    gen = torch.Generator().manual_seed(14)
    x = torch.randn(1, 8, 16, 32, generator=gen).cuda()
    c = [torch.randn(1, 4, 16, 32, generator=gen).cuda()]
    ops.set_precision("split_bf16")
    try:
        with torch.no_grad():
            (y0,), _ = blk((x,), c=c)
            torch.manual_seed(15)
            blk.apply(N.subnet_initialization)                      # `.data` edits + invalidate_packs
            for m in blk.modules():
                if isinstance(m, torch.nn.Conv2d) and m.bias is not None:
                    m.bias.data += 0.5
            ops.invalidate_packs()
            (y1,), _ = blk((x,), c=c)
        ops.set_precision("fp32")
        with torch.no_grad():
            (y2,), _ = blk((x,), c=c)                               # the fp32 path reads the parameters live
    finally:
        ops.set_precision("fp32")
    assert not torch.equal(y0, y1)
    assert_close(y1, y2, 1e-5, "re-packed split path vs live fp32 path")


@pytest.mark.parametrize("tool", [["stress_tilings.py", "3", "30"], ["stress_round3.py", "4", "4"]])
def test_random_shapes_through_the_new_kernels(tool):
    """tools/stress_tilings.py (random 3x3 banks over every tiling / epilogue / statistics form of the split kernel) and
    tools/stress_round3.py (composed / fused / short first layer, tape layer, split weight gradients, split Conv3d) against float64
    torch, a short run of each (their seeds differ from the runs recorded in DESIGN.md)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", tool[0]), *tool[1:]], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
