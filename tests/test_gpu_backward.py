"""GPU (MI355X): the backward kernels of the flow-step training loss against torch autograd in
float64 on the CPU -- the thing `full_loss.backward()` (CWFA.py:1002-1006) runs in the reference.
Tolerance: max|d|/max|ref| and L2-relative <= 1e-4 (fp32 sums over up to 2.6e5 pixels; observed ~1e-6)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cwfa_amd import _lib
    _lib.lib()
    yield
    torch.cuda.synchronize()


@pytest.mark.parametrize("cfg", [  # (B, Cin, Cout, H, W, ks)
    (1, 64, 64, 64, 64, 3), (2, 64, 64, 37, 45, 3), (1, 58, 64, 33, 31, 1), (2, 64, 24, 20, 70, 3), (1, 96, 96, 17, 33, 3),
    (1, 64, 64, 40, 40, 1), (3, 7, 5, 9, 11, 3), (1, 130, 70, 8, 8, 1), (1, 64, 64, 1, 1, 3), (1, 64, 64, 128, 128, 3),
    (1, 64, 64, 37, 45, 7), (2, 6, 70, 12, 66, 7),
    # W % 4 == 0: the LDS-DMA / row-paired form of the 3x3 kernel (ragged right edge, odd heights, partial channel tiles)
    (2, 64, 64, 37, 44, 3), (1, 70, 130, 19, 36, 3), (1, 8, 8, 6, 4, 3), (1, 64, 64, 5, 100, 3), (3, 29, 48, 16, 32, 3)])
def test_conv_weight_gradient_vs_autograd(cfg):
    from cwfa_amd import ops
    B, Cin, Cout, H, W, ks = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(B, Cin, H, W, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    w = torch.zeros(Cout, Cin, ks, ks, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), w, padding=ks // 2) * dy.double()).sum().backward()
    got, gb = ops.conv2d_wgrad(x.cuda(), dy.cuda(), ks, want_bias=True)
    assert_close(got, w.grad, TOL, f"dW {cfg}")
    assert_close(gb, dy.double().sum((0, 2, 3)), TOL, f"db {cfg}")
    # accumulate into an existing gradient buffer; channel-sliced (strided-batch) operands
    base = torch.randn(Cout, Cin, ks, ks, generator=g)
    xb = torch.randn(B, Cin + 3, H, W, generator=g)
    xb[:, 2:2 + Cin] = x
    bb = torch.randn(Cout, generator=g)
    acc, accb = ops.conv2d_wgrad(xb.cuda()[:, 2:2 + Cin], dy.cuda(), ks, out=base.cuda().clone(), accumulate=True, bias_out=bb.cuda().clone())
    assert_close(acc, w.grad + base.double(), TOL, f"dW accumulate {cfg}")
    assert_close(accb, dy.double().sum((0, 2, 3)) + bb.double(), TOL, f"db accumulate {cfg}")
    # deterministic: the partial sums are combined in a fixed order
    assert torch.equal(got, ops.conv2d_wgrad(x.cuda(), dy.cuda(), ks))


@pytest.mark.parametrize("ks", [3, 1])
@pytest.mark.parametrize("cfg", [  # (B, Cin, Cout, H, W): W % 4 == 0 (else the fp32 form runs)
    (1, 64, 64, 64, 64), (2, 64, 64, 37, 44), (1, 70, 130, 19, 36), (1, 8, 8, 6, 4), (1, 64, 64, 5, 100), (3, 29, 48, 16, 32),
    (1, 64, 64, 33, 32), (1, 128, 64, 70, 96), (2, 16, 96, 65, 8)])
def test_conv_weight_gradient_split_form_vs_autograd(cfg, ks):
    """The 3x3 / 1x1 weight gradient on the bf16 matrix cores in split arithmetic (option "wgrad_split", what set_precision("split_bf16")
    selects): same float64 reference and bound as the fp32 forms -- segment borders (SEG = 32 rows), ragged right edges, partial
    channel tiles, several samples, accumulation into an existing buffer, strided operands, determinism."""
    from cwfa_amd import ops
    B, Cin, Cout, H, W = cfg
    g = torch.Generator().manual_seed(sum(cfg) + 5)
    x = torch.randn(B, Cin, H, W, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    w = torch.zeros(Cout, Cin, ks, ks, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), w, padding=ks // 2) * dy.double()).sum().backward()
    ops.set_option("wgrad_split", 1)
    try:
        got, gb = ops.conv2d_wgrad(x.cuda(), dy.cuda(), ks, want_bias=True)
        base = torch.randn(Cout, Cin, ks, ks, generator=g)
        xb = torch.randn(B, Cin + 3, H, W, generator=g)
        xb[:, 2:2 + Cin] = x
        bb = torch.randn(Cout, generator=g)
        acc, accb = ops.conv2d_wgrad(xb.cuda()[:, 2:2 + Cin], dy.cuda(), ks, out=base.cuda().clone(), accumulate=True, bias_out=bb.cuda().clone())
        again = ops.conv2d_wgrad(x.cuda(), dy.cuda(), ks)
        ops.set_option("wgrad_split", 0)
        fp32 = ops.conv2d_wgrad(x.cuda(), dy.cuda(), ks)
    finally:
        ops.set_option("wgrad_split", 0)
    assert_close(got, w.grad, 3e-6, f"dW {cfg}")
    assert_close(gb, dy.double().sum((0, 2, 3)), TOL, f"db {cfg}")
    assert_close(acc, w.grad + base.double(), 3e-6, f"dW accumulate {cfg}")
    assert_close(accb, dy.double().sum((0, 2, 3)) + bb.double(), TOL, f"db accumulate {cfg}")
    assert torch.equal(got, again)
    assert_close(got, fp32, 3e-6, "split form vs fp32 form")


def test_elu_backward():
    from cwfa_amd import ops
    g0 = torch.Generator().manual_seed(3)
    q = torch.randn(2, 6, 10, 14, generator=g0, dtype=torch.float64, requires_grad=True)
    a = F.elu(q)
    gup = torch.randn(2, 6, 10, 14, generator=g0)
    add = torch.randn(2, 6, 10, 14, generator=g0)
    a.backward(gup.double())
    got = ops.elu_bwd(gup.cuda(), a.detach().float().cuda())
    assert_close(got, q.grad, 1e-6, "elu backward")
    got = ops.elu_bwd(gup.cuda(), a.detach().float().cuda(), add=add.cuda())
    assert_close(got, q.grad + add.double(), 1e-6, "elu backward + add")


def _torch_chain(hi, stages, final_perm):
    """float64 restatement of cwfa_chain_fwd_f32's stage loop; returns (z, sum of s per sample)."""
    v, logdet = hi, 0.0
    for st in stages:
        if st["perm"] is not None:
            v = v.index_select(st["axis"], st["perm"])
        s = st["clamp"] * 0.636 * torch.atan(st["s_raw"] * st["pre"]) if st["s_raw"] is not None else None
        if st["t"] is not None:
            t = -st["t"] / math.sqrt(2.0) if st["neg"] else st["t"] * st["pre"]
        else:
            t = 0.0
        if s is not None:
            v = torch.exp(s) * v + t
            logdet = logdet + s.flatten(1).sum(1)
        else:
            v = v + t
    return (v if final_perm is None else v.index_select(1, final_perm)), logdet


@pytest.mark.parametrize("shape", [(2, 6, 9, 12), (1, 12, 16, 16), (3, 3, 5, 7)])
def test_chain_backward_vs_autograd(shape):
    """Five stages with channel / row / column gathers, the `_first` pass-through stage (t = -mean/sqrt 2) and a trailing
    channel permutation -- the step graph of networks.py:305-366 -- against float64 autograd of the same chain."""
    from cwfa_amd import ops
    B, Cc, H, W = shape
    g0 = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, 2 * Cc, H, W, generator=g0)
    axes = [None, 1, 2, 3, 1]
    ref_stages, stages, leaves = [], [], []
    for k, ax in enumerate(axes):
        s_raw = torch.randn(B, Cc, H, W, generator=g0)
        t = torch.randn(B, Cc, H, W, generator=g0)
        perm = None if ax is None else torch.randperm([0, Cc, H, W][ax], generator=g0)
        neg = k == 0
        sr, tr = s_raw.double().requires_grad_(), t.double().requires_grad_()
        leaves.append((sr, tr))
        ref_stages.append({"s_raw": sr, "t": tr, "perm": perm, "axis": ax, "clamp": 2.0, "pre": 1.0, "neg": neg})
        stages.append(ops.stage(s_raw.cuda(), t.cuda(), "ATAN", 2.0, t_neg_div_sqrt2=neg, perm=None if perm is None else perm.cuda(),
                                axis=ax or 1))
    final_perm = torch.randperm(Cc, generator=g0)
    xd = x.double()
    hi = ((xd[:, 0::2] - xd[:, 1::2]) / math.sqrt(2.0)).requires_grad_()
    zr, ld = _torch_chain(hi, ref_stages, final_perm)
    numel = zr.numel()
    gscale, ldscale = 1.0 / numel, 1.0 / (B * numel)
    loss = gscale * 0.5 * (zr ** 2).sum() - ldscale * ld.sum()
    loss.backward()
    z, low = ops.chain_fwd(x.cuda(), stages, final_perm.cuda())
    assert_close(z, zr.detach(), 1e-5, "chain forward")
    grads = [(torch.empty(B, Cc, H, W, device="cuda"), torch.empty(B, Cc, H, W, device="cuda")) for _ in axes]
    gv0 = ops.chain_bwd(z, stages, grads, final_perm.cuda(), gscale, ldscale, want_input_grad=True)
    for k, ((ds, dt), (sr, tr)) in enumerate(zip(grads, leaves)):
        assert_close(ds, sr.grad, TOL, f"ds_raw stage {k}")
        assert_close(dt, tr.grad, TOL, f"dt stage {k}")
    assert_close(gv0, hi.grad, TOL, "gradient of the detail band")


def _golden_step(name):
    import numpy as np
    from conftest import load_golden, sd_of
    from cwfa_amd import networks as N
    from test_host_logic import build_step
    fx = load_golden(name)
    ix, n_ch = int(fx["ix"]), int(fx["n_ch"])
    keep = N.networks_n_chans
    try:
        _, g = build_step("CAT", ix, n_ch=n_ch)
    finally:
        N.networks_n_chans = keep
    g.load_state_dict(sd_of(fx))
    for i, m in enumerate(g.module_list):
        if f"meta/axis_{i}" in fx:
            assert int(m.axis) == int(fx[f"meta/axis_{i}"])
    return fx, g.train().cuda()


@pytest.mark.parametrize("name", ["g13_step_grad_k0_ch8", "g13_step_grad_k1_ch8", "g13_step_grad_k0_ch64"])
def test_step_nll_backward_golden(name):
    """Forward + backward of one CAT step's training NLL through the HIP path against the reference's own autograd
    (fixture g13, generated by importing the reference: oracle/make_golden.py step_grad): loss, the gradient of every
    parameter the loss reaches, and the gradients of both conditions."""
    from cwfa_amd import training
    fx, g = _golden_step(name)
    x = torch.from_numpy(fx["x"]).cuda()
    c = [torch.from_numpy(fx["c0"]).cuda(), torch.from_numpy(fx["c1"]).cuda()]
    nll, (z, low), cg = training.nll_backward(g, x, c, want_cond_grads=True)
    assert abs(float(nll) - float(fx["loss"])) <= 1e-5 * abs(float(fx["loss"]))
    assert_close(z, fx["z"], TOL, "z")
    want = {k[len("grad/"):]: v for k, v in fx.items() if k.startswith("grad/")}
    got = {k: p.grad for k, p in g.named_parameters() if p.grad is not None}
    assert set(got) == set(want), sorted(set(got) ^ set(want))[:6]
    for k in sorted(want):
        assert_close(got[k], want[k], TOL, k)
    assert_close(cg[0], fx["gc0"], TOL, "d loss / d omega")
    assert_close(cg[1], fx["gc1"], TOL, "d loss / d mean detail")
    # a second backward accumulates, as torch's .grad does
    training.nll_backward(g, x, c)
    k0 = sorted(want)[0]
    assert_close(dict(g.named_parameters())[k0].grad, 2 * torch.from_numpy(want[k0]), TOL, "accumulated gradient")


def test_step_nll_backward_golden_in_split_precision():
    """The same fixture (64-channel sub-networks) with the training forward on the split-bf16 kernels -- the sub-network layers in
    their tape form (cwfa_subnet_layer_split_tape_f32), data-gradient convolutions on the split 3x3 kernel: fp32-equivalent
    arithmetic, same bound."""
    from cwfa_amd import ops, training
    fx, g = _golden_step("g13_step_grad_k0_ch64")
    x = torch.from_numpy(fx["x"]).cuda()
    c = [torch.from_numpy(fx["c0"]).cuda(), torch.from_numpy(fx["c1"]).cuda()]
    ops.set_precision("split_bf16")
    try:
        nll, (z, low), cg = training.nll_backward(g, x, c, want_cond_grads=True)
    finally:
        ops.set_precision("fp32")
    assert abs(float(nll) - float(fx["loss"])) <= 1e-5 * abs(float(fx["loss"]))
    assert_close(z, fx["z"], TOL, "z")
    want = {k[len("grad/"):]: v for k, v in fx.items() if k.startswith("grad/")}
    got = {k: p.grad for k, p in g.named_parameters() if p.grad is not None}
    assert set(got) == set(want)
    for k in sorted(want):
        assert_close(got[k], want[k], TOL, k)
    assert_close(cg[0], fx["gc0"], TOL, "d loss / d omega")


def test_training_steps_reduce_the_nll():
    """Three plain gradient steps on one batch lower the NLL (end-to-end sign / scale check of the backward)."""
    from cwfa_amd import training
    fx, g = _golden_step("g13_step_grad_k0_ch8")
    x = torch.from_numpy(fx["x"]).cuda()
    c = [torch.from_numpy(fx["c0"]).cuda(), torch.from_numpy(fx["c1"]).cuda()]
    params = [p for p in g.parameters() if p.requires_grad]
    hist = []
    for _ in range(4):
        for p in params:
            p.grad = None
        nll, _, _ = training.nll_backward(g, x, c)
        hist.append(float(nll))
        training.sgd_step(params, 0.05)
    assert hist[-1] < hist[0] and all(b <= a + 1e-6 for a, b in zip(hist, hist[1:])), hist


def _train_rank(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)      # one card here: gloo (RCCL needs one GPU per rank)
    try:
        from cwfa_amd import training
        fx, g = _golden_step("g13_step_grad_k0_ch8")
        x = torch.from_numpy(fx["x"]).cuda()
        c = [torch.from_numpy(fx["c0"]).cuda(), torch.from_numpy(fx["c1"]).cuda()]
        B = x.shape[0]
        lo, hi = (0, 1) if rank == 0 else (1, B)                       # uneven shards: 1 + 2 samples
        nll, _, _ = training.nll_backward(g, x[lo:hi].contiguous(), [t[lo:hi].contiguous() for t in c])
        params = [p for p in g.parameters() if p.requires_grad]
        nb = training.allreduce_gradients(params, bucket_bytes=1 << 12)
        grads = {k: p.grad.cpu().numpy() for k, p in g.named_parameters() if p.grad is not None}     # plain arrays: no fd passing
        q.put((rank, float(nll), nb, grads))
    finally:
        dist.destroy_process_group()


def test_sharded_training_step_two_ranks_on_the_gpu_path():
    """Data-parallel training step through the product: two processes share this card, each runs forward + backward
    on its (uneven) batch shard with the GLOBAL normalisation, the gradients are summed through flat buckets; both ranks
    end with the loss and the gradients of the reference's single-process autograd (fixture g13)."""
    import socket
    import torch.multiprocessing as mp
    from conftest import load_golden
    fx = load_golden("g13_step_grad_k0_ch8")
    want = {k[len("grad/"):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("grad/")}
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_rank, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=300) for _ in procs]
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    for rank, nll, nb, grads in res:
        assert abs(nll - float(fx["loss"])) <= 1e-5 * abs(float(fx["loss"]))
        assert nb >= 2
        assert set(grads) == set(want)
        for k in want:
            assert_close(grads[k], want[k], TOL, f"rank {rank} {k}")


@pytest.mark.parametrize("name", ["g13_step_grad_k0_ch8", "g13_step_grad_k1_ch8"])
@pytest.mark.parametrize("kind", ["l2", "l1"])
def test_full_training_loss_backward_golden(name, kind):
    """The reference's default training loss of a flow step -- 0.40984 * mse(gt, xhat) + 0.59016 * NLL with xhat from the
    inverse pass on a sampled z (CWFA.py:905-911,952-987; main.py:43,107), and its L1 variant -- forward values and
    every gradient against the reference's own autograd (fixture g13 `grad_l2/`, `grad_l1/`)."""
    from cwfa_amd import training
    fx, g = _golden_step(name)
    cu = lambda k: torch.from_numpy(fx[k]).cuda()       # noqa: E731
    c = [cu("c0"), cu("c1")]
    out = training.step_backward(g, cu("x"), c, low=cu("full/low_in"), z=cu("full/z_in"), cond_weight=float(fx["full/w_c"]),
                                 loss_func=kind.upper(), want_cond_grads=True)
    if kind == "l1":
        pass                                            # xhat is the same reconstruction for both losses
    assert_close(out["xhat"], fx["full/xhat"], TOL, "xhat")
    for key, ref in (("full_loss", f"full_{kind}/loss"), ("recon", f"full_{kind}/recon")):
        assert abs(float(out[key]) - float(fx[ref])) <= 1e-5 * abs(float(fx[ref])), key
    want = {k[len(f"grad_{kind}/"):]: v for k, v in fx.items() if k.startswith(f"grad_{kind}/")}
    got = {k: p.grad for k, p in g.named_parameters() if p.grad is not None}
    assert set(got) == set(want) and len(want) == 80
    for k in sorted(want):
        assert_close(got[k], want[k], TOL if kind == "l2" else 5e-4, k)
    assert_close(out["cond_grads"][0], fx[f"full_{kind}/gc0"], TOL if kind == "l2" else 5e-4, "d loss / d omega")
    assert_close(out["cond_grads"][1], fx[f"full_{kind}/gc1"], TOL if kind == "l2" else 5e-4, "d loss / d mean detail")


@pytest.mark.parametrize("cfg", [(2, 6, 9, 11, 5), (1, 12, 20, 70, 32), (1, 3, 5, 130, 8), (3, 1, 4, 4, 2), (2, 5, 7, 36, 9),
                                 (1, 8, 16, 64, 32)])   # (B, D, H, W, K); W % 4 == 0 takes the four-voxel input-gradient kernel
def test_conv3d_stage_backward_vs_autograd(cfg):
    """Conv3d(1->K) -> PReLU -> Conv3d(K->1) over (H, W, depth) (networks.py:221-225,239): every gradient against float64
    autograd -- input, both filter banks, both biases, the PReLU slope."""
    from cwfa_amd import ops
    B, D, H, W, K = cfg
    g0 = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(B, D, H, W, generator=g0)
    dy = torch.randn(B, D, H, W, generator=g0)
    w1 = torch.randn(K, 1, 3, 3, 3, generator=g0) * 0.3
    b1 = torch.randn(K, generator=g0) * 0.1
    w2 = torch.randn(1, K, 3, 3, 3, generator=g0) * 0.3
    b2 = torch.randn(1, generator=g0)
    alpha = torch.tensor([0.25])
    leaves = [t.double().requires_grad_() for t in (x, w1, b1, w2, b2, alpha)]
    xd, w1d, b1d, w2d, b2d, ad = leaves
    vol = xd.permute(0, 2, 3, 1).unsqueeze(1)                                  # [B,1,H,W,D] as networks.py:239
    y = F.conv3d(F.prelu(F.conv3d(vol, w1d, b1d, padding=1), ad), w2d, b2d, padding=1)[:, 0].permute(0, 3, 1, 2)
    (y * dy.double()).sum().backward()
    with torch.no_grad():
        got_y = ops.conv3d_1k1(x.cuda(), w1.cuda(), b1.cuda(), alpha.cuda(), w2.cuda(), b2.cuda())
    assert_close(got_y, y.detach(), 1e-5, "forward")
    dx, dW1, db1, dW2, db2, dalpha = ops.conv3d_1k1_backward(x.cuda(), dy.cuda(), w1.cuda(), b1.cuda(), alpha.cuda(), w2.cuda())
    from conftest import rel_err
    bad = []
    for name, got, ref in (("dx", dx, xd.grad), ("dW1", dW1, w1d.grad), ("db1", db1, b1d.grad), ("dW2", dW2, w2d.grad),
                           ("db2", db2, b2d.grad), ("dalpha", dalpha, ad.grad)):
        e = max(rel_err(got, ref))
        if not e <= TOL:
            bad.append((name, e))
    assert not bad, (cfg, bad)


def test_prelu_backward():
    from cwfa_amd import ops
    g0 = torch.Generator().manual_seed(8)
    q = torch.randn(2, 5, 7, 9, generator=g0, dtype=torch.float64, requires_grad=True)
    a = torch.tensor([0.3], dtype=torch.float64, requires_grad=True)
    o = F.prelu(q, a)
    gup = torch.randn(2, 5, 7, 9, generator=g0)
    o.backward(gup.double())
    dalpha = torch.zeros(1, dtype=torch.float64, device="cuda")
    got = ops.prelu_bwd(gup.cuda(), o.detach().float().cuda(), a.detach().float().cuda(), dalpha)
    assert_close(got, q.grad, 1e-6, "prelu backward")
    assert_close(dalpha, a.grad, 1e-5, "d alpha")


@pytest.mark.parametrize("name", ["g13_step_grad_k0_ch8", "g13_step_grad_k1_ch8"])
def test_training_loss_backward_into_the_condition_net_golden(name):
    """The default training loss with the condition computed by the step's own condition net (eval mode, CWFA.py:527-528,
    893): gradients of the condition net's parameters (`optimizer_cond`) and of the flow step against the reference's
    autograd (fixture g13 `condgrad/`, `flowgrad_cond/`)."""
    from conftest import sd_of
    from cwfa_amd import networks as N, training
    from test_host_logic import build_step
    fx, g = _golden_step(name)
    keep = N.networks_n_chans
    try:
        cond_net, _ = build_step("CAT", int(fx["ix"]), n_ch=8)
    finally:
        N.networks_n_chans = keep
    cond_net.load_state_dict(sd_of(fx, "condsd/"))
    cond_net = cond_net.eval().cuda()
    for p in cond_net.parameters():       # the PReLU is ONE module shared by every ResidualBlock of the process (the reference's
        p.grad = None                     # default-argument instance, networks.py:200): clear what earlier tests left on it
    cu = lambda k: torch.from_numpy(fx[k]).cuda()       # noqa: E731
    omega, ctape = training.cond_forward_train(cond_net, cu("cond/views"))
    assert_close(omega, fx["cond/omega"], TOL, "omega")
    out = training.step_backward(g, cu("x"), [omega, cu("c1")], low=cu("full/low_in"), z=cu("full/z_in"),
                                 cond_weight=float(fx["full/w_c"]), loss_func="L2", want_cond_grads=True)
    assert abs(float(out["full_loss"]) - float(fx["cond/loss"])) <= 1e-5 * abs(float(fx["cond/loss"]))
    training.cond_backward(ctape, out["cond_grads"][0])
    want = {k[len("condgrad/"):]: v for k, v in fx.items() if k.startswith("condgrad/")}
    got = {k: p.grad for k, p in cond_net.named_parameters() if p.grad is not None}
    assert set(got) == set(want), sorted(set(got) ^ set(want))
    for k in sorted(want):
        assert_close(got[k], want[k], TOL, "condition net " + k)
    wantf = {k[len("flowgrad_cond/"):]: v for k, v in fx.items() if k.startswith("flowgrad_cond/")}
    gotf = {k: p.grad for k, p in g.named_parameters() if p.grad is not None}
    for k in sorted(wantf):
        assert_close(gotf[k], wantf[k], TOL, "flow step " + k)
    if "drop/mask" in fx:
        # the optimised step's condition net in TRAIN mode (CWFA.py:768,859): Dropout3d(0.5) on the hidden Conv3d channels
        # with the draw pinned by the fixture's keep / scale table
        for p in list(cond_net.parameters()) + list(g.parameters()):
            p.grad = None
        omega, ctape = training.cond_forward_train(cond_net, cu("cond/views"), drop_mask=cu("drop/mask"))
        assert_close(omega, fx["drop/omega"], TOL, "omega with dropout")
        out = training.step_backward(g, cu("x"), [omega, cu("c1")], low=cu("full/low_in"), z=cu("full/z_in"),
                                     cond_weight=float(fx["full/w_c"]), loss_func="L2", want_cond_grads=True)
        assert abs(float(out["full_loss"]) - float(fx["drop/loss"])) <= 1e-5 * abs(float(fx["drop/loss"]))
        training.cond_backward(ctape, out["cond_grads"][0])
        want = {k[len("dropgrad/"):]: v for k, v in fx.items() if k.startswith("dropgrad/")}
        got = {k: p.grad for k, p in cond_net.named_parameters() if p.grad is not None}
        assert set(got) == set(want), sorted(set(got) ^ set(want))
        for k in sorted(want):
            assert_close(got[k], want[k], TOL, "condition net (dropout) " + k)
        # train mode without an explicit table draws one from torch's RNG: kept channels are scaled by 2, dropped ones vanish
        cond_net.train()
        try:
            _, t2 = training.cond_forward_train(cond_net, cu("cond/views"))
            assert t2.drop_mask is not None and set(t2.drop_mask.unique().tolist()) <= {0.0, 2.0}
        finally:
            cond_net.eval()


def test_full_size_directional_derivative():
    """BASELINE.json configs[3] per-rank shape (512x512x96, the finest flow step, 64 internal channels) -- too large for the
    CPU oracle's autograd, so a size-independent property: along a random direction v in parameter space the central
    difference of the loss, (L(theta + e v) - L(theta - e v)) / 2e, equals <grad, v> from step_backward."""
    from cwfa_amd import CWFA, training
    torch.manual_seed(0)
    conv_inn, cond_nets = CWFA.build_networks(96, 512, 2, with_lrnn=False, device="cuda")
    g = conv_inn[0].train()
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(1, 96, 512, 512, generator=gen).cuda()
    c = [torch.randn(1, 48, 512, 512, generator=gen).cuda(), (0.1 * torch.randn(1, 48, 512, 512, generator=gen)).cuda()]
    low = torch.randn(1, 48, 512, 512, generator=gen).cuda()
    out = training.step_backward(g, x, c, low=low)
    params = [p for p in g.parameters() if p.grad is not None]
    assert len(params) == 80
    vs = [torch.randn(p.shape, generator=gen).cuda() * p.detach().abs().mean().clamp_min(1e-3) for p in params]
    slope = sum(float((p.grad.double() * v.double()).sum()) for p, v in zip(params, vs))
    eps = 2e-2

    def loss_at(sign):
        with torch.no_grad():
            for p, v in zip(params, vs):
                p.add_(v, alpha=sign * eps)
        try:
            for p in params:
                p.grad = None
            return float(training.step_backward(g, x, c, low=low)["full_loss"])
        finally:
            with torch.no_grad():
                for p, v in zip(params, vs):
                    p.add_(v, alpha=-sign * eps)

    fd = (loss_at(+1) - loss_at(-1)) / (2 * eps)
    assert abs(fd - slope) <= 2e-2 * abs(slope) + 1e-7, (fd, slope, float(out["full_loss"]))


@pytest.mark.parametrize("bias", [0, 1])
def test_unet_backward_golden(bias):
    """UNet in train mode (batch-statistics BatchNorm, PReLU, max-pool, skip adds, transposed convs; unet.py:72-113,161-195):
    forward, input gradient and every parameter gradient against the reference's own autograd (fixture g14)."""
    from conftest import load_golden, sd_of
    from cwfa_amd import training
    from cwfa_amd.unet import UNet
    fx = load_golden(f"g14_unet_grad_bias{bias}")
    u = UNet(5, 4, depth=3, wf=3, drop_out=0, use_bias=bool(bias), skip_conn=True, up_mode="upconv", batch_norm=True)
    u.load_state_dict(sd_of(fx))
    u = u.train().cuda()
    out, tape = training.unet_forward_train(u, torch.from_numpy(fx["x"]).cuda())
    assert_close(out, fx["y"], TOL, "train-mode forward")
    gx = training.unet_backward(tape, torch.from_numpy(fx["dy"]).cuda())
    assert_close(gx, fx["gx"], TOL, "input gradient")
    want = {k[len("grad/"):]: v for k, v in fx.items() if k.startswith("grad/")}
    got = {k: p.grad for k, p in u.named_parameters() if p.grad is not None}
    assert set(got) == set(want), sorted(set(got) ^ set(want))[:6]
    from conftest import rel_err
    # a PReLU slope gradient is ONE number, a sum with heavy cancellation over every element of a feature map: the
    # reference's own fp32 autograd carries ~1e-4 of relative noise there, so scalars get 1e-3
    bad = [(k, max(rel_err(got[k], want[k]))) for k in sorted(want)
           if not max(rel_err(got[k], want[k])) <= (1e-3 if want[k].size == 1 else TOL)]
    assert not bad, bad[:8]


def test_lrnn_mean_branch_backward_golden():
    """The LRNN's mean-volume branch at a small size (two ConvNeXt blocks: 1x1, 7x7, LayerNorm over (C,H,W), 1x1 + GELU,
    residual; GlobalAttention; out = x + 2 m (att - 0.5); networks.py:468-503,244-262,552-554): forward and every gradient
    against the reference's own autograd (fixture g15)."""
    from conftest import load_golden, rel_err, sd_of
    from cwfa_amd import networks as N, ops, training
    fx = load_golden("g15_meanbranch_grad")
    cn1, cn2, ga = N.ConvNeXt(6, 10, drop_prob=0.0, size=16), N.ConvNeXt(10, 6, drop_prob=0.0, size=16), N.GlobalAttention(6)
    for tag, mod in (("cn1", cn1), ("cn2", cn2), ("ga", ga)):
        mod.load_state_dict(sd_of(fx, f"sd_{tag}/"))
        mod.train().cuda()
    cu = lambda k: torch.from_numpy(fx[k]).cuda()       # noqa: E731
    mean, x, dy = cu("mean"), cu("x"), cu("dy")
    m1, t1 = training._convnext_forward_train(cn1, mean)
    m, t2 = training._convnext_forward_train(cn2, m1)
    assert_close(m, fx["m"], TOL, "m = ConvNeXt(ConvNeXt(mean))")
    out = ga.combine(mean, m, x)
    assert_close(out, fx["out"], TOL, "combined output")
    att = ga.m
    g_m, pg = ops.attention_bwd(mean, att[0].weight, att[0].bias, att[2].weight, att[2].bias, m, dy)
    g_m1 = training._convnext_backward(t2, g_m, True)
    training._convnext_backward(t1, g_m1, False)
    bad = []
    n1, n2 = 6 * 6 * 3, 6 * 6
    got_ga = {"m.0.weight": pg[:n1].reshape(6, 6, 3), "m.0.bias": pg[n1:n1 + 6], "m.2.weight": pg[n1 + 6:n1 + 6 + n2].reshape(6, 6, 1),
              "m.2.bias": pg[n1 + 6 + n2:]}
    for k, v in got_ga.items():
        e = max(rel_err(v, fx["grad_ga/" + k]))
        if not e <= TOL:
            bad.append(("ga." + k, e))
    for tag, mod in (("cn1", cn1), ("cn2", cn2)):
        want = {k[len(f"grad_{tag}/"):]: v for k, v in fx.items() if k.startswith(f"grad_{tag}/")}
        got = {k: p.grad for k, p in mod.named_parameters() if p.grad is not None}
        assert set(got) == set(want), (tag, sorted(set(got) ^ set(want)))
        for k in sorted(want):
            e = max(rel_err(got[k], want[k]))
            if not e <= TOL:
                bad.append((tag + "." + k, e))
    assert not bad, bad


def test_full_size_lrnn_step_directional_derivative():
    """The last pyramid step at BASELINE's size (LRNN on 512x512 views with the mean-volume branch, 63.7 M parameters, L2 loss,
    CWFA.py:880-886,936-950; stochastic layers off): the central difference of the loss along a random direction in
    parameter space equals <grad, v> from lrnn_step_backward."""
    from cwfa_amd import networks as N, training
    torch.manual_seed(0)
    enc = N.Encoder(29, 6, 5, 64, True).cuda().train()
    lr = enc.net
    lr.deconv[1].drop_out = 0
    for cn in lr.conv3d:
        cn.drop_prob = 0.0
    gen = torch.Generator().manual_seed(21)
    views = torch.randn(1, 29, 512, 512, generator=gen).cuda()
    mean = (0.1 * torch.randn(1, 6, 512, 512, generator=gen)).cuda()
    gt = torch.randn(1, 6, 512, 512, generator=gen).cuda()
    loss, out = training.lrnn_step_backward(enc, views, mean, gt)
    assert out.shape == (1, 6, 512, 512)
    params = [p for p in lr.parameters() if p.grad is not None]
    assert sum(p.numel() for p in params) > 63e6
    vs = [torch.randn(p.shape, generator=gen).cuda() * p.detach().abs().mean().clamp_min(1e-3) for p in params]
    slope = sum(float((p.grad.double() * v.double()).sum()) for p, v in zip(params, vs))
    eps = 1e-2

    def loss_at(sign):
        with torch.no_grad():
            for p, v in zip(params, vs):
                p.add_(v, alpha=sign * eps)
        try:
            for p in params:
                p.grad = None
            return float(training.lrnn_step_backward(enc, views, mean, gt)[0])
        finally:
            with torch.no_grad():
                for p, v in zip(params, vs):
                    p.add_(v, alpha=-sign * eps)

    fd = (loss_at(+1) - loss_at(-1)) / (2 * eps)
    assert abs(fd - slope) <= 3e-2 * abs(slope) + 1e-7, (fd, slope, float(loss))


def test_full_size_training_iteration_reduces_the_loss():
    """The whole pyramid at BASELINE's size (512x512x96: LRNN + 4 flow steps with their condition nets, 67 M parameters), the
    reference's per-step order (CWFA.py:865-1027), torch.optim.Adam as the optimiser on the plain Parameters / .grad
    tensors: three iterations on one synthetic batch lower every step's loss."""
    from cwfa_amd import CWFA, training
    torch.manual_seed(0)
    np_seed = 0
    import numpy as np
    np.random.seed(np_seed)
    conv_inn, cond_nets = CWFA.build_networks(96, 512, 5, with_lrnn=True, device="cuda")
    enc = cond_nets[-1]
    enc.net.deconv[1].drop_out = 0
    for cn in enc.net.conv3d:
        cn.drop_prob = 0.0
    gen = torch.Generator().manual_seed(31)
    gt = torch.randn(1, 96, 512, 512, generator=gen).cuda()
    views = torch.randn(1, 29, 512, 512, generator=gen).cuda()
    means = [(0.1 * torch.randn(1, 96 // 2 ** (n + 1), 512, 512, generator=gen)).cuda() for n in range(4)]
    opts = []
    for n in range(5):
        mods = [cond_nets[n]] if n == 4 else [conv_inn[n], cond_nets[n]]
        opts.append(torch.optim.Adam([p for m in mods for p in m.parameters() if p.requires_grad], lr=1e-4))
    hist = []
    for _ in range(3):
        res = training.train_iteration(conv_inn, cond_nets, gt, views, means, optimizers=opts)
        hist.append([float(v) for v in res["losses"]])
    assert res["volume"].shape == (1, 96, 512, 512)
    assert all(np.isfinite(h).all() for h in hist), hist
    for n in range(5):
        assert hist[-1][n] < hist[0][n], (n, hist)


def test_full_size_iteration_through_autograd_equals_the_manual_path():
    """training.train_iteration_autograd -- the iteration as the reference writes it (modules as autograd nodes, loss with torch
    operators, full_loss.backward(), CWFA.py:865-1027) -- at 512x512x96 against training.train_iteration (the manual backward pinned
    to the reference's gradients by the g13-g15 fixtures): the same per-step losses, the same reconstruction and the same gradients
    (no optimiser step on either side), in split precision."""
    from cwfa_amd import CWFA, ops, training
    import numpy as np
    torch.manual_seed(0)
    np.random.seed(0)
    conv_inn, cond_nets = CWFA.build_networks(96, 512, 5, with_lrnn=True, device="cuda")
    enc = cond_nets[-1]
    enc.net.deconv[1].drop_out = 0                       # the stochastic layers off: two runs must see the same function
    for cn in enc.net.conv3d:
        cn.drop_prob = 0.0
    gen = torch.Generator().manual_seed(33)
    gt = torch.randn(1, 96, 512, 512, generator=gen).cuda()
    views = torch.randn(1, 29, 512, 512, generator=gen).cuda()
    means = [(0.1 * torch.randn(1, 96 // 2 ** (n + 1), 512, 512, generator=gen)).cuda() for n in range(4)]
    mods = list(conv_inn) + list(cond_nets)

    class Keep:                                          # an "optimiser" that only snapshots the step's gradients
        def __init__(self, ms):
            self.ps, self.grads = [p for m in ms for p in m.parameters() if p.requires_grad], None

        def step(self):
            self.grads = [None if p.grad is None else p.grad.clone() for p in self.ps]

        def zero_grad(self, set_to_none=True):
            for p in self.ps:
                p.grad = None

    def keepers():
        return [Keep([cond_nets[n]] if n == 4 else [conv_inn[n], cond_nets[n]]) for n in range(5)]

    ops.set_precision("split_bf16")
    try:
        bn_state = {k: v.clone() for k, v in enc.state_dict().items() if "running" in k or "num_batches" in k}
        ka = keepers()
        a = training.train_iteration_autograd(conv_inn, cond_nets, gt, views, means, optimizers=ka)
        enc.load_state_dict(bn_state, strict=False)      # (the LRNN's BatchNorm buffers moved: same starting point for the second run)
        km = keepers()
        m = training.train_iteration(conv_inn, cond_nets, gt, views, means, optimizers=km)
    finally:
        ops.set_precision("fp32")
    for n in range(5):
        assert abs(float(a["losses"][n]) - float(m["losses"][n])) <= 2e-5 * abs(float(m["losses"][n])), (n, a["losses"], m["losses"])
    assert_close(a["volume"], m["volume"], 1e-5, "finest reconstruction")
    for n in range(5):
        names = [k for m_ in ([cond_nets[n]] if n == 4 else [conv_inn[n], cond_nets[n]]) for k, p in m_.named_parameters() if p.requires_grad]
        worst = (0.0, "")
        for k, ga, gm in zip(names, ka[n].grads, km[n].grads):
            assert (ga is None) == (gm is None), k
            if ga is not None and float(gm.abs().max()) > 0:
                d = float((ga - gm).abs().max()) / float(gm.abs().max())
                if d > worst[0]:
                    worst = (d, k + " max|g| = %.3e" % float(gm.abs().max()))
        assert worst[0] <= 2e-5, (n, worst)
