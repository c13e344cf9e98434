"""Backward of the flow-step NLL and the data-parallel gradient exchange.

Reference: the training branch of ``run_CWFA`` -- ``Z, log_jac_det = conv_inn[n_net](curr_gt, c=cond_processed)``,
``curr_LL_loss = (0.5*||Z0||^2 - log_jac_det.mean()) / numel`` (CWFA.py:966-978) followed by
``scaler.scale(full_loss).backward()`` and the optimiser step (CWFA.py:1002-1027).  There torch autograd records every
stock op; here the step's structure is used instead:

* the coupling coefficients of a CAT step depend on the conditions only, so the forward is "evaluate each block's
  sub-network (kept unfused, its seven post-activation maps saved), then ONE fused chain launch";
* the flow is invertible, so the chain's backward needs no saved activations (``ops.chain_bwd``: one launch that walks
  the stages backwards from z, recomputing every stage input by inverting the stage);
* a sub-network's backward is 8 data-gradient convolutions on the forward MFMA kernels (filters transposed + flipped),
  8 weight-gradient GEMMs (``ops.conv2d_wgrad``), ELU-backward passes and per-channel sums for the biases.

Multi-GPU (one process per GPU, batch sharded as in SURVEY.md 8(e)): the loss is defined on the GLOBAL batch, so every
rank scales its local gradient by the global 1/numel and the ranks' gradients are SUMMED -- ``allreduce_gradients``
packs them into a few large flat buckets (default 64 MiB: the flow parameters of one step are 13.7 MB, so one message)
and issues one all-reduce per bucket (RCCL over xGMI on MI355X; gloo in the CPU test).

Scope: gradients of a flow step's own parameters (the reference's ``optimizer``), of its condition net
(``optimizer_cond`` for the flow steps: ``cond_forward_train`` / ``cond_backward``) and of the LRNN, the last step's
network (``lrnn_step_backward``: UNet + mean-volume branch, L1/L2 loss).
"""
from typing import Sequence

import torch

from . import ops

__all__ = ["subnet_forward_train", "subnet_backward", "nll_backward", "step_backward", "cond_forward_train", "cond_backward", "unet_forward_train", "unet_backward",
           "lrnn_forward_train", "lrnn_backward", "lrnn_step_backward", "train_iteration", "allreduce_gradients",
           "sgd_step"]


class _Tape:
    """Saved post-activation maps of one sub-network evaluation."""
    __slots__ = ("net", "conv_in", "conv_out", "u", "b", "h")

    def __init__(self, net, conv_in, conv_out, u):
        self.net, self.conv_in, self.conv_out, self.u = net, conv_in, conv_out, u
        self.b, self.h = [], []


def _layers(net):
    return (net.block2, net.block4, net.block6)


def subnet_forward_train(net, u, conv_in, conv_out):
    """``wavelet_flow_subnetwork._stack`` (networks.py:641-667) with every layer's output kept for the backward:
    b0 = conv_in(u);  h_i = ELU(conv3x3(b_{i-1}));  b_i = ELU(conv1x1(h_i) + b_{i-1});  a = conv_out(b_3)."""
    P = net._packed.get
    tape = _Tape(net, conv_in, conv_out, u)
    b = ops.conv2d(u, P(conv_in), bias=conv_in.bias)
    tape.b.append(b)
    fused = net.n_ch == 64 and all(blk[0].bias is not None and blk[2].bias is not None for blk in _layers(net))
    split = ops._split_bf16 >= 2
    for blk in _layers(net):
        if fused and split:     # the split-bf16 layer kernel in its tape form (cwfa_subnet_layer_split_tape_f32)
            b, h = ops.subnet_layer(b, net._split3(blk[0], blk[2]), blk[0].bias, None, blk[2].bias, want_hidden=True)
        elif fused:     # the inference layer kernel, which also writes the hidden map it holds in registers anyway
            b, h = ops.subnet_layer(b, P(blk[0]), blk[0].bias, net._panel(blk[2]), blk[2].bias, want_hidden=True)
        else:
            h = ops.conv2d(b, P(blk[0]), bias=blk[0].bias, act="elu")
            b = ops.conv2d(h, P(blk[2]), bias=blk[2].bias, residual=b, act2="elu")
        tape.h.append(h)
        tape.b.append(b)
    return ops.conv2d(b, P(conv_out), bias=conv_out.bias), tape


def _packT(conv):
    """Filter bank of the data-gradient convolution: W^T[ci][co][ky][kx] = W[co][ci][K-1-ky][K-1-kx], in kernel layout.
    Cached ON the module (a process-wide table keyed by id() would outlive the module and could hand a later module that
    happens to reuse the id, the version counter and the storage address somebody else's filter bank)."""
    w = conv.weight
    hit = getattr(conv, "_cwfa_packT", None)
    if hit is None or hit[1] != w._version or hit[2] != w.data_ptr() or hit[0].epoch != ops.pack_epoch():
        wt = w.detach().transpose(0, 1).flip(2, 3).contiguous()
        hit = (ops.pack_conv_weight(wt), w._version, w.data_ptr())
        object.__setattr__(conv, "_cwfa_packT", hit)
    return hit[0]


def _conv_param_grads(conv, x, dy):
    """Accumulate dL/dW, dL/db of ``y = conv(x)`` into conv.weight.grad / conv.bias.grad."""
    ks = conv.kernel_size[0]
    w, b = conv.weight, conv.bias
    if b is not None and (w.grad is None) != (b.grad is None):            # mixed state: give the missing one zeros
        for t in (w, b):
            if t.grad is None:
                t.grad = torch.zeros_like(t)
    if w.grad is None:
        if b is None:
            w.grad = ops.conv2d_wgrad(x, dy, ks)
        else:
            w.grad, b.grad = ops.conv2d_wgrad(x, dy, ks, want_bias=True)      # sum(dy) comes out of the same pass
    else:
        ops.conv2d_wgrad(x, dy, ks, out=w.grad, accumulate=True, bias_out=None if b is None else b.grad)


def subnet_backward(tape, g_a, want_input_grad=False):
    """Backward of ``subnet_forward_train``: accumulates the .grad of its eight convolutions from dL/da ``g_a``
    (overwritten as scratch) and returns dL/du if wanted."""
    net = tape.net
    _conv_param_grads(tape.conv_out, tape.b[3], g_a)
    g_b = ops.conv2d(g_a, _packT(tape.conv_out))
    for i in (2, 1, 0):
        blk = _layers(net)[i]
        b_in, h, b_out = tape.b[i], tape.h[i], tape.b[i + 1]
        g_r = ops.elu_bwd(g_b, b_out, out=g_b)                       # through the trailing ELU; also the residual branch
        _conv_param_grads(blk[2], h, g_r)
        g_q = ops.conv2d(g_r, _packT(blk[2]))
        ops.elu_bwd(g_q, h, out=g_q)
        _conv_param_grads(blk[0], b_in, g_q)
        g_b = ops.conv2d(g_q, _packT(blk[0]), residual=g_r)
    _conv_param_grads(tape.conv_in, tape.u, g_b)
    return ops.conv2d(g_b, _packT(tape.conv_in)) if want_input_grad else None


def nll_backward(graph, x, c, group=None, want_cond_grads=False):
    """One training-step forward + backward of a CAT flow step's NLL on this rank's batch shard.

    Returns ``(nll, Z, cond_grads)``: the global-batch NLL of CWFA.py:978 (identical on every rank), the forward
    outputs, and dL/d(condition) per condition tensor (or None).  Parameter gradients are accumulated into ``.grad``
    (LOCAL contributions: call ``allreduce_gradients`` before the optimiser step when running on several ranks)."""
    out = step_backward(graph, x, c, group=group, want_cond_grads=want_cond_grads, cond_weight=0.0)
    return out["nll"], out["Z"], out["cond_grads"]


def step_backward(graph, gt, c, low=None, z=None, cond_weight=0.40984, loss_func="L2", group=None, want_cond_grads=False):
    """Forward + backward of the reference's training loss of one CAT flow step (CWFA.py:905-911,952-987):

        full_loss = cond_weight * loss_func(gt, xhat) + (1 - cond_weight) * NLL,
        xhat = graph([z, low], c, rev=True)  (the reconstruction, z sampled by the caller or None = 0),
        NLL  = (0.5*||Z0||^2 - mean_b logdet) / numel,  Z = graph(gt, c).

    The reference evaluates every sub-network twice (inverse and forward pass) and lets autograd add the two gradient
    contributions; both passes see the same conditions, so here the sub-networks run ONCE (with a tape), the two chain
    backward kernels add their coefficient gradients in place, and one sub-network backward follows.
    ``cond_weight = 0`` or ``low is None``: the NLL alone.  ``loss_func``: "L1" | "L2" (main.py:43).
    Returns a dict: full_loss, nll, recon (mean), Z, xhat, cond_grads."""
    from .CWFA import allreduce_nll
    plan = getattr(graph, "_plan", None)
    if plan is None or hasattr(plan, "rest") or any(k == "act" for k, _ in plan.chain):    # no plan / mixed plan / ActNorm stages
        raise NotImplementedError("step_backward: only conditional-affine (CAT) steps lowered to a chain plan are built")
    if loss_func not in ("L1", "L2"):
        raise NotImplementedError(f"step_backward: loss_func {loss_func!r} (L1 and L2 are built)")
    recon = low is not None and cond_weight != 0.0
    w_c = float(cond_weight) if recon else 0.0
    x = gt
    cond_of = dict(zip(graph.condition_nodes, c))
    tapes, cache = [], {}

    def coefficients(module, parts):
        hit = cache.get(id(module))
        if hit is not None:                              # second direction: same conditions, same coefficients
            return hit
        net, n_s = module.subnet, module.channels
        if not hasattr(net, "block12"):
            raise NotImplementedError("step_backward: sub-network type without a HIP backward")
        if net.normal:
            u = parts[0] if len(parts) == 1 else ops.concat_channels(parts)
            a, tape = subnet_forward_train(net, u, net.block12, net.block72[1])
            tapes.append((tape, a, parts, "normal"))
            res = (a[:, :n_s], a[:, n_s:], False)
        else:
            n = net.c_in // 2
            if not (len(parts) == 2 and parts[1].shape[1] == n and net.c_out // 2 == n_s):
                raise NotImplementedError("step_backward: `_first` sub-network with an unexpected condition layout")
            mean, om = parts
            a, tape = subnet_forward_train(net, om, net.block1, net.block7[1])
            tapes.append((tape, a, parts, "first"))
            res = (a, mean, True)
        cache[id(module)] = res
        return res

    stages, pending = plan._stages(cond_of, False, coefficients=coefficients)
    final_perm = None
    if pending is not None:
        if pending[1] == 1:
            final_perm = pending[0]
        else:
            stages.append(ops.stage(None, None, perm=pending[0], axis=pending[1]))
    xhat = None
    if recon:
        rstages, rpending = plan._stages(cond_of, True, coefficients=coefficients)
        if rpending is not None:
            rstages.append(ops.stage(None, None, perm=rpending[0], axis=rpending[1]))
        xhat = ops.chain_inv(z, low, rstages)
    B = x.shape[0]
    logdet = torch.zeros(B, dtype=torch.float64, device=x.device)
    sumsq = torch.zeros(1, dtype=torch.float64, device=x.device)
    zf, lowf = ops.chain_fwd(x, stages, final_perm, logdet=logdet, sumsq=sumsq)
    terms = torch.stack([sumsq[0], logdet.sum(), torch.tensor(float(B), dtype=torch.float64, device=x.device)])
    terms = allreduce_nll(terms, group)
    Bg = float(terms[2])
    numel_total = Bg * x[0].numel()             # `upsampled_vol.numel()`: the step's whole input volume (CWFA.py:911,978)
    nll = (0.5 * terms[0] - terms[1] / terms[2]) / numel_total

    # backward: the chains (one launch per direction), then every block's sub-network
    Cc, H, W = zf.shape[1:]
    grads, holders = [], []
    it = iter(tapes)
    for st, _keep in stages:
        if not st.s_raw and not st.t:
            grads.append((None, None))
            continue
        tape, a, parts, kind = next(it)
        g_a = torch.empty_like(a)
        if kind == "normal":
            grads.append((g_a[:, :Cc], g_a[:, Cc:]))
            holders.append((tape, g_a, None, parts))
        else:
            g_mean = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device) if want_cond_grads else None
            grads.append((g_a, g_mean))
            holders.append((tape, g_a, g_mean, parts))
    recon_mean = None
    if recon:
        kind = 2 if loss_func == "L2" else 1
        rsum = ops.chain_inv_bwd(xhat, x, stages, grads, gscale=w_c * kind / numel_total, loss_kind=kind)
        rsum = allreduce_nll(rsum, group)
        recon_mean = rsum[0] / numel_total
    ops.chain_bwd(zf, stages, grads, final_perm, gscale=(1.0 - w_c) / numel_total, ldscale=(1.0 - w_c) / (Bg * numel_total),
                  accumulate=recon)
    cond_grads = None
    if want_cond_grads:
        cond_grads = {id(t): None for t in c}
    for tape, g_a, g_mean, parts in holders:
        g_u = subnet_backward(tape, g_a, want_input_grad=want_cond_grads)
        if not want_cond_grads:
            continue
        if g_mean is not None:                           # `_first`: parts = (mean, omega)
            pieces = [g_mean, g_u]
        else:
            pieces, c0 = [], 0
            for t in parts:
                pieces.append(g_u[:, c0:c0 + t.shape[1]])
                c0 += t.shape[1]
        for t, g in zip(parts, pieces):
            cur = cond_grads.get(id(t))
            cond_grads[id(t)] = g.contiguous() if cur is None else ops.axpby(g, 1.0, cur, 1.0)
    if want_cond_grads:
        cond_grads = [cond_grads[id(t)] for t in c]
    full = (1.0 - w_c) * nll + (w_c * recon_mean if recon else 0.0)
    return {"full_loss": full, "nll": nll, "recon": recon_mean, "Z": (zf, lowf), "xhat": xhat, "cond_grads": cond_grads}


class _CondTape:
    __slots__ = ("block", "x", "o1", "o", "drop_mask")

    def __init__(self, block, x, o1, o, drop_mask=None):
        self.block, self.x, self.o1, self.o, self.drop_mask = block, x, o1, o, drop_mask


def cond_forward_train(cond_net, views, drop_mask=None):
    """``cond_network.forward`` (networks.py:165-242) with the two 2-D maps the backward needs kept.  Returns (omega, tape).

    Mode: the reference builds the flow steps' condition nets in eval mode (CWFA.py:527-528) but switches the OPTIMISED
    step's net to train mode (``cond_nets[steps_to_optimize[0]].train()``, CWFA.py:768,859), so its ``Dropout3d(p=0.5)`` on
    the hidden Conv3d channels is live (networks.py:224).  Here: if the block is in train mode a keep / scale table
    ``drop_mask`` [B, K] (entries 0 or 1/(1-p)) is drawn from torch's RNG (or taken from the argument: fixtures pin a fixed
    mask).  Dropout3d zeroes whole hidden channels per sample and the second Conv3d is linear in them, so the mask is
    FOLDED INTO THAT CONV'S WEIGHTS per sample, w2[b, k] = mask[b, k] * w2[k]: the fused 1 -> K -> 1 kernel and its backward
    run unchanged, one launch per sample."""
    blk = cond_net.subnetworks[0] if hasattr(cond_net, "subnetworks") else cond_net      # a cond_network or its ResidualBlock
    a = blk.relu.weight
    P = blk._packed.get
    c1, c2, ds = blk.conv1[0], blk.conv2[0], blk.downsample[0]
    o1 = ops.conv2d(views, P(c1), bias=c1.bias, act="prelu", prelu_alpha=a)
    r = ops.conv2d(views, P(ds), bias=ds.bias)
    o = ops.conv2d(o1, P(c2), bias=c2.bias, residual=r, act2="prelu", prelu_alpha=a)
    k1, k2 = blk.conv3d[0], blk.conv3d[3]
    if drop_mask is None and blk.training:
        p_drop = float(getattr(blk.conv3d[2], "p", 0.0))
        K = k1.weight.shape[0]
        if p_drop > 0.0:
            keep = (torch.rand(o.shape[0], K, device=o.device) >= p_drop).to(torch.float32)
            drop_mask = keep / (1.0 - p_drop)
    if drop_mask is None:
        omega = ops.conv3d_1k1(o, k1.weight, k1.bias, a, k2.weight, k2.bias)
    else:
        drop_mask = drop_mask.to(device=o.device, dtype=torch.float32)
        if tuple(drop_mask.shape) != (o.shape[0], k1.weight.shape[0]):
            raise ValueError(f"cond_forward_train: drop_mask must be [B, K] = {(o.shape[0], k1.weight.shape[0])}")
        omega = torch.cat([ops.conv3d_1k1(o[b:b + 1], k1.weight, k1.bias, a, _masked_w2(k2.weight, drop_mask[b]), k2.bias)
                           for b in range(o.shape[0])], 0)
    return omega, _CondTape(blk, views, o1, o, drop_mask)


def _masked_w2(w2, mask_b):
    """Conv3d(K -> 1) weight [1, K, 3, 3, 3] with hidden channel k scaled by mask_b[k]."""
    return (w2.detach() * mask_b.view(1, -1, 1, 1, 1)).contiguous()


def _acc(param, g):
    g = g.to(param.dtype).reshape(param.shape)
    if param.grad is None:
        param.grad = g.clone()
    else:
        param.grad.add_(g)


def cond_backward(tape, g_omega):
    """Backward of ``cond_forward_train`` from dL/d(omega): accumulates the .grad of the three 2-D convolutions, the two
    Conv3d layers and the (shared, single-parameter) PReLU slope -- the reference's ``optimizer_cond`` parameters
    (CWFA.py:1008-1012).  The slope must be positive (the pre-activations are recovered from the outputs)."""
    blk = tape.block
    a = blk.relu.weight
    if not float(a.detach()) > 0.0:
        raise NotImplementedError("cond_backward: PReLU slope <= 0 (pre-activations are recovered from the layer outputs)")
    c1, c2, ds = blk.conv1[0], blk.conv2[0], blk.downsample[0]
    k1, k2 = blk.conv3d[0], blk.conv3d[3]
    if tape.drop_mask is None:
        g_o, dW1, db1, dW2, db2, dalpha = ops.conv3d_1k1_backward(tape.o, g_omega, k1.weight, k1.bias, a, k2.weight)
    else:            # per sample with the masked second-conv weights; d/dw2[k] = mask[b, k] * d/d(masked weight)
        parts = [ops.conv3d_1k1_backward(tape.o[b:b + 1], g_omega[b:b + 1], k1.weight, k1.bias, a, _masked_w2(k2.weight, tape.drop_mask[b]))
                 for b in range(tape.o.shape[0])]
        g_o = torch.cat([p_[0] for p_ in parts], 0)
        dW1, db1 = sum(p_[1] for p_ in parts), sum(p_[2] for p_ in parts)
        dW2 = sum(p_[3] * tape.drop_mask[b].view(1, -1, 1, 1, 1).to(p_[3].dtype) for b, p_ in enumerate(parts))
        db2, dalpha = sum(p_[4] for p_ in parts), sum(p_[5] for p_ in parts)
    _acc(k1.weight, dW1)
    _acc(k1.bias, db1)
    _acc(k2.weight, dW2)
    _acc(k2.bias, db2)
    g_pre = ops.prelu_bwd(g_o, tape.o, a, dalpha, out=g_o)           # o = PReLU(conv2(o1) + downsample(x))
    _conv_param_grads(c2, tape.o1, g_pre)
    _conv_param_grads(ds, tape.x, g_pre)
    g_o1 = ops.conv2d(g_pre, _packT(c2))
    ops.prelu_bwd(g_o1, tape.o1, a, dalpha, out=g_o1)               # o1 = PReLU(conv1(x))
    _conv_param_grads(c1, tape.x, g_o1)
    _acc(a, dalpha.to(torch.float32))


# ---------------------------------------------------------------------------------------------------------------------
# UNet of the LRNN (unet.py:72-113,161-195), train-mode BatchNorm (CWFA.py:532)
# ---------------------------------------------------------------------------------------------------------------------
class _LayerRec:
    __slots__ = ("conv", "alpha", "bn", "u", "y", "mean", "invstd", "mask", "n", "batch_stats")


def _prelu_of(act):
    import torch.nn as nn
    if not isinstance(act, nn.PReLU) or act.weight.numel() != 1:
        raise NotImplementedError("UNet backward: single-parameter PReLU activations only (what the LRNN builds)")
    if not float(act.weight.detach()) > 0.0:
        # the backward kernels recover the pre-activation sign from the layer OUTPUT and form d(alpha) as g*y/alpha
        # (conv_bwd.hip: bn_act_bwd / prelu_bwd): a slope <= 0 would give silently wrong gradients or NaN
        raise NotImplementedError("UNet backward: PReLU slope <= 0 (pre-activations are recovered from the layer outputs)")
    return act.weight


def _block_forward_train(block, u, out_mask):
    """UNetConvBlock (conv3x3 -> PReLU -> BatchNorm) x 2, unfused: every BatchNorm output is materialised (the inference
    path applies it inside the next convolution's load) because the weight gradient needs it as an operand."""
    recs = []
    layers = block._layers()
    for li, (conv, act, bn) in enumerate(layers):
        if bn is None:
            raise NotImplementedError("UNet backward: blocks without BatchNorm")
        alpha = _prelu_of(act)
        y = ops.conv2d(u, block._packed.get(conv), bias=conv.bias, act="prelu", prelu_alpha=alpha)
        m = out_mask if li == len(layers) - 1 else None
        Cc = y.shape[1]
        n = y.numel() // Cc
        batch_stats = bn.training or not bn.track_running_stats
        if batch_stats:
            st = ops.channel_stats(y)
            if bn.track_running_stats and bn.momentum is not None:
                ops.bn_running_update(st, n, bn.momentum, bn.running_mean, bn.running_var, bn.num_batches_tracked)
            sv = st.view(Cc, 2)
            mean = sv[:, 0] / n
            invstd = torch.rsqrt(sv[:, 1] / n - mean * mean + bn.eps)
        else:                                   # eval mode: the running statistics are constants of the graph
            mean = bn.running_mean.detach().double()
            invstd = torch.rsqrt(bn.running_var.detach().double() + bn.eps)
        scale = bn.weight.detach().double() * invstd
        shift = bn.bias.detach().double() - mean * scale
        if m is not None:
            scale, shift = m.double() * scale[None, :], m.double() * shift[None, :]
        r = _LayerRec()
        r.conv, r.alpha, r.bn, r.u, r.y, r.mean, r.invstd, r.mask, r.n = conv, alpha, bn, u, y, mean, invstd, m, n
        r.batch_stats = batch_stats
        recs.append(r)
        u = ops.plane_affine(y, scale.float(), shift.float())
    return u, recs


def _block_backward(recs, g):
    """dL/d(block output) -> dL/d(block input); accumulates conv / BatchNorm / PReLU gradients."""
    for r in reversed(recs):
        st = ops.bn_bwd_stats(g, r.y, r.mask)
        s1, s2 = st[:, 0], st[:, 1]
        s2h = (s2 - r.mean * s1) * r.invstd                        # sum g m xhat
        k = r.bn.weight.detach().double() * r.invstd
        A = k if r.mask is None else r.mask.double() * k[None, :]
        if r.batch_stats:
            Cc = -k * r.invstd * s2h / r.n
            Bc = -k * s1 / r.n - Cc * r.mean
        else:                                   # running statistics: a fixed per-channel affine, no statistics terms
            Cc = torch.zeros_like(k)
            Bc = torch.zeros_like(k)
        _acc(r.bn.weight, s2h)
        _acc(r.bn.bias, s1)
        dalpha = torch.zeros(1, dtype=torch.float64, device=g.device)
        g_pre = ops.bn_act_bwd(g, r.y, A.float(), Bc.float(), Cc.float(), r.alpha, dalpha)
        _acc(r.alpha, dalpha)
        _conv_param_grads(r.conv, r.u, g_pre)
        g = ops.conv2d(g_pre, _packT(r.conv))
    return g


class _UNetTape:
    __slots__ = ("unet", "down", "pools", "ups", "last_u", "out")


def unet_forward_train(unet, x):
    """``UNet.forward`` (unet.py:72-91) in training mode with everything the backward needs kept.  Returns (out, tape)."""
    from .unet import _drop_mask
    B, dev = x.shape[0], x.device
    t = _UNetTape()
    t.unet, t.down, t.pools, t.ups = unet, [], [], []
    u, skips = x, []
    for i, down in enumerate(unet.down_path):
        u, recs = _block_forward_train(down, u, None)
        t.down.append(recs)
        if i != len(unet.down_path) - 1:
            H, W = u.shape[2:]
            if H % 2 or W % 2:
                raise NotImplementedError("UNet backward: even feature-map sizes only")
            full = u
            pooled = ops.maxpool(full, H // 2, W // 2)
            m = _drop_mask(unet.drop_out, B, pooled.shape[1], dev)
            u = pooled if m is None else ops.plane_affine(pooled, m, torch.zeros_like(m))
            t.pools.append((full, m))
            skips.append(full)
    for i, up in enumerate(unet.up_path):
        m = _drop_mask(unet.drop_out, B, up.conv_block.block[0].out_channels, dev)
        upv = ops.conv2d(u, up._packed.get(up.up, transposed=True), bias=up.up.bias)
        v = ops.plane_affine(upv, add=up.center_crop(skips[-i - 1], upv.shape[2:])) if up.skip_conn else upv
        u_in = u
        u, recs = _block_forward_train(up.conv_block, v, m)
        t.ups.append((up, u_in, recs))
    alpha = _prelu_of(unet.last[1])
    t.last_u = u
    t.out = ops.conv2d(u, unet._packed.get(unet.last[0]), bias=unet.last[0].bias, act="prelu", prelu_alpha=alpha)
    return t.out, t


def _packT4(convT):
    """1x1 filter bank of a ConvTranspose2d(k2,s2)'s data gradient: [Cin][Co*4] (cout index c*4 + dy*2 + dx, as the forward)."""
    w = convT.weight
    hit = getattr(convT, "_cwfa_packT", None)
    if hit is None or hit[1] != w._version or hit[2] != w.data_ptr() or hit[0].epoch != ops.pack_epoch():
        cin = w.shape[0]
        hit = (ops.pack_conv_weight(w.detach().reshape(cin, -1, 1, 1).contiguous()), w._version, w.data_ptr())
        object.__setattr__(convT, "_cwfa_packT", hit)
    return hit[0]


def unet_backward(tape, g_out):
    """Backward of ``unet_forward_train``: accumulates every parameter gradient, returns dL/d(input)."""
    unet = tape.unet
    alpha = _prelu_of(unet.last[1])
    dalpha = torch.zeros(1, dtype=torch.float64, device=g_out.device)
    g_pre = ops.prelu_bwd(g_out, tape.out, alpha, dalpha)
    _acc(alpha, dalpha)
    _conv_param_grads(unet.last[0], tape.last_u, g_pre)
    g = ops.conv2d(g_pre, _packT(unet.last[0]))
    n_skip = len(tape.pools)
    g_skips = [None] * n_skip
    for i in reversed(range(len(tape.ups))):
        up, u_in, recs = tape.ups[i]
        g_v = _block_backward(recs, g)                              # v = up(u_in) + skip
        if up.skip_conn:
            g_skips[n_skip - 1 - i] = g_v
        B, Co, H2, W2 = g_v.shape
        g4 = g_v.view(B, Co, H2 // 2, 2, W2 // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(B, Co * 4, H2 // 2, W2 // 2).contiguous()
        w = up.up.weight
        if up.up.bias is not None:
            _acc(up.up.bias, ops.channel_stats(g_v).view(-1, 2)[:, 0])
        dW4 = ops.conv2d_wgrad(u_in, g4, 1)                          # [4 Co, Cin, 1, 1]
        _acc(w, dW4.view(Co, 2, 2, w.shape[0]).permute(3, 0, 1, 2))
        g = ops.conv2d(g4, _packT4(up.up))
    for i in reversed(range(len(tape.down))):
        if i != len(tape.down) - 1:
            full, m = tape.pools[i]
            g_pool = g if m is None else ops.plane_affine(g, m, torch.zeros_like(m))
            g = ops.maxpool2_bwd(full, g_pool, g_skips[i])
        g = _block_backward(tape.down[i], g)
    return g


def _convnext_forward_train(cn, x):
    """ConvNeXt (networks.py:468-503): u = 1x1(x); out = GELU(1x1(LayerNorm_{C,H,W}(7x7(u)))) + drop_path(u), unfused."""
    P = cn._packed.get
    u = ops.conv2d(x, P(cn.input), bias=cn.input.bias)
    v = ops.conv2d(u, P(cn.m[0]), bias=cn.m[0].bias)
    ln = cn.m[1]
    B, n = v.shape[0], v[0].numel()
    st = ops.sample_stats(v).view(B, 2)
    mean = st[:, 0] / n
    invstd = torch.rsqrt(st[:, 1] / n - mean * mean + ln.eps)
    vln = ops.layernorm_apply(v, st.reshape(-1), ln.weight, ln.bias, ln.eps)
    p = ops.conv2d(vln, P(cn.m[2]), bias=cn.m[2].bias)
    gate = None
    if cn.drop_prob and cn.training:                      # per-sample stochastic depth on the residual (networks.py:370-385)
        keep = 1 - cn.drop_prob
        gate = torch.floor(keep + torch.rand(B, 1, dtype=u.dtype, device=u.device)) / keep
        res = ops.scale_channels(u, gate.expand(B, u.shape[1]).contiguous())
    else:
        res = u
    out = ops.gelu_add(p, res)
    return out, (cn, x, u, v, vln, p, mean.float(), invstd.float(), gate)


def _convnext_backward(tape, g_out, want_input_grad):
    cn, x, u, v, vln, p, mean, invstd, gate = tape
    ln = cn.m[1]
    g_p = ops.gelu_bwd(g_out, p)
    _conv_param_grads(cn.m[2], vln, g_p)
    g_vln = ops.conv2d(g_p, _packT(cn.m[2]))
    for t in (ln.weight, ln.bias):
        if t.grad is None:
            t.grad = torch.zeros_like(t)
    g_v = ops.layernorm_bwd(g_vln, v, ln.weight, mean, invstd, ln.weight.grad, ln.bias.grad)
    _conv_param_grads(cn.m[0], u, g_v)
    g_u = ops.conv2d(g_v, _packT(cn.m[0]))
    g_res = g_out if gate is None else ops.scale_channels(g_out, gate.expand(g_out.shape[0], g_out.shape[1]).contiguous())
    g_u = ops.axpby(g_u, 1.0, g_res, 1.0)
    _conv_param_grads(cn.input, x, g_u)
    return ops.conv2d(g_u, _packT(cn.input)) if want_input_grad else None


def lrnn_forward_train(lrnn, views, mean_vol=None):
    """``LRNN.forward`` (networks.py:544-555) in training mode with a tape: 1x1 conv + UNet on the views, and with
    ``mean_vol`` the mean-volume branch (two ConvNeXt blocks + global attention, combined as x + 2 m (att - 0.5))."""
    c0 = lrnn.deconv[0]
    x0 = ops.conv2d(views, lrnn._packed.get(c0), bias=c0.bias)
    x, utape = unet_forward_train(lrnn.deconv[1], x0)
    if mean_vol is None:
        return x, (lrnn, views, utape, None)
    m1, t1 = _convnext_forward_train(lrnn.conv3d[0], mean_vol)
    m, t2 = _convnext_forward_train(lrnn.conv3d[1], m1)
    out = lrnn.attention_3d.combine(mean_vol, m, x)
    return out, (lrnn, views, utape, (mean_vol, m, t1, t2))


def lrnn_backward(tape, g_out):
    """Backward of ``lrnn_forward_train``: accumulates the .grad of every LRNN parameter the loss reaches."""
    lrnn, views, utape, mtape = tape
    if mtape is not None:
        mean_vol, m, t1, t2 = mtape
        att = lrnn.attention_3d.m
        g_m, pg = ops.attention_bwd(mean_vol, att[0].weight, att[0].bias, att[2].weight, att[2].bias, m, g_out)
        Cc = mean_vol.shape[1]
        n1, n2 = Cc * Cc * 3, Cc * Cc
        _acc(att[0].weight, pg[:n1])
        _acc(att[0].bias, pg[n1:n1 + Cc])
        _acc(att[2].weight, pg[n1 + Cc:n1 + Cc + n2])
        _acc(att[2].bias, pg[n1 + Cc + n2:])
        g_m1 = _convnext_backward(t2, g_m, True)
        _convnext_backward(t1, g_m1, False)
    g_x0 = unet_backward(utape, g_out)                              # dL/dx = g (out = x + ...)
    _conv_param_grads(lrnn.deconv[0], views, g_x0)


def lrnn_step_backward(encoder, views, mean_vol, gt, loss_func="L2", group=None):
    """Training step of the LAST pyramid step (`is_last_step`, CWFA.py:880-886,936-950): upsampled_vol = LRNN(views, mean_vol),
    loss = F.mse_loss / F.l1_loss(curr_gt, upsampled_vol) (main.py:42: L2), backward into every LRNN parameter.
    ``encoder`` is the ``Encoder`` (or its ``.net``).  The loss is the mean over the GLOBAL batch when torch.distributed is
    initialised (gradients are local contributions: ``allreduce_gradients`` sums them).  Returns (loss, upsampled_vol)."""
    from .CWFA import allreduce_nll
    lrnn = getattr(encoder, "net", encoder)
    if loss_func not in ("L1", "L2"):
        raise NotImplementedError(f"lrnn_step_backward: loss_func {loss_func!r} (L1 and L2 are built)")
    out, tape = lrnn_forward_train(lrnn, views, mean_vol)
    cnt = allreduce_nll(torch.tensor([float(out.numel())], dtype=torch.float64, device=out.device), group)
    numel = float(cnt[0])
    diff = ops.axpby(out, 1.0, gt, -1.0)
    if loss_func == "L2":
        g = ops.axpby(diff, 2.0 / numel)
        lsum = ops.sample_stats(diff.reshape(1, -1, 1, 1))[1:2]
    else:
        g = torch.sign(diff) / numel                      # tiny elementwise glue on the loss gradient
        lsum = diff.abs().sum(dtype=torch.float64).reshape(1)
    loss = allreduce_nll(lsum.clone(), group)[0] / numel
    lrnn_backward(tape, g)
    return loss, out


def train_iteration_autograd(conv_inn, cond_nets, gt_volume, cond_input, mean_vols_cache, cond_weight=0.40984, optimizers=None):
    """The same iteration written the way the reference writes it (CWFA.py:865-1027): forward through the modules, the loss with
    torch operators, ``full_loss.backward()`` -- i.e. what a user of ``cwfa_amd.install()`` runs without touching CWFA.py; the
    modules are autograd nodes (cwfa_amd/autograd.py).  z = 0 (the reference's default temperature), L2 losses, the nets in the
    modes the caller left them in.  ``optimizers``: optional list (index = pyramid step) of torch optimisers stepped after each
    backward.  Returns {"losses": [...], "volume": finest reconstruction}."""
    import torch.nn.functional as F
    from . import CWFA
    S = len(conv_inn) + 1
    gt_cache = [gt_volume]
    with torch.no_grad():
        for _ in range(S - 1):
            y = ops.haar1d(gt_cache[-1], False)
            gt_cache.append(y[:, :y.shape[1] // 2].contiguous())
    for m in list(conv_inn) + list(cond_nets):
        for p in m.parameters():
            p.grad = None

    def step_opt(n):
        opt = None if optimizers is None else optimizers[n]
        for o in (opt if isinstance(opt, (tuple, list)) else (opt,)):
            if o is not None:
                o.step()
                o.zero_grad(set_to_none=True)

    losses = [None] * S
    with torch.enable_grad():
        up = cond_nets[S - 1](cond_input, mean_vols_cache[S - 2])[-1]                # CWFA.py:880-886
        loss = F.mse_loss(gt_cache[S - 1], up)                                       # CWFA.py:936-950
        loss.backward()
        losses[S - 1] = loss.detach()
        step_opt(S - 1)
        up = up.detach()                                                             # CWFA.py:1015
        for n in range(S - 2, -1, -1):
            g = conv_inn[n]
            cond = [cond_nets[n](cond_input)[-1].float(), mean_vols_cache[n]]        # CWFA.py:893-899
            z = CWFA.sample_z_truncated((up.shape[0],) + tuple(g.global_out_shapes[0]), device=up.device, temperature=0)
            xhat, _ = g([z, up], c=cond, rev=True)                                   # CWFA.py:911
            full = F.mse_loss(gt_cache[n], xhat) * cond_weight                       # CWFA.py:952-959
            Z, ld = g(gt_cache[n], c=cond)                                           # CWFA.py:966
            full = full + (0.5 * torch.norm(Z[0]) ** 2 - ld.mean()) / xhat.numel() * (1 - cond_weight)      # CWFA.py:970-987
            full.backward()                                                          # CWFA.py:1002-1006
            losses[n] = full.detach()
            step_opt(n)
            up = xhat.detach()
    return {"losses": losses, "volume": up}


def train_iteration(conv_inn, cond_nets, gt_volume, cond_input, mean_vols_cache, optimizers=None, lr=None, cond_weight=0.40984,
                    loss_func_reg="L2", loss_func_first_step="L2", z_sampler=None, use_mean_branch=True, group=None,
                    views_noise_std=0.0, cond_dropout=False):
    """One training iteration over the whole pyramid for one batch, in the order of the reference's loop (CWFA.py:865-1027):
    the last step first -- LRNN on the views (+ mean-volume branch), L2 loss against the coarsest level of the ground-truth
    pyramid -- then every flow step from coarse to fine: condition net, inverse pass from the (detached) previous
    reconstruction, the weighted reconstruction + NLL loss, backward, gradient exchange, optimiser step, detach
    (CWFA.py:1015).  ``conv_inn`` / ``cond_nets`` as ``CWFA.build_networks`` returns them (cond_nets[-1] = the LRNN Encoder).

    ``optimizers``: None (plain gradient steps with ``lr``, or no update at all if ``lr`` is None), or one entry per pyramid
    step n = 0..S-1: a torch optimiser over that step's parameters, or a pair (flow optimiser, condition-net optimiser)
    as the reference keeps them (``optimizer`` / ``optimizer_cond``).  ``z_sampler(shape) -> tensor`` draws the latent of the
    inverse pass (None: z = 0, the reference's default temperature, main.py:109).
    The reference's two regularisers of this loop, both OFF by default here so that a call is deterministic (they draw from
    torch's RNG): ``views_noise_std`` -- N(0, std) noise added to the views of the LRNN step (CWFA.py:881 uses std 0.5 when
    ``--add_noise 1``, main.py:46's default); ``cond_dropout`` -- the optimised step's condition net in train mode, i.e.
    Dropout3d(0.5) on its hidden Conv3d channels (CWFA.py:768,859; networks.py:224), see ``cond_forward_train``.
    Returns {"losses": per-step full_loss (index = pyramid step), "nll": ..., "recon": ..., "volume": finest reconstruction}."""
    S = len(conv_inn) + 1
    gt_cache = [gt_volume]
    for _ in range(S - 1):                                   # the forward pyramid keeps the low band (CWFA.py:146-195)
        y = ops.haar1d(gt_cache[-1], False)
        gt_cache.append(y[:, :y.shape[1] // 2].contiguous())

    def params_of(mods):
        return [p for m in mods for p in m.parameters() if p.requires_grad]

    def update(n, mods):
        ps = params_of(mods)
        allreduce_gradients(ps, group)
        opt = None if optimizers is None else optimizers[n]
        if opt is None:
            if lr is not None:
                sgd_step(ps, lr)
        else:
            for o in (opt if isinstance(opt, (tuple, list)) else (opt,)):
                if o is not None:
                    o.step()
        for p in ps:
            p.grad = None

    losses, nlls, recons = [None] * S, [None] * S, [None] * S
    enc = cond_nets[S - 1]
    for p in params_of([enc]):
        p.grad = None
    views_lrnn = cond_input
    if views_noise_std:                                      # CWFA.py:881: torch.normal(0, 0.5, cond_input.size())
        views_lrnn = ops.axpby(cond_input, 1.0, torch.randn_like(cond_input), float(views_noise_std))
    loss, up = lrnn_step_backward(enc, views_lrnn, mean_vols_cache[S - 2] if use_mean_branch else None, gt_cache[S - 1],
                                  loss_func=loss_func_first_step, group=group)
    losses[S - 1] = recons[S - 1] = loss
    update(S - 1, [enc])
    for n in range(S - 2, -1, -1):
        g, cn = conv_inn[n], cond_nets[n]
        for p in params_of([g, cn]):
            p.grad = None
        was_training = cn.training
        if cond_dropout:
            cn.train()                                       # CWFA.py:859
        try:
            omega, ctape = cond_forward_train(cn, cond_input)
        finally:
            cn.train(was_training)
        z = None if z_sampler is None else z_sampler((up.shape[0],) + tuple(g.global_out_shapes[0]))
        out = step_backward(g, gt_cache[n], [omega, mean_vols_cache[n]], low=up, z=z, cond_weight=cond_weight,
                            loss_func=loss_func_reg, group=group, want_cond_grads=True)
        cond_backward(ctape, out["cond_grads"][0])
        losses[n], nlls[n], recons[n] = out["full_loss"], out["nll"], out["recon"]
        update(n, [g, cn])
        up = out["xhat"]                                     # `upsampled_vol.detach()`: nothing here records a graph
    return {"losses": losses, "nll": nlls, "recon": recons, "volume": up}


def allreduce_gradients(params: Sequence[torch.nn.Parameter], group=None, bucket_bytes: int = 64 << 20):
    """SUM the ranks' gradients (the loss is defined on the global batch, see the module docstring).  Gradients are
    packed into flat buckets of at most ``bucket_bytes`` in parameter order, one all-reduce per bucket, and copied back;
    every rank must pass the same parameter list.  No-op without an initialised process group or with one rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return 0
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    # which parameters did ANY rank's loss reach?  (one tiny message; the others keep .grad = None, as under autograd)
    has = torch.tensor([float(p.grad is not None) for p in params], dtype=torch.float32, device=params[0].device)
    dist.all_reduce(has, op=dist.ReduceOp.MAX, group=group)
    params = [p for p, h in zip(params, has.tolist()) if h > 0]
    n_buckets, i = 0, 0
    while i < len(params):
        j, size = i, 0
        while j < len(params) and (j == i or size + params[j].numel() * 4 <= bucket_bytes):
            size += params[j].numel() * 4
            j += 1
        chunk = params[i:j]
        flat = torch.zeros(size // 4, dtype=torch.float32, device=chunk[0].device)
        off = 0
        for p in chunk:                                  # a parameter without a local gradient contributes zeros
            if p.grad is not None:
                flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
            off += p.numel()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for p in chunk:
            g = flat[off:off + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += p.numel()
        n_buckets += 1
        i = j
    return n_buckets


def sgd_step(params, lr):
    """Plain gradient step (enough to show the loss going down in tests; the reference trains with torch optimisers,
    which work on these parameters / .grad tensors unchanged)."""
    with torch.no_grad():
        for p in params:
            if p.grad is not None:
                p.add_(p.grad, alpha=-lr)
