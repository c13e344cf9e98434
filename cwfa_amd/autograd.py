"""torch.autograd integration of the HIP path: ``full_loss.backward()`` (CWFA.py:1002-1006) on cwfa_amd modules.

The reference trains by letting torch autograd record every stock operator of a forward call.  The modules here run HIP kernels
torch knows nothing about, so every differentiable piece is a ``torch.autograd.Function`` whose backward launches the matching
backward kernels (``cwfa_amd.training`` holds the manual forward-with-tape / backward pairs they are built on):

  module level   the LRNN / Encoder (UNet + mean-volume branch), a condition net (ResidualBlock: 2-D convolutions + the fused
                 Conv3d 1 -> K -> 1), one coupling sub-network (1x1, three residual layers, 3x3);
  operator level the fused step chain in both directions (Haar1D + Split + gathers + affines: ``chain_fwd`` / ``chain_inv``, whose
                 backward recomputes every stage input by inverting the stage -- no stored activations), one affine coupling
                 stage (the data-dependent blocks: GLOW / RNVP / GIN / NICE / one-sided / AllInOne), gathers, the depth Haar
                 transform, channel concatenation, per-channel affines (ActNorm, AllInOne's global affine).

A module's ``forward`` takes these paths only when torch would record a graph (``tracking``): grad mode on and a parameter or an
input requires grad.  Under ``torch.no_grad()`` -- inference, the benchmark -- nothing changes.  Parameter gradients are returned
to autograd as the gradients of the Function's parameter inputs, so ``.grad`` accumulation, ``GradScaler`` and every
``torch.optim`` optimiser work unchanged.  Double backward is not supported (the tapes are released by the backward pass).
"""
import math
import weakref

import torch
from torch.autograd import Function

from . import ops

__all__ = ["tracking", "subnet", "chain_fwd", "chain_inv", "affine", "gather", "concat", "haar1d", "channel_affine",
           "cond_net", "lrnn", "ENABLED"]

ENABLED = True          # False: the modules never take the autograd paths (forward results are plain tensors)


def tracking(*objs):
    """Would torch record a graph for a call with these tensors / modules / parameter lists?"""
    if not (ENABLED and torch.is_grad_enabled()):
        return False
    for o in objs:
        if o is None:
            continue
        if torch.is_tensor(o):
            if o.requires_grad:
                return True
        elif isinstance(o, torch.nn.Module):
            if any(p.requires_grad for p in o.parameters()):
                return True
        elif isinstance(o, (list, tuple)):
            if tracking(*o):
                return True
    return False


class _capture:
    """Run a manual backward routine (which ACCUMULATES into ``p.grad``) and hand what it produced to autograd instead: the
    parameters' existing ``.grad`` are set aside for the duration and restored afterwards."""

    def __init__(self, params):
        self.params = params

    def __enter__(self):
        self.saved = [p.grad for p in self.params]
        for p in self.params:
            p.grad = None
        return self

    def __exit__(self, *exc):
        self.grads = [p.grad for p in self.params]
        for p, g in zip(self.params, self.saved):
            p.grad = g
        return False


def _params_of(mods):
    seen, out = set(), []
    for m in mods:
        for p in (m.parameters() if isinstance(m, torch.nn.Module) else [m]):
            if p is not None and p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out


def _f32c(g):
    """A gradient tensor as the kernels want it: fp32, contiguous (autograd may hand over expanded or strided views)."""
    return g.to(torch.float32).contiguous()


# ------------------------------------------------------------------------------------------------ module level
class _SubnetFn(Function):
    @staticmethod
    def forward(ctx, net, conv_in, conv_out, u, *params):
        from . import training
        a, tape = training.subnet_forward_train(net, u, conv_in, conv_out)
        ctx.tape, ctx.params, ctx.net = tape, params, net
        ctx.set_materialize_grads(False)
        return a

    @staticmethod
    def backward(ctx, g):
        from . import training
        n = len(ctx.params)
        _forget(ctx.net)                         # the node's tape is spent: a later call must build a new node
        if g is None:
            return (None,) * (4 + n)
        with _capture(ctx.params) as cap:
            gu = training.subnet_backward(ctx.tape, _f32c(g).clone(), want_input_grad=ctx.needs_input_grad[3])
        ctx.tape = None
        return (None, None, None, gu, *cap.grads)


REUSE_SUBNET_NODES = True     # see subnet()


_subnet_memo = weakref.WeakKeyDictionary()      # sub-network module -> (key, weakref of its output, weakref of its input); NOT an attribute of
#                                                 the module: weak references would make `torch.save(module)` / pickle fail


def _forget(net):
    _subnet_memo.pop(net, None)


def subnet(net, u, conv_in, conv_out):
    """``wavelet_flow_subnetwork._stack`` (networks.py:641-667) as one autograd node: 1x1 -> 3 residual layers -> 3x3.

    The reference's training step evaluates every sub-network of a conditional-affine step TWICE on the same tensors -- in the
    inverse pass of the reconstruction term (CWFA.py:911) and in the forward pass of the NLL term (CWFA.py:966) -- and lets
    autograd add the two gradient contributions.  A sub-network is a deterministic function of (input, parameters), so the second
    call returns the FIRST call's output tensor: one forward, and -- the node being shared -- one backward with the summed gradient.
    The memo (a WeakKeyDictionary over the modules) is keyed by the input tensor object / version and every parameter's version, holds its tensors
    weakly and is dropped as soon as the node's backward has run."""
    convs = [conv_in, net.block2[0], net.block2[2], net.block4[0], net.block4[2], net.block6[0], net.block6[2], conv_out]
    params = _params_of([c.weight for c in convs] + [c.bias for c in convs])
    key = None
    if REUSE_SUBNET_NODES:
        key = (id(u), u._version, id(conv_in), id(conv_out), ops.pack_epoch(), ops._split_bf16) + tuple((id(p), p._version) for p in params)
        hit = _subnet_memo.get(net)
        if hit is not None and hit[0] == key:
            out, src = hit[1](), hit[2]()
            if out is not None and src is u:
                return out
    out = _SubnetFn.apply(net, conv_in, conv_out, u, *params)
    if key is not None:
        _subnet_memo[net] = (key, weakref.ref(out), weakref.ref(u))
    return out


class _CondFn(Function):
    @staticmethod
    def forward(ctx, cond_net, views, *params):
        from . import training
        omega, tape = training.cond_forward_train(cond_net, views)
        ctx.tape, ctx.params = tape, params
        ctx.set_materialize_grads(False)
        return omega

    @staticmethod
    def backward(ctx, g):
        from . import training
        n = len(ctx.params)
        if g is None:
            return (None,) * (2 + n)
        with _capture(ctx.params) as cap:
            training.cond_backward(ctx.tape, _f32c(g))
        ctx.tape = None
        return (None, None, *cap.grads)


def cond_net(net, views):
    """A condition net (networks.py:165-242) as one autograd node; ``net``: the cond_network or its ResidualBlock.  The views
    get no gradient (the reference's do not require one)."""
    return _CondFn.apply(net, views, *_params_of([net]))


class _LrnnFn(Function):
    @staticmethod
    def forward(ctx, net, views, mean_vol, *params):
        from . import training
        out, tape = training.lrnn_forward_train(net, views, mean_vol)
        ctx.tape, ctx.params = tape, params
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import training
        n = len(ctx.params)
        if g is None:
            return (None,) * (3 + n)
        with _capture(ctx.params) as cap:
            training.lrnn_backward(ctx.tape, _f32c(g))
        ctx.tape = None
        return (None, None, None, *cap.grads)


def lrnn(net, views, mean_vol=None):
    """The LRNN (networks.py:505-555: 1x1 + UNet, + the mean-volume branch) as one autograd node."""
    return _LrnnFn.apply(net, views, mean_vol, *_params_of([net]))


# ------------------------------------------------------------------------------------------------ the fused step chain
def _slots(stages):
    """Indices of the stages that carry coefficient tensors (coupling blocks), in order."""
    return [k for k, (st, _keep) in enumerate(stages) if st.s_raw or st.t]


class _ChainFwdFn(Function):
    @staticmethod
    def forward(ctx, x, stages, final_perm, tabs, *coef):             # coef = (s_raw, t) per coefficient stage, flat
        logdet = torch.zeros(x.shape[0], dtype=torch.float64, device=x.device)
        z, low = ops.chain_fwd(x, stages, final_perm, logdet=logdet, tables=tabs)
        ctx.stages, ctx.final_perm = stages, final_perm
        ctx.save_for_backward(z, *[t for t in coef if t is not None])  # the stage structs point into these tensors' storage
        ctx.coef_shape = [None if t is None else tuple(t.shape) for t in coef]
        ctx.set_materialize_grads(False)
        return z, low, logdet.to(torch.float32)

    @staticmethod
    def backward(ctx, gz, glow, gld):
        z = ctx.saved_tensors[0]
        stages, n = ctx.stages, len(ctx.coef_shape)
        B, Cc, H, W = z.shape
        grads, flat = [], [None] * n
        it = iter(range(0, n, 2))
        for k, (st, _keep) in enumerate(stages):
            if not (st.s_raw or st.t):
                grads.append((None, None))
                continue
            i = next(it)
            need_s = st.s_raw and ctx.needs_input_grad[4 + i]
            need_t = st.t and ctx.needs_input_grad[4 + i + 1]
            flat[i] = torch.empty((B, Cc, H, W), dtype=torch.float32, device=z.device) if need_s else None
            flat[i + 1] = torch.empty((B, Cc, H, W), dtype=torch.float32, device=z.device) if need_t else None
            grads.append((flat[i], flat[i + 1]))
        need_x = ctx.needs_input_grad[0]
        gv0 = ops.chain_bwd(z, stages, grads, ctx.final_perm, gz=None if gz is None else _f32c(gz), want_input_grad=need_x,
                            gld=None if gld is None else _f32c(gld))
        gx = None
        if need_x:      # x -> Haar1D -> (low, detail): the transform is orthonormal, its transpose is its inverse
            lo = _f32c(glow) if glow is not None else torch.zeros_like(gv0)
            gx = ops.haar1d(None, rev=True, lo=lo, hi=gv0)
        return (gx, None, None, None, *flat)


def chain_fwd(x, stages, final_perm, tabs, coef):
    """``ops.chain_fwd`` as an autograd node.  ``coef``: per coefficient stage (in stage order) the (s_raw, t) tensors the stage
    structs were built from.  Returns (z, low, logdet[B])."""
    flat = [t for pair in coef for t in pair]
    return _ChainFwdFn.apply(x, stages, final_perm, tabs, *flat)


class _ChainInvFn(Function):
    @staticmethod
    def forward(ctx, z, low, rstages, rtabs, fstages, final_perm, *coef):     # coef in FORWARD stage order
        logdet = torch.zeros(low.shape[0], dtype=torch.float64, device=low.device)
        xhat = ops.chain_inv(z, low, rstages, logdet=logdet, tables=rtabs)
        ctx.fstages, ctx.final_perm, ctx.has_z = fstages, final_perm, z is not None
        ctx.save_for_backward(xhat, *[t for t in coef if t is not None])
        ctx.n = len(coef)
        ctx.set_materialize_grads(False)
        return xhat, logdet.to(torch.float32)

    @staticmethod
    def backward(ctx, gx, gld):
        n = ctx.n
        if gld is not None:
            raise NotImplementedError("cwfa_amd: a loss on the log-det of the INVERSE pass has no backward (the reference "
                                      "discards it, CWFA.py:912)")
        if gx is None:
            return (None,) * (6 + n)
        xhat = ctx.saved_tensors[0]
        B, D, H, W = xhat.shape
        Cc = D // 2
        grads, flat = [], [None] * n
        it = iter(range(0, n, 2))
        for st, _keep in ctx.fstages:
            if not (st.s_raw or st.t):
                grads.append((None, None))
                continue
            i = next(it)
            need_s = st.s_raw and ctx.needs_input_grad[6 + i]
            need_t = st.t and ctx.needs_input_grad[6 + i + 1]
            flat[i] = torch.empty((B, Cc, H, W), dtype=torch.float32, device=xhat.device) if need_s else None
            flat[i + 1] = torch.empty((B, Cc, H, W), dtype=torch.float32, device=xhat.device) if need_t else None
            grads.append((flat[i], flat[i + 1]))
        need_z, need_low = ctx.has_z and ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        out = ops.chain_inv_bwd(xhat, _f32c(gx), ctx.fstages, grads, gscale=1.0, loss_kind=0, want_latent_grad=need_z,
                                want_low_grad=need_low)
        gz = glow = None
        if need_z or need_low:
            _, gz, glow = out
            if gz is not None and ctx.final_perm is not None:      # z = gather(v_n, final_perm): dL/dz[k] = dL/dv_n[perm[k]]
                gz = ops.gather(gz, ctx.final_perm, 1)
        return (gz, glow, None, None, None, None, *flat)


def chain_inv(z, low, rstages, rtabs, fstages, final_perm, coef):
    """``ops.chain_inv`` as an autograd node.  ``fstages`` / ``final_perm`` / ``coef``: the FORWARD-direction chain over the same
    coefficient tensors (what the backward kernel walks).  Returns (xhat, logdet[B])."""
    flat = [t for pair in coef for t in pair]
    return _ChainInvFn.apply(z, low, rstages, rtabs, fstages, final_perm, *flat)


# ------------------------------------------------------------------------------------------------ single operators
class _AffineFn(Function):
    @staticmethod
    def forward(ctx, x, s_raw, t, kw, rev):
        st = ops.stage(s_raw, t, **kw)
        logdet = torch.zeros(x.shape[0], dtype=torch.float64, device=x.device)
        y = ops.affine(x, st, rev, logdet=None if kw.get("gin") else logdet)
        ctx.kw, ctx.rev = kw, rev
        ctx.save_for_backward(x, s_raw, t)
        ctx.set_materialize_grads(False)
        return y, logdet.to(torch.float32)

    @staticmethod
    def backward(ctx, g, gld):
        x, s_raw, t = ctx.saved_tensors
        if g is None and gld is None:
            return None, None, None, None, None
        if g is None:
            g = torch.zeros_like(x)
        st = ops.stage(s_raw, t, **ctx.kw)
        gx, gs, gt = ops.affine_bwd(x, _f32c(g), st, ctx.rev, gld=None if gld is None else _f32c(gld), want=ctx.needs_input_grad[:3])
        return gx, gs, gt, None, None


def affine(x, s_raw, t, rev, clamp_kind="ATAN", clamp=2.0, pre_scale=1.0, t_neg_div_sqrt2=False, gin=False):
    """One affine coupling stage (coupling_layers.py:50-60) as an autograd node: returns (y, logdet[B])."""
    kw = dict(clamp_kind=clamp_kind, clamp=clamp, pre_scale=pre_scale, t_neg_div_sqrt2=t_neg_div_sqrt2, gin=gin)
    return _AffineFn.apply(x, s_raw, t, kw, bool(rev))


class _GatherFn(Function):
    @staticmethod
    def forward(ctx, x, perm, perm_inv, axis):
        ctx.perm_inv, ctx.axis = perm_inv, axis
        return ops.gather(x, perm, axis)

    @staticmethod
    def backward(ctx, g):
        return ops.gather(_f32c(g), ctx.perm_inv, ctx.axis), None, None, None


def gather(x, perm, axis, perm_inv=None):
    """y = x.index_select(axis, perm) as an autograd node (``perm_inv``: the inverse table, computed when not given)."""
    if perm_inv is None:
        perm_inv = torch.argsort(perm)
    return _GatherFn.apply(x, perm.detach(), perm_inv.detach(), int(axis))


class _ConcatFn(Function):
    @staticmethod
    def forward(ctx, *parts):
        ctx.sizes = [p.shape[1] for p in parts]
        return ops.concat_channels(list(parts))

    @staticmethod
    def backward(ctx, g):
        out, c0 = [], 0
        for i, n in enumerate(ctx.sizes):
            out.append(g[:, c0:c0 + n].contiguous() if ctx.needs_input_grad[i] else None)
            c0 += n
        return tuple(out)


_concat_memo = []             # [(key, weakrefs of the parts, weakref of the result)], a handful of entries


def concat(parts):
    """torch.cat(parts, 1) through the strided plane-copy kernel, as an autograd node.  The same tensor OBJECTS (same versions)
    concatenated again give the same result tensor while it is alive, so that sub-network nodes fed by it can be reused
    (``subnet``): the condition list of a step is concatenated once for the inverse and the forward pass."""
    parts = list(parts)
    if len(parts) == 1:
        return parts[0]
    key = tuple((id(t), t._version) for t in parts)
    if REUSE_SUBNET_NODES:
        for k, refs, res in _concat_memo:
            if k == key:
                out = res()
                if out is not None and all(r() is t for r, t in zip(refs, parts)):
                    return out
    out = _ConcatFn.apply(*parts)
    if REUSE_SUBNET_NODES:
        _concat_memo.append((key, [weakref.ref(t) for t in parts], weakref.ref(out)))
        del _concat_memo[:-8]
    return out


class _Haar1dFn(Function):
    @staticmethod
    def forward(ctx, x, rev):
        ctx.rev = rev
        return ops.haar1d(x, rev)

    @staticmethod
    def backward(ctx, g):
        return ops.haar1d(_f32c(g), not ctx.rev), None          # orthonormal: the transpose is the inverse


def haar1d(x, rev):
    return _Haar1dFn.apply(x, bool(rev))


class _ChannelAffineFn(Function):
    @staticmethod
    def forward(ctx, x, scale, shift, inverse):
        y = ops.channel_affine(x, scale.reshape(-1).contiguous(), shift.reshape(-1).contiguous(), inverse=inverse)
        ctx.inverse = inverse
        ctx.save_for_backward(y if inverse else x, scale, shift)
        return y

    @staticmethod
    def backward(ctx, g):
        v, scale, shift = ctx.saved_tensors                       # v = x (forward form) or y (inverse form)
        g = _f32c(g)
        sc = scale.reshape(-1).contiguous()
        zero = torch.zeros_like(sc)
        st = ops.bn_bwd_stats(g, v).to(torch.float32)             # [C, 2]: (sum g, sum g * v) over (B,H,W)
        if not ctx.inverse:                                       # y = x * scale + shift
            gx = ops.channel_affine(g, sc, zero, inverse=False)
            gscale, gshift = st[:, 1], st[:, 0]
        else:                                                     # y = (x - shift) / scale
            gx = ops.channel_affine(g, sc, zero, inverse=True)
            gscale, gshift = -st[:, 1] / sc, -st[:, 0] / sc
        return gx, gscale.reshape(scale.shape), gshift.reshape(shift.shape), None


def channel_affine(x, scale, shift, inverse=False):
    """y = x * scale_c + shift_c (or its inverse) as an autograd node; ``scale`` / ``shift``: [C]-sized tensors that may carry
    their own (tiny, torch-recorded) graph -- ActNorm's exp(scale), AllInOne's 0.1 softplus(global_scale)."""
    return _ChannelAffineFn.apply(x, scale, shift, bool(inverse))


class _MixFn(Function):
    """y = conv1x1(x, w) for a dense C x C channel mix (AllInOneBlock's soft / Householder permutations, all_in_one_block.py:
    191-204): data gradient = the transposed mix, weight gradient on the MFMA weight-gradient kernel."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return ops.conv2d(x, ops.pack_conv_weight(w.detach().contiguous()))

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = _f32c(g)
        gx = ops.conv2d(g, ops.pack_conv_weight(w.detach().transpose(0, 1).contiguous())) if ctx.needs_input_grad[0] else None
        gw = ops.conv2d_wgrad(x, g, 1) if ctx.needs_input_grad[1] else None
        return gx, gw


def mix1x1(x, w):
    """x . w for a [C, C, 1, 1] mix ``w`` (which may carry its own torch graph: the Householder product), as an autograd node."""
    return _MixFn.apply(x, w)


class _ScaleSamplesFn(Function):
    """y[b] = x[b] * f[b] with a per-sample factor f [B] that carries a torch graph (AllInOneBlock's GIN mean, :218-219)."""

    @staticmethod
    def forward(ctx, x, f):
        B, Cc = x.shape[:2]
        ctx.save_for_backward(x, f)
        return ops.scale_channels(x.contiguous(), f.reshape(B, 1).expand(B, Cc).contiguous())

    @staticmethod
    def backward(ctx, g):
        x, f = ctx.saved_tensors
        B, Cc = x.shape[:2]
        g = _f32c(g)
        gx = ops.scale_channels(g, f.reshape(B, 1).expand(B, Cc).contiguous()) if ctx.needs_input_grad[0] else None
        gf = None
        if ctx.needs_input_grad[1]:           # sum_{c,h,w} g x per sample: the (sum g, sum g v) reduction with samples as "channels"
            st = ops.bn_bwd_stats(g.reshape(1, B, -1, 1), x.contiguous().reshape(1, B, -1, 1))
            gf = st[:, 1].to(torch.float32).reshape(f.shape)
        return gx, gf


def scale_samples(x, f):
    return _ScaleSamplesFn.apply(x, f)


class _SumPerSampleFn(Function):
    """sum over (C,H,W) of soft_clamp(pre * s_raw) per sample -- the quantity AllInOneBlock's GIN mode centres (:218-219)."""

    @staticmethod
    def forward(ctx, s_raw, kw):
        st = ops.stage(s_raw, None, **kw)
        B, Cc, H, W = s_raw.shape
        acc = torch.zeros(B, dtype=torch.float64, device=s_raw.device)
        ops.affine(None, st, False, shape=(B, Cc, H, W), logdet=acc)
        ctx.kw = kw
        ctx.save_for_backward(s_raw)
        return acc.to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        (s_raw,) = ctx.saved_tensors
        st = ops.stage(s_raw, None, **ctx.kw)
        zero = torch.zeros_like(s_raw)
        _, gs, _ = ops.affine_bwd(zero, zero, st, False, gld=_f32c(g), want=(False, True, False))   # dL/ds = gld_b for every element
        return gs, None


def clamped_sum(s_raw, clamp_kind, clamp, pre_scale):
    return _SumPerSampleFn.apply(s_raw, dict(clamp_kind=clamp_kind, clamp=clamp, pre_scale=pre_scale))


def neg_div_sqrt2(mean):
    """-mean / sqrt(2) (networks.py:671), recorded by torch itself: a rare path (the `_first` sub-network called as a module)."""
    return mean * (-1.0 / math.sqrt(2.0))
