"""Coupling blocks on the HIP affine-apply kernel.

Reference semantics: FrEIA/modules/coupling_layers.py (NICE :124-157, RNVP :160-229, GLOW :232-302, GIN :305-381,
AffineCouplingOneSided :384-437, ConditionalAffineTransform :440-500) and all_in_one_block.py:45-268.
Parameter names (``subnet``, ``subnet1``, ``subnet_s1`` ..., ``F``/``G``, ``global_scale`` ...) are the reference's so
that checkpoints load unchanged.

The sub-network is any ``nn.Module`` mapping a HIP tensor to a HIP tensor.  If it offers
``affine_parts(parts, n_s) -> (s_raw, t, t_neg_div_sqrt2)`` (the CWFA subnets do) the concatenations of the reference
(condition cat, s|t split, the ``_first`` pass-through cat) are never materialised.
"""
from typing import Callable, Union

import torch
import torch.nn as nn

from ... import autograd as AG
from ... import ops
from .base import InvertibleModule, as_jac, new_logdet

__all__ = ["NICECouplingBlock", "RNVPCouplingBlock", "GLOWCouplingBlock", "GINCouplingBlock",
           "AffineCouplingOneSided", "ConditionalAffineTransform", "AllInOneBlock"]


def subnet_st(net, parts, n_s):
    """Run a sub-network on the (virtual) channel concatenation of ``parts`` -> (s_raw, t, t_neg_div_sqrt2)."""
    if hasattr(net, "affine_parts"):
        return net.affine_parts(list(parts), n_s)
    u = parts[0] if len(parts) == 1 else (AG.concat(parts) if AG.tracking(list(parts)) else ops.concat_channels(parts))
    a = net(u)
    return a[:, :n_s], a[:, n_s:], False


class _BaseCouplingBlock(InvertibleModule):
    """Split sizes, condition bookkeeping and the soft clamp.  coupling_layers.py:8-60."""

    def __init__(self, dims_in, dims_c=[], clamp: float = 2., clamp_activation: Union[str, Callable] = "ATAN"):
        super().__init__(dims_in, dims_c)
        self.channels = dims_in[0][0]
        self.ndims = len(dims_in[0])
        self.split_len1 = self.channels // 2
        self.split_len2 = self.channels - self.channels // 2
        self.clamp = clamp
        assert all([tuple(dims_c[i][1:]) == tuple(dims_in[0][1:]) for i in range(len(dims_c))]), \
            "Dimensions of input and one or more conditions don't agree."
        self.conditional = (len(dims_c) > 0)
        self.condition_length = sum([dims_c[i][0] for i in range(len(dims_c))])
        if isinstance(clamp_activation, str):
            if clamp_activation not in ("ATAN", "TANH", "SIGMOID"):
                raise ValueError(f'Unknown clamp activation "{clamp_activation}"')
            self.clamp_kind = clamp_activation
        else:
            # a python callable cannot run inside the HIP kernel
            self.clamp_kind = None
            self.f_clamp = clamp_activation

    def _stage(self, s_raw, t, **kw):
        if self.clamp_kind is None:
            raise NotImplementedError("custom clamp_activation callables have no HIP implementation; use "
                                      "'ATAN', 'TANH' or 'SIGMOID'")
        return ops.stage(s_raw, t, self.clamp_kind, self.clamp, **kw)

    def _cond(self, first, c):
        return [first, *c] if self.conditional else [first]

    def output_dims(self, input_dims):
        if len(input_dims) != 1:
            raise ValueError("Can only use 1 input")
        return input_dims


class _TwoSided(_BaseCouplingBlock):
    """x=(x1|x2): fwd y1=A(x1|net2(x2,c)), y2=A(x2|net1(y1,c)); rev runs coupling2 first.  coupling_layers.py:62-87."""
    _gin = False

    def _nets(self, which, parts, n_out):
        raise NotImplementedError

    def _has_jac(self):
        return not self._gin

    def _forward_tracked(self, x0, c, rev):
        """The same data flow as ``forward`` with every piece an autograd node (cwfa_amd.autograd): sub-networks, affine stages,
        the channel concatenation of the two halves.  No in-place writes, nothing fused across the coupling."""
        l1 = self.split_len1
        x1, x2 = x0[:, :l1], x0[:, l1:]

        def couple(xa, parts, which, n_out):
            s_raw, t = self._nets(which, parts, n_out)
            if self.clamp_kind is None:
                raise NotImplementedError("custom clamp_activation callables have no HIP implementation")
            return AG.affine(xa, s_raw, t, rev, self.clamp_kind, self.clamp, gin=self._gin)

        if not rev:
            y1, j1 = couple(x1, self._cond(x2, c), 2, l1)
            y2, j2 = couple(x2, self._cond(y1, c), 1, self.split_len2)
        else:
            y2, j2 = couple(x2, self._cond(x1, c), 1, self.split_len2)
            y1, j1 = couple(x1, self._cond(y2, c), 2, l1)
        return (AG.concat([y1, y2]),), ((j1 + j2) if self._has_jac() else 0.)

    def forward(self, x, c=[], rev=False, jac=True):
        x0 = x[0]
        if AG.tracking(x0, list(c), self):
            return self._forward_tracked(x0, c, rev)
        l1 = self.split_len1
        x1, x2 = x0[:, :l1], x0[:, l1:]
        out = torch.empty_like(x0, memory_format=torch.contiguous_format)
        y1, y2 = out[:, :l1], out[:, l1:]
        acc = new_logdet(x0) if self._has_jac() else None
        if not rev:
            self._couple(x1, y1, self._cond(x2, c), 2, l1, rev, acc)
            self._couple(x2, y2, self._cond(y1, c), 1, self.split_len2, rev, acc)
        else:
            self._couple(x2, y2, self._cond(x1, c), 1, self.split_len2, rev, acc)
            self._couple(x1, y1, self._cond(y2, c), 2, l1, rev, acc)
        return (out,), (as_jac(acc) if acc is not None else 0.)

    def _couple(self, xa, ya, parts, which, n_out, rev, acc):
        net = self._fusable_net(which)
        if net is not None and net.couple(parts, xa, ya, self.clamp_kind, self.clamp, 1.0, rev, acc):
            return                      # s, t stayed in the accumulators of the sub-network's last convolution
        s_raw, t = self._nets(which, parts, n_out)
        ops.affine(xa, self._stage(s_raw, t, gin=self._gin), rev, logdet=acc, out=ya)

    def _fusable_net(self, which):
        """The sub-network that predicts [s | t] of coupling ``which`` if it can apply the coupling itself."""
        return None


class NICECouplingBlock(_TwoSided):
    """Additive coupling.  coupling_layers.py:124-157."""

    def __init__(self, dims_in, dims_c=[], subnet_constructor: callable = None):
        super().__init__(dims_in, dims_c, clamp=0., clamp_activation="ATAN")
        self.F = subnet_constructor(self.split_len2 + self.condition_length, self.split_len1)
        self.G = subnet_constructor(self.split_len1 + self.condition_length, self.split_len2)

    def _has_jac(self):
        return False

    def _nets(self, which, parts, n_out):
        net = self.F if which == 2 else self.G
        u = parts[0] if len(parts) == 1 else (AG.concat(parts) if AG.tracking(list(parts)) else ops.concat_channels(parts))
        return None, net(u)


class RNVPCouplingBlock(_TwoSided):
    """Four sub-networks (s and t separately).  coupling_layers.py:160-229."""

    def __init__(self, dims_in, dims_c=[], subnet_constructor: Callable = None, clamp: float = 2.,
                 clamp_activation: Union[str, Callable] = "ATAN"):
        super().__init__(dims_in, dims_c, clamp, clamp_activation)
        self.subnet_s1 = subnet_constructor(self.split_len1 + self.condition_length, self.split_len2)
        self.subnet_t1 = subnet_constructor(self.split_len1 + self.condition_length, self.split_len2)
        self.subnet_s2 = subnet_constructor(self.split_len2 + self.condition_length, self.split_len1)
        self.subnet_t2 = subnet_constructor(self.split_len2 + self.condition_length, self.split_len1)

    def _nets(self, which, parts, n_out):
        u = parts[0] if len(parts) == 1 else (AG.concat(parts) if AG.tracking(list(parts)) else ops.concat_channels(parts))
        return getattr(self, f"subnet_s{which}")(u), getattr(self, f"subnet_t{which}")(u)


class GLOWCouplingBlock(_TwoSided):
    """One sub-network predicts [s|t] per half.  coupling_layers.py:232-302."""

    def __init__(self, dims_in, dims_c=[], subnet_constructor: Callable = None, clamp: float = 2.,
                 clamp_activation: Union[str, Callable] = "ATAN"):
        super().__init__(dims_in, dims_c, clamp, clamp_activation)
        self.subnet1 = subnet_constructor(self.split_len1 + self.condition_length, self.split_len2 * 2)
        self.subnet2 = subnet_constructor(self.split_len2 + self.condition_length, self.split_len1 * 2)

    def _nets(self, which, parts, n_out):
        s_raw, t, _ = subnet_st(self.subnet1 if which == 1 else self.subnet2, parts, n_out)
        return s_raw, t

    def _fusable_net(self, which):
        net = self.subnet1 if which == 1 else self.subnet2
        return net if (not self._gin and self.clamp_kind is not None and hasattr(net, "couple")) else None


class GINCouplingBlock(GLOWCouplingBlock):
    """Volume preserving: the channel mean of s is removed at every pixel, log-det 0.  coupling_layers.py:305-381."""
    _gin = True


class AffineCouplingOneSided(_BaseCouplingBlock):
    """Only the second half is transformed.  coupling_layers.py:384-437."""

    def __init__(self, dims_in, dims_c=[], subnet_constructor: Callable = None, clamp: float = 2.,
                 clamp_activation: Union[str, Callable] = "ATAN"):
        super().__init__(dims_in, dims_c, clamp, clamp_activation)
        self.subnet = subnet_constructor(self.split_len1 + self.condition_length, 2 * self.split_len2)

    def forward(self, x, c=[], rev=False, jac=True):
        x0 = x[0]
        l1 = self.split_len1
        x1, x2 = x0[:, :l1], x0[:, l1:]
        if AG.tracking(x0, list(c), self):            # every piece an autograd node (cwfa_amd.autograd)
            s_raw, t, _ = subnet_st(self.subnet, self._cond(x1, c), self.split_len2)
            if self.clamp_kind is None:
                raise NotImplementedError("custom clamp_activation callables have no HIP implementation")
            y2, j = AG.affine(x2, s_raw, t, rev, self.clamp_kind, self.clamp)
            return (AG.concat([x1, y2]),), j
        out = torch.empty(x0.shape, dtype=torch.float32, device=x0.device)
        ops.copy_channels(x1, out[:, :l1])            # x1 passes through; the x2 half is written by the coupling below
        acc = new_logdet(x0)
        if not (self.clamp_kind is not None and hasattr(self.subnet, "couple") and
                self.subnet.couple(self._cond(x1, c), x2, out[:, l1:], self.clamp_kind, self.clamp, 1.0, rev, acc)):
            s_raw, t, _ = subnet_st(self.subnet, self._cond(x1, c), self.split_len2)
            ops.affine(x2, self._stage(s_raw, t), rev, logdet=acc, out=out[:, l1:])
        return (out,), as_jac(acc)


class ConditionalAffineTransform(_BaseCouplingBlock):
    """Affine transform of the WHOLE input with coefficients predicted from the conditions only.
    coupling_layers.py:440-500.  (This is CWFA's default block: every s,t of a step is independent of the data
    flowing through it, which is what lets GraphINN fuse a whole step into one chain kernel.)"""

    def __init__(self, dims_in, dims_c=[], subnet_constructor: Callable = None, clamp: float = 2.,
                 clamp_activation: Union[str, Callable] = "ATAN"):
        super().__init__(dims_in, dims_c, clamp, clamp_activation)
        if not self.conditional:
            raise ValueError("ConditionalAffineTransform must have a condition")
        self.subnet = subnet_constructor(self.condition_length, 2 * self.channels)

    def coefficients(self, c):
        """(s_raw, t, t_neg_div_sqrt2) from the conditions -- used by forward() and by GraphINN's fused plans."""
        return subnet_st(self.subnet, list(c), self.channels)

    def stage(self, c, perm=None, axis=1):
        s_raw, t, tneg = self.coefficients(c)
        return self._stage(s_raw, t, t_neg_div_sqrt2=tneg, perm=perm, axis=axis)

    def forward(self, x, c=[], rev=False, jac=True):
        if AG.tracking(x[0], list(c), self):
            s_raw, t, tneg = self.coefficients(c)
            if self.clamp_kind is None:
                raise NotImplementedError("custom clamp_activation callables have no HIP implementation")
            y, j = AG.affine(x[0], s_raw, t, rev, self.clamp_kind, self.clamp, t_neg_div_sqrt2=tneg)
            return (y,), j
        acc = new_logdet(x[0])
        y = ops.affine(x[0], self.stage(c), rev, logdet=acc)
        return (y,), as_jac(acc)


class AllInOneBlock(InvertibleModule):
    """GLOW-style block: one-sided affine coupling (tanh clamp, coefficients x0.1) + per-channel global affine
    (softplus) + fixed channel permutation.  all_in_one_block.py:45-268.

    Hard permutations (the default, what CWFA builds: networks.py:295,341-351) are index gathers fused with the global
    affine (bit-exact).  The other options of the reference run too: ``permute_soft`` (a random SO(C) matrix) and
    ``learned_householder_permutation`` are dense C x C mixes = 1x1 convolutions on the MFMA kernel;
    ``reverse_permutation`` pre-multiplies with the inverse mix; ``gin_block`` removes the per-sample mean of s over
    (C, H, W) (:218-219) and drops the global scale (:183-185)."""

    def __init__(self, dims_in, dims_c=[], subnet_constructor: Callable = None, affine_clamping: float = 2.,
                 gin_block: bool = False, global_affine_init: float = 1., global_affine_type: str = 'SOFTPLUS',
                 permute_soft: bool = False, learned_householder_permutation: int = 0,
                 reverse_permutation: bool = False):
        super().__init__(dims_in, dims_c)
        import numpy as np
        channels = dims_in[0][0]
        self.input_rank = len(dims_in[0]) - 1
        if self.input_rank != 2:
            raise ValueError("cwfa_amd AllInOneBlock handles image data [C,H,W] only")
        if len(dims_c) == 0:
            self.conditional, self.condition_channels = False, 0
        else:
            assert tuple(dims_c[0][1:]) == tuple(dims_in[0][1:]), \
                F"Dimensions of input and condition don't agree: {dims_c} vs {dims_in}."
            self.conditional, self.condition_channels = True, sum(dc[0] for dc in dims_c)
        self.splits = [channels - channels // 2, channels // 2]
        self.in_channels = channels
        self.clamp = affine_clamping
        self.GIN = bool(gin_block)
        self.reverse_pre_permute = bool(reverse_permutation)
        self.householder = int(learned_householder_permutation)
        self.global_affine_type = global_affine_type
        if permute_soft and channels > 512:
            import warnings
            warnings.warn(("Soft permutation will take a very long time to initialize "
                           f"with {channels} feature channels. Consider using hard permutation instead."))
        if global_affine_type == 'SIGMOID':
            global_scale = 2. - np.log(10. / global_affine_init - 1.)
        elif global_affine_type == 'SOFTPLUS':
            global_scale = 2. * np.log(np.exp(0.5 * 10. * global_affine_init) - 1)
        elif global_affine_type == 'EXP':
            global_scale = np.log(global_affine_init)
        else:
            raise ValueError('Global affine activation must be "SIGMOID", "SOFTPLUS" or "EXP"')
        self.global_scale = nn.Parameter(torch.ones(1, channels, 1, 1) * float(global_scale))
        self.global_offset = nn.Parameter(torch.zeros(1, channels, 1, 1))
        # the reference draws these from numpy's / scipy's global RNG (all_in_one_block.py:143-148)
        if permute_soft:
            from scipy.stats import special_ortho_group
            w = special_ortho_group.rvs(channels)
        else:
            w = np.zeros((channels, channels))
            for i, j in enumerate(np.random.permutation(channels)):
                w[i, j] = 1.
        if self.householder:
            self.vk_householder = nn.Parameter(0.2 * torch.randn(self.householder, channels), requires_grad=True)
            self.w_perm = None
            self.w_perm_inv = None
            self.w_0 = nn.Parameter(torch.FloatTensor(w), requires_grad=False)
        else:
            self.w_perm = nn.Parameter(torch.FloatTensor(w).view(channels, channels, 1, 1), requires_grad=False)
            self.w_perm_inv = nn.Parameter(torch.FloatTensor(w.T).view(channels, channels, 1, 1), requires_grad=False)
        if subnet_constructor is None:
            raise ValueError("Please supply a callable subnet_constructor function or object (see docstring)")
        self.subnet = subnet_constructor(self.splits[0] + self.condition_channels, 2 * self.splits[1])
        self.last_jac = None
        self._tables = None

    def _construct_householder_permutation(self):
        """w_0 . prod_k (I - 2 v_k v_k^T / v_k^T v_k), all_in_one_block.py:170-179 (a C x C host-side product)."""
        w = self.w_0
        for vk in self.vk_householder:
            w = torch.mm(w, torch.eye(self.in_channels, device=w.device) - 2 * torch.ger(vk, vk) / torch.dot(vk, vk))
        return w.view(self.in_channels, self.in_channels, 1, 1)

    # ---- tiny parameter-side tables ([C] vectors, C x C mixes; rebuilt when the parameters change)
    def _prepare(self):
        wp = self._construct_householder_permutation().detach() if self.householder else self.w_perm.detach()
        src = (self.vk_householder, self.w_0) if self.householder else (self.w_perm,)
        key = (tuple(t._version for t in src), tuple(t.data_ptr() for t in src), self.global_scale._version,
               self.global_offset._version, self.global_scale.data_ptr(), ops.pack_epoch())
        if self._tables is None or self._tables[0] != key:
            w = wp[:, :, 0, 0]
            hard = bool(((w == 0) | (w == 1)).all()) and bool((w.sum(0) == 1).all() and (w.sum(1) == 1).all())
            if hard:
                mix = (w.argmax(1).contiguous(), w.t().argmax(1).contiguous())   # fwd: out[:, i] = v[:, perm[i]]; rev: perm_inv
            else:      # dense mix: y = conv1x1(x, w) / conv1x1(x, w^T) (all_in_one_block.py:191-204)
                mix = (ops.pack_conv_weight(wp.contiguous()), ops.pack_conv_weight(wp.transpose(0, 1).contiguous()))
            g = self.global_scale.detach().reshape(-1)
            if self.GIN:
                scale = torch.ones_like(g)                                        # :183-185
            elif self.global_affine_type == 'SOFTPLUS':
                scale = 0.1 * torch.nn.functional.softplus(g, beta=0.5)
            elif self.global_affine_type == 'SIGMOID':
                scale = 10 * torch.sigmoid(g - 2.)
            else:
                scale = torch.exp(g)
            self._tables = (key, hard, mix, scale.contiguous(), self.global_offset.detach().reshape(-1).contiguous(),
                            torch.log(scale).sum().to(torch.float64))
        return self._tables[1:]

    def _mix(self, x, hard, mix, inverse):
        """x . w_perm (inverse=False) or x . w_perm_inv: a gather for hard permutations, a 1x1 conv otherwise."""
        if hard:
            return ops.gather(x, mix[1] if inverse else mix[0], 1)
        return ops.conv2d(x, mix[1] if inverse else mix[0])

    def _forward_tracked(self, x0, c, rev):
        """all_in_one_block.py:206-268 with every piece an autograd node (cwfa_amd.autograd).  The [C]-sized global-affine
        activation runs through torch's own graph; hard permutations are gathers; the coupling is one affine stage."""
        wp = self._construct_householder_permutation() if self.householder else self.w_perm      # (the product carries its graph)
        w = wp[:, :, 0, 0]
        hard = (not self.householder and bool(((w == 0) | (w == 1)).all()) and bool((w.sum(0) == 1).all() and (w.sum(1) == 1).all()))
        if hard:
            p_fwd, p_inv = w.argmax(1).contiguous(), w.t().argmax(1).contiguous()     # fwd: out[:, i] = v[:, p_fwd[i]]
            mix = lambda v, inverse: AG.gather(v, p_inv if inverse else p_fwd, 1, p_fwd if inverse else p_inv)   # noqa: E731
        else:          # dense mix: y = conv1x1(x, w) / conv1x1(x, w^T) (:191-204)
            mix = lambda v, inverse: AG.mix1x1(v, wp.transpose(0, 1) if inverse else wp)                          # noqa: E731
        g = self.global_scale.reshape(-1)
        if self.GIN:
            scale = torch.ones_like(g)                                                 # :183-185
        elif self.global_affine_type == 'SOFTPLUS':
            scale = 0.1 * torch.nn.functional.softplus(g, beta=0.5)
        elif self.global_affine_type == 'SIGMOID':
            scale = 10 * torch.sigmoid(g - 2.)
        else:
            scale = torch.exp(g)
        offset = self.global_offset.reshape(-1)
        l1, l2 = self.splits
        n_pix = x0.shape[2] * x0.shape[3]
        if rev:                                                                    # :191-193: undo permutation, then global affine
            v = AG.channel_affine(mix(x0, True), scale, offset, inverse=True)
        elif self.reverse_pre_permute:
            v = mix(x0, True)
        else:
            v = x0
        x1, x2 = v[:, :l1], v[:, l1:]
        parts = [x1, *c] if self.conditional else [x1]
        s_raw, t, _ = subnet_st(self.subnet, parts, l2)
        if self.GIN:
            # s <- s - mean_{C,H,W}(s) per sample (:218-219), log-det 0: the plain coupling and a per-sample factor exp(-+mean)
            m = AG.clamped_sum(s_raw, "TANH", self.clamp, 0.1) / float(l2 * n_pix)
            if rev:
                y2, _ = AG.affine(x2, s_raw, t, True, "TANH", self.clamp, pre_scale=0.1)
                y2 = AG.scale_samples(y2, torch.exp(m))
            else:
                y2, _ = AG.affine(AG.scale_samples(x2, torch.exp(-m)), s_raw, t, False, "TANH", self.clamp, pre_scale=0.1)
            j = torch.zeros(x0.shape[0], dtype=torch.float32, device=x0.device)
        else:
            y2, j = AG.affine(x2, s_raw, t, rev, "TANH", self.clamp, pre_scale=0.1)
        u = AG.concat([x1, y2])
        if rev:
            out = mix(u, False) if self.reverse_pre_permute else u
        else:
            out = mix(AG.channel_affine(u, scale, offset, inverse=False), False)
        j = j + (-1) ** int(rev) * n_pix * torch.log(scale).sum()
        return (out,), j

    def forward(self, x, c=[], rev=False, jac=True):
        if AG.tracking(x[0], list(c), self):
            return self._forward_tracked(x[0], c, rev)
        hard, mix, scale, offset, log_scale_sum = self._prepare()
        x0 = x[0]
        l1, l2 = self.splits
        B, _, H, W = x0.shape
        n_pix = H * W
        acc = new_logdet(x0)
        if rev:                                                                    # :191-193
            if hard:
                v = ops.channel_affine(x0, scale, offset, inverse=True, perm_in=mix[1])
            else:
                v = ops.channel_affine(self._mix(x0, hard, mix, True), scale, offset, inverse=True)
        elif self.reverse_pre_permute:                                             # :198-204
            v = self._mix(x0, hard, mix, True)
        else:
            v = x0
        x1, x2 = v[:, :l1], v[:, l1:]
        parts = [x1, *c] if self.conditional else [x1]
        if rev or self.reverse_pre_permute:
            u = v                                       # a tensor we own: its x2 half is overwritten in place
        else:
            u = torch.empty(x0.shape, dtype=torch.float32, device=x0.device)
            ops.copy_channels(x1, u[:, :l1])            # only the passive half is copied; the coupling writes the other one
        # the coupling from the accumulators of the sub-network's last convolution where that form applies (`a *= 0.1`, :213)
        if (not self.GIN and hasattr(self.subnet, "couple") and
                self.subnet.couple(parts, x2, u[:, l1:], "TANH", self.clamp, 0.1, rev, acc)):
            st = None
        else:
            s_raw, t, _ = subnet_st(self.subnet, parts, l2)
            st = ops.stage(s_raw, t, "TANH", self.clamp, pre_scale=0.1)
        if st is None:
            pass                                        # done: x2 half of u written, log-det accumulated
        elif self.GIN:
            # s <- s - mean_{C,H,W}(s) per sample (:218-219): sum s with one reduction pass (affine of a zero input), then the
            # plain affine and a per-sample factor exp(-+mean) on the coupled half; log-det of the coupling is 0
            ssum = new_logdet(x0)
            ops.affine(None, st, False, shape=(B, l2, H, W), logdet=ssum)
            m = (ssum / float(l2 * n_pix)).to(torch.float32)
            tab = torch.exp(m if rev else -m).view(B, 1).expand(B, l2).contiguous()
            if rev:
                u = ops.concat_channels([x1, ops.scale_channels(ops.affine(x2, st, True), tab)])
            else:
                ops.affine(ops.scale_channels(x2, tab), st, False, out=u[:, l1:])
        else:
            ops.affine(x2, st, rev, logdet=acc, out=u[:, l1:])
        if rev:
            out = self._mix(u, hard, mix, False) if self.reverse_pre_permute else u   # :262-263
        else:
            if hard:
                out = ops.channel_affine(u, scale, offset, inverse=False, perm_out=mix[0])     # :194-196
            else:
                out = self._mix(ops.channel_affine(u, scale, offset, inverse=False), hard, mix, False)
        acc = acc + (-1) ** int(rev) * n_pix * log_scale_sum
        return (out,), as_jac(acc)

    def output_dims(self, input_dims):
        return input_dims
