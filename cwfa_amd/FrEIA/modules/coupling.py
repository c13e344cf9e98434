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

from ... import ops
from .base import InvertibleModule, as_jac, new_logdet

__all__ = ["NICECouplingBlock", "RNVPCouplingBlock", "GLOWCouplingBlock", "GINCouplingBlock",
           "AffineCouplingOneSided", "ConditionalAffineTransform", "AllInOneBlock"]


def subnet_st(net, parts, n_s):
    """Run a sub-network on the (virtual) channel concatenation of ``parts`` -> (s_raw, t, t_neg_div_sqrt2)."""
    if hasattr(net, "affine_parts"):
        return net.affine_parts(list(parts), n_s)
    u = parts[0] if len(parts) == 1 else ops.concat_channels(parts)
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

    def forward(self, x, c=[], rev=False, jac=True):
        x0 = x[0]
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
        s_raw, t = self._nets(which, parts, n_out)
        ops.affine(xa, self._stage(s_raw, t, gin=self._gin), rev, logdet=acc, out=ya)


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
        u = parts[0] if len(parts) == 1 else ops.concat_channels(parts)
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
        u = parts[0] if len(parts) == 1 else ops.concat_channels(parts)
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
        s_raw, t, _ = subnet_st(self.subnet, self._cond(x1, c), self.split_len2)
        out = ops.concat_channels([x1, x2])           # x1 passes through; x2 half is overwritten below
        acc = new_logdet(x0)
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
        acc = new_logdet(x[0])
        y = ops.affine(x[0], self.stage(c), rev, logdet=acc)
        return (y,), as_jac(acc)


class AllInOneBlock(InvertibleModule):
    """GLOW-style block: one-sided affine coupling (tanh clamp, coefficients x0.1) + per-channel global affine
    (softplus) + fixed channel permutation.  all_in_one_block.py:45-268.

    Supported: hard permutation, SOFTPLUS / SIGMOID / EXP global affine, conditions.  Not on the HIP path (unused by
    CWFA, which builds the block with defaults, networks.py:295,341-351): gin_block, permute_soft,
    learned_householder_permutation, reverse_permutation -> NotImplementedError at construction.
    """

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
        if gin_block or permute_soft or learned_householder_permutation or reverse_permutation:
            raise NotImplementedError("AllInOneBlock: gin_block / permute_soft / learned_householder_permutation / "
                                      "reverse_permutation are outside the CWFA hot path and have no HIP kernel")
        if len(dims_c) == 0:
            self.conditional, self.condition_channels = False, 0
        else:
            assert tuple(dims_c[0][1:]) == tuple(dims_in[0][1:]), \
                F"Dimensions of input and condition don't agree: {dims_c} vs {dims_in}."
            self.conditional, self.condition_channels = True, sum(dc[0] for dc in dims_c)
        self.splits = [channels - channels // 2, channels // 2]
        self.in_channels = channels
        self.clamp = affine_clamping
        self.GIN = False
        self.global_affine_type = global_affine_type
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
        # the reference draws this from numpy's global RNG (all_in_one_block.py:147): w[i, perm[i]] = 1
        w = np.zeros((channels, channels))
        for i, j in enumerate(np.random.permutation(channels)):
            w[i, j] = 1.
        self.w_perm = nn.Parameter(torch.FloatTensor(w).view(channels, channels, 1, 1), requires_grad=False)
        self.w_perm_inv = nn.Parameter(torch.FloatTensor(w.T).view(channels, channels, 1, 1), requires_grad=False)
        if subnet_constructor is None:
            raise ValueError("Please supply a callable subnet_constructor function or object (see docstring)")
        self.subnet = subnet_constructor(self.splits[0] + self.condition_channels, 2 * self.splits[1])
        self.last_jac = None
        self._tables = None

    # ---- tiny parameter-side tables ([C] vectors; rebuilt when the parameters change)
    def _prepare(self):
        key = (self.w_perm._version, self.global_scale._version, self.global_offset._version, self.w_perm.data_ptr(),
               self.global_scale.data_ptr())
        if self._tables is None or self._tables[0] != key:
            w = self.w_perm.detach()[:, :, 0, 0]
            if not bool(((w == 0) | (w == 1)).all()) or not bool((w.sum(0) == 1).all() and (w.sum(1) == 1).all()):
                raise NotImplementedError("AllInOneBlock: w_perm is not a hard permutation matrix")
            perm = w.argmax(1).contiguous()                     # fwd: out[:, i] = v[:, perm[i]]
            perm_inv = w.t().argmax(1).contiguous()             # rev: out[:, i] = x[:, perm_inv[i]]
            g = self.global_scale.detach().reshape(-1)
            if self.global_affine_type == 'SOFTPLUS':
                scale = 0.1 * torch.nn.functional.softplus(g, beta=0.5)
            elif self.global_affine_type == 'SIGMOID':
                scale = 10 * torch.sigmoid(g - 2.)
            else:
                scale = torch.exp(g)
            self._tables = (key, perm, perm_inv, scale.contiguous(), self.global_offset.detach().reshape(-1).contiguous(),
                            torch.log(scale).sum().to(torch.float64))
        return self._tables[1:]

    def forward(self, x, c=[], rev=False, jac=True):
        perm, perm_inv, scale, offset, log_scale_sum = self._prepare()
        x0 = x[0]
        l1, l2 = self.splits
        n_pix = x0.shape[2] * x0.shape[3]
        acc = new_logdet(x0)
        if rev:
            v = ops.channel_affine(x0, scale, offset, inverse=True, perm_in=perm_inv)       # all_in_one_block.py:191-193
            x1, x2 = v[:, :l1], v[:, l1:]
            s_raw, t, _ = subnet_st(self.subnet, [x1, *c] if self.conditional else [x1], l2)
            st = ops.stage(s_raw, t, "TANH", self.clamp, pre_scale=0.1)                     # `a *= 0.1`, :213
            ops.affine(x2, st, True, logdet=acc, out=x2)
            out = v
        else:
            x1, x2 = x0[:, :l1], x0[:, l1:]
            s_raw, t, _ = subnet_st(self.subnet, [x1, *c] if self.conditional else [x1], l2)
            u = ops.concat_channels([x1, x2])
            ops.affine(x2, ops.stage(s_raw, t, "TANH", self.clamp, pre_scale=0.1), False, logdet=acc, out=u[:, l1:])
            out = ops.channel_affine(u, scale, offset, inverse=False, perm_out=perm)        # :194-196
        acc = acc + (-1) ** int(rev) * n_pix * log_scale_sum
        return (out,), as_jac(acc)

    def output_dims(self, input_dims):
        return input_dims
