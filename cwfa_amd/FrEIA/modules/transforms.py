"""Fixed / reshaping / topology operators on HIP kernels.

Reference: FrEIA/modules/fixed_transforms.py (PermuteRandom :11-46, Fixed1x1Conv :95-133), graph_topology.py
(Split :10-88, Concat :92-152), reshapes.py (HaarDownsampling :191-318, HaarUpsampling :321-374),
invertible_resnet.py (ActNorm :11-85).
"""
import warnings
from copy import deepcopy
from typing import Sequence, Union

import numpy as np
import torch
import torch.nn as nn

from ... import autograd as AG
from ... import ops
from .base import InvertibleModule

__all__ = ["PermuteRandom", "Fixed1x1Conv", "Split", "Concat", "HaarDownsampling", "HaarUpsampling", "ActNorm",
           "Split1D", "SplitChannel", "Concat1d", "ConcatChannel"]


def _inverse_perm(perm: np.ndarray) -> np.ndarray:
    inv = np.zeros_like(perm)
    inv[perm] = np.arange(len(perm))
    return inv


class PermuteRandom(InvertibleModule):
    """Fixed random channel permutation; ``y[:, i] = x[:, perm[i]]``.  fixed_transforms.py:11-46.

    The table comes from numpy's GLOBAL legacy RNG exactly as in the reference (reseeded only when ``seed`` is given),
    so a graph built here draws the same tables as one built with the reference."""

    def __init__(self, dims_in, dims_c=None, seed: Union[int, None] = None):
        super().__init__(dims_in, dims_c)
        self.in_channels = dims_in[0][0]
        if seed is not None:
            np.random.seed(seed)
        perm = np.random.permutation(self.in_channels)
        self.perm = nn.Parameter(torch.LongTensor(perm), requires_grad=False)
        self.perm_inv = nn.Parameter(torch.LongTensor(_inverse_perm(perm)), requires_grad=False)
        self.axis = 1

    def table(self, rev):
        return self.perm_inv if rev else self.perm

    def forward(self, x, rev=False, jac=True):
        if AG.tracking(x[0]):
            return [AG.gather(x[0], self.table(rev), 1, self.table(not rev))], 0.
        return [ops.gather(x[0], self.table(rev), 1)], 0.

    def output_dims(self, input_dims):
        if len(input_dims) != 1:
            raise ValueError(f"{self.__class__.__name__} can only use 1 input")
        return input_dims


class Fixed1x1Conv(InvertibleModule):
    """Dense fixed 1x1 convolution with an invertible matrix M.  fixed_transforms.py:95-133."""

    def __init__(self, dims_in, dims_c=None, M: torch.Tensor = None):
        super().__init__(dims_in, dims_c)
        if M is None:
            raise ValueError("Need to specify the M argument, the matrix to be multiplied.")
        self.M = nn.Parameter(M.t().view(*M.shape, 1, 1), requires_grad=False)
        self.M_inv = nn.Parameter(M.t().inverse().view(*M.shape, 1, 1), requires_grad=False)
        self.logDetM = nn.Parameter(torch.slogdet(M)[1], requires_grad=False)
        self._packed = {}

    def _bank(self, rev):
        w = self.M_inv if rev else self.M
        pc = self._packed.get(rev)
        if pc is None or pc.version != w._version or pc.src_ptr != w.data_ptr() or pc.epoch != ops.pack_epoch():
            pc = self._packed[rev] = ops.pack_conv_weight(w)
        return pc

    def forward(self, x, rev=False, jac=True):
        n_pixels = x[0][0, 0].numel()
        j = self.logDetM * n_pixels
        return (ops.conv2d(x[0], self._bank(rev)),), (-j if rev else j)

    def output_dims(self, input_dims):
        if len(input_dims) != 1:
            raise ValueError(f"{self.__class__.__name__} can only use 1 input")
        if len(input_dims[0]) != 3:
            raise ValueError(f"{self.__class__.__name__} requires 3D input (channels, height, width)")
        return input_dims


class Split(InvertibleModule):
    """Split along the channel axis: forward hands out VIEWS, reverse concatenates.  graph_topology.py:10-88."""

    def __init__(self, dims_in: Sequence[Sequence[int]], section_sizes: Union[int, Sequence[int]] = None,
                 n_sections: int = 2, dim: int = 0):
        super().__init__(dims_in)
        assert len(dims_in) == 1, "Split layer takes exactly one input tensor"
        assert len(dims_in[0]) >= dim, "Split dimension index out of range"
        self.dim = dim
        l_dim = dims_in[0][dim]
        if section_sizes is None:
            assert 2 <= n_sections, "'n_sections' must be a least 2"
            if l_dim % n_sections != 0:
                warnings.warn('Split will create sections of unequal size')
            self.split_size_or_sections = ([l_dim // n_sections + 1] * (l_dim % n_sections) +
                                           [l_dim // n_sections] * (n_sections - l_dim % n_sections))
        else:
            if isinstance(section_sizes, int):
                assert section_sizes < l_dim, "'section_sizes' too large"
            else:
                assert isinstance(section_sizes, (list, tuple)), "'section_sizes' must be either int or list/tuple of int"
                assert sum(section_sizes) <= l_dim, "'section_sizes' too large"
                if sum(section_sizes) < l_dim:
                    warnings.warn("'section_sizes' too small, adding additional section")
                    section_sizes = list(section_sizes) + [l_dim - sum(section_sizes)]
            self.split_size_or_sections = section_sizes

    def forward(self, x, rev=False, jac=True):
        if self.dim != 0:
            raise NotImplementedError("cwfa_amd Split: only the channel axis (dim=0) is on the HIP path")
        if rev:
            return [AG.concat(list(x)) if AG.tracking(list(x)) else ops.concat_channels(list(x))], 0
        return torch.split(x[0], self.split_size_or_sections, dim=1), 0     # views, no data movement

    def output_dims(self, input_dims):
        assert len(input_dims) == 1, "Split layer takes exactly one input tensor"
        sizes = self.split_size_or_sections
        if isinstance(sizes, int):
            l_dim = input_dims[0][self.dim]
            sizes = [sizes] * (l_dim // sizes) + ([l_dim % sizes] if l_dim % sizes else [])
        return [tuple(input_dims[0][j] if (j != self.dim) else section_size for j in range(len(input_dims[0])))
                for section_size in sizes]


class Concat(InvertibleModule):
    """Concatenate along the channel axis; reverse hands out views.  graph_topology.py:92-152."""

    def __init__(self, dims_in: Sequence[Sequence[int]], dim: int = 0):
        super().__init__(dims_in)
        assert len(dims_in) > 1, "Concatenation only makes sense for multiple inputs"
        assert len(dims_in[0]) >= dim, "Merge dimension index out of range"
        assert all(len(dims_in[i]) == len(dims_in[0]) for i in range(len(dims_in))), \
            "All input tensors must have same number of dimensions"
        assert all(dims_in[i][j] == dims_in[0][j] for i in range(len(dims_in)) for j in range(len(dims_in[i]))
                   if j != dim), "All input tensor dimensions except merge dimension must be identical"
        self.dim = dim
        self.split_size_or_sections = [dims_in[i][dim] for i in range(len(dims_in))]

    def forward(self, x, rev=False, jac=True):
        if self.dim != 0:
            raise NotImplementedError("cwfa_amd Concat: only the channel axis (dim=0) is on the HIP path")
        if rev:
            return torch.split(x[0], self.split_size_or_sections, dim=1), 0
        return [AG.concat(list(x)) if AG.tracking(list(x)) else ops.concat_channels(list(x))], 0

    def output_dims(self, input_dims):
        assert len(input_dims) > 1, "Concatenation only makes sense for multiple inputs"
        output_dims = deepcopy(list(input_dims[0]))
        output_dims[self.dim] = sum(input_dim[self.dim] for input_dim in input_dims)
        return [tuple(output_dims)]


def _deprecated_by(orig_class):
    class deprecated_class(orig_class):
        def __init__(self, *args, **kwargs):
            warnings.warn(F"{self.__class__.__name__} is deprecated and will be removed in the public release. "
                          F"Use {orig_class.__name__} instead.", DeprecationWarning)
            super().__init__(*args, **kwargs)
    return deprecated_class


Split1D = _deprecated_by(Split)
SplitChannel = _deprecated_by(Split)
Concat1d = _deprecated_by(Concat)
ConcatChannel = _deprecated_by(Concat)


class HaarDownsampling(InvertibleModule):
    """2x2 spatial Haar: [C,H,W] <-> [4C,H/2,W/2].  reshapes.py:191-318.

    The log-det returned is ``numel * jac_fwd`` forwards and ``numel * jac_rev`` in reverse, as python floats, exactly as
    the reference computes them (reshapes.py:278,290).  Unlike the reference, ``rev`` does not scale the caller's input
    tensor in place (reshapes.py:296 does when order_by_wavelet is off)."""

    def __init__(self, dims_in, dims_c=None, order_by_wavelet: bool = False, rebalance: float = 1.):
        super().__init__(dims_in, dims_c)
        if rebalance == 0:
            raise ValueError("'rebalance' argument must be != 0.")
        self.in_channels = dims_in[0][0]
        self.fac_fwd = 0.5 * rebalance
        self.jac_fwd = (np.log(16.) + 4 * np.log(self.fac_fwd)) / 4.
        self.fac_rev = 0.5 / rebalance
        self.jac_rev = (np.log(16.) + 4 * np.log(self.fac_rev)) / 4.
        self.permute = order_by_wavelet
        # kept for state_dict compatibility with the reference (a fixed +-1 filter bank, reshapes.py:246-256)
        hw = torch.ones(4, 1, 2, 2)
        hw[1, 0, 0, 1] = hw[1, 0, 1, 1] = -1
        hw[2, 0, 1, 0] = hw[2, 0, 1, 1] = -1
        hw[3, 0, 1, 0] = hw[3, 0, 0, 1] = -1
        self.haar_weights = nn.Parameter(torch.cat([hw] * self.in_channels, 0), requires_grad=False)

    def forward(self, x, c=None, jac=True, rev=False):
        inp = x[0]
        ndims = inp[0].numel()
        if not rev:
            return (ops.haar2d(inp, False, self.permute, self.fac_fwd),), ndims * self.jac_fwd
        return (ops.haar2d(inp, True, self.permute, self.fac_rev),), ndims * self.jac_rev

    def output_dims(self, input_dims):
        if len(input_dims) != 1:
            raise ValueError("HaarDownsampling must have exactly 1 input")
        if len(input_dims[0]) != 3:
            raise ValueError("HaarDownsampling can only transform 2D images of the shape CxWxH (channels, width, height)")
        c, w, h = input_dims[0]
        c2, w2, h2 = c * 4, w // 2, h // 2
        if c * h * w != c2 * h2 * w2:
            raise ValueError("Input cannot be cleanly reshaped, most likely because the input height or width are an "
                             "odd number")
        return ((c2, w2, h2),)


class HaarUpsampling(HaarDownsampling):
    """Inverse of HaarDownsampling.  reshapes.py:321-374."""

    def __init__(self, dims_in, dims_c=None, order_by_wavelet: bool = False, rebalance: float = 1.):
        inv_shape = self.output_dims(dims_in)
        super().__init__(inv_shape, dims_c, order_by_wavelet, rebalance)

    def forward(self, x, c=None, jac=True, rev=False):
        return super().forward(x, c=None, rev=not rev)

    def output_dims(self, input_dims):
        if len(input_dims) != 1:
            raise ValueError("i-revnet downsampling must have exactly 1 input")
        if len(input_dims[0]) != 3:
            raise ValueError("i-revnet downsampling can only tranform 2d images of the shape cxwxh (channels, width, height)")
        c, w, h = input_dims[0]
        c2, w2, h2 = c // 4, w * 2, h * 2
        if c * h * w != c2 * h2 * w2:
            raise ValueError("input cannot be cleanly reshaped, most likely because the input height or width are an "
                             "odd number")
        return ((c2, w2, h2),)


class ActNorm(InvertibleModule):
    """Per-channel affine with data-dependent initialisation on the first batch.  invertible_resnet.py:11-85.
    y = x*exp(scale_c) + bias_c;  log-det = +-H*W*sum(scale).  Loading a state_dict disables the data init."""

    def __init__(self, dims_in, dims_c=None, init_data: Union[torch.Tensor, None] = None):
        super().__init__(dims_in, dims_c)
        self.dims_in = dims_in[0]
        param_dims = [1, self.dims_in[0]] + [1 for i in range(len(self.dims_in) - 1)]
        self.scale = nn.Parameter(torch.zeros(*param_dims))
        self.bias = nn.Parameter(torch.zeros(*param_dims))
        if init_data is not None:
            self._initialize_with_data(init_data)
        else:
            self.init_on_next_batch = True

        def on_load_state_dict(*args):
            self.init_on_next_batch = False
        self._register_load_state_dict_pre_hook(on_load_state_dict)

    def _initialize_with_data(self, data):
        assert all([data.shape[i + 1] == self.dims_in[i] for i in range(len(self.dims_in))]), \
            "Can't initialize ActNorm layer, provided data don't match input dimensions."
        st = ops.channel_stats(data).view(-1, 2)               # per-channel (sum, sumsq) in float64
        n = data.numel() // data.shape[1]
        mean = st[:, 0] / n
        var = (st[:, 1] - st[:, 0] * mean) / (n - 1)            # unbiased, as torch.std
        scale = torch.log(1 / var.sqrt())
        with torch.no_grad():
            self.scale.data = scale.to(torch.float32).view_as(self.scale)
            self.bias.data = (-(mean * scale.exp())).to(torch.float32).view_as(self.bias)
        self.init_on_next_batch = False

    def forward(self, x, rev=False, jac=True):
        if self.init_on_next_batch:
            self._initialize_with_data(x[0])
        j = (self.scale.sum() * np.prod(self.dims_in[1:])).repeat(x[0].shape[0])
        if AG.tracking(x[0], self.scale, self.bias):         # scale / bias through torch's own graph ([C]-sized), the map through the HIP node
            y = AG.channel_affine(x[0], self.scale.exp().reshape(-1), self.bias.reshape(-1), inverse=rev)
            return [y], (-j if rev else j)
        es = self.scale.detach().exp().reshape(-1).contiguous()
        bs = self.bias.detach().reshape(-1).contiguous()
        if not rev:
            return [ops.channel_affine(x[0], es, bs, inverse=False)], j
        return [ops.channel_affine(x[0], es, bs, inverse=True)], -j

    def chain_stage(self, perm=None, axis=1):
        """This module as a stage of a fused step chain (GraphINN plan lowering): s = scale_c, t = bias_c, no clamp --
        ``y = exp(s) x + t`` / ``(x - t) exp(-s)`` with log-det +-sum s.  The chain kernels read s, t per element, so the two
        per-channel vectors are broadcast once per parameter version into [1,C,H,W] tables shared by the whole batch."""
        key = (self.scale._version, self.bias._version, self.scale.data_ptr(), self.bias.data_ptr(), ops.pack_epoch())
        if getattr(self, "_chain_tabs", None) is None or self._chain_tabs[0] != key:
            Cc, H, W = self.dims_in
            s_tab = self.scale.detach().reshape(1, Cc, 1, 1).expand(1, Cc, H, W).contiguous()
            t_tab = self.bias.detach().reshape(1, Cc, 1, 1).expand(1, Cc, H, W).contiguous()
            self._chain_tabs = (key, s_tab, t_tab)
        st, keep = ops.stage(self._chain_tabs[1], self._chain_tabs[2], "NONE", 1.0, perm=perm, axis=axis)
        st.s_bs = st.t_bs = 0                           # every sample reads the same table
        return st, keep

    def output_dims(self, input_dims):
        assert len(input_dims) == 1, "Can only use 1 input"
        return input_dims
