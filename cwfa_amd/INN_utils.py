"""CWFA's own invertible modules on HIP kernels (reference: INN_utils.py).

In scope: ``HaarTransform1D`` (:126-174) and ``PermuteDim`` (:46-87).  ``Inv2Dto3D`` / ``Inv3Dto2D`` / ``turn_*`` are
never instantiated by the CWFA path and are not provided.
"""
from typing import Union

import numpy as np
import torch
import torch.nn as nn

from . import autograd as AG
from . import ops
from .FrEIA import modules as Fm

__all__ = ["HaarTransform1D", "PermuteDim"]


class PermuteDim(Fm.InvertibleModule):
    """Fixed random permutation of the ROWS (axis 2) or COLUMNS (axis 3) of every channel.  INN_utils.py:46-87.

    numpy global-RNG call order is the reference's: the axis is drawn BEFORE the optional reseed (:61-64), then
    ``permutation(H or W)``.  The axis is not part of the state_dict in the reference either; it is exposed as
    ``dims_to_permute`` (reference name) and ``axis``."""

    def __init__(self, dims_in, dims_c=None, dims_to_permute=[1, 2], seed: Union[int, None] = None):
        super().__init__(dims_in, dims_c)
        options = [[1, 2], [1, 3]]
        self.in_channels = dims_in[0][0]
        self.dims_to_permute = options[np.random.randint(0, len(options))]
        if seed is not None:
            np.random.seed(seed)
        perm = np.random.permutation(dims_in[0][self.dims_to_permute[1] - 1])
        inv = np.zeros_like(perm)
        inv[perm] = np.arange(len(perm))
        self.perm = nn.Parameter(torch.LongTensor(perm), requires_grad=False)
        self.perm_inv = nn.Parameter(torch.LongTensor(inv), requires_grad=False)

    @property
    def axis(self):
        return self.dims_to_permute[1]

    def table(self, rev):
        return self.perm_inv if rev else self.perm

    def forward(self, x, rev=False, jac=True):
        if AG.tracking(x[0]):
            return [AG.gather(x[0], self.table(rev), self.axis, self.table(not rev))], 0.
        return [ops.gather(x[0], self.table(rev), self.axis)], 0.

    def output_dims(self, input_dims):
        if len(input_dims) != 1:
            raise ValueError(f"{self.__class__.__name__} can only use 1 input")
        return input_dims


class HaarTransform1D(Fm.InvertibleModule):
    """Orthonormal Haar transform along the depth (= channel) axis; spatial size unchanged.  INN_utils.py:126-174.
    fwd: channels [0,h) = (even+odd)/sqrt2, [h,2h) = (even-odd)/sqrt2.  ``order_by_wavelet`` is accepted and ignored,
    as in the reference (:130-140).  log-det = +-numel*jac_{fwd,rev} (0 for rebalance=1), a python float."""

    def __init__(self, dims_in, dims_c=None, order_by_wavelet: bool = False, rebalance: float = 1.):
        super().__init__(dims_in, dims_c)
        self.fac_fwd = 0.5 * rebalance
        self.fac_rev = 0.5 / rebalance
        self.jac_fwd = (np.log(16.) + 4 * np.log(self.fac_fwd)) / 4.
        self.jac_rev = (np.log(16.) + 4 * np.log(self.fac_rev)) / 4.

    def forward(self, x_in, c=None, jac=True, rev=False):
        x = x_in[0]
        ndims = x[0].numel()
        out = AG.haar1d(x, rev) if AG.tracking(x) else ops.haar1d(x, rev)
        return (out,), (-ndims * self.jac_rev if rev else ndims * self.jac_fwd)

    def output_dims(self, input_dims):
        if len(input_dims) != 1:
            raise ValueError("HaarDownsampling must have exactly 1 input")
        if len(input_dims[0]) != 3:
            raise ValueError("HaarDownsampling can only transform 2D imagesof the shape CxWxH (channels, width, height)")
        c2, w2, h2 = input_dims[0]
        return ((c2, w2, h2),)
