"""Operator contract shared by every invertible module (reference: FrEIA/modules/base.py:7-112)."""
from typing import Iterable, List, Tuple

import torch
import torch.nn as nn

__all__ = ["InvertibleModule", "new_logdet", "as_jac"]


class InvertibleModule(nn.Module):
    """``module(x_or_z: tuple[Tensor], c=..., rev=False, jac=True) -> (tuple[Tensor], logdet)``.

    ``logdet`` is log|det J| of the direction that was evaluated (so ``jac_rev == -jac_fwd``), a Tensor[B] or a
    python number when it is constant.  ``output_dims(input_dims)`` does shape inference at graph-build time.
    """

    def __init__(self, dims_in: Iterable[Tuple[int]], dims_c: Iterable[Tuple[int]] = None):
        super().__init__()
        self.dims_in = list(dims_in)
        self.dims_c = list(dims_c) if dims_c is not None else []

    def forward(self, x_or_z, c=None, rev: bool = False, jac: bool = True):
        raise NotImplementedError(f"{self.__class__.__name__} does not provide forward(...) method")

    def log_jacobian(self, *args, **kwargs):
        raise DeprecationWarning("module.log_jacobian(...) is deprecated. module.forward(..., jac=True) returns a "
                                 "tuple (out, jacobian) now.")

    def output_dims(self, input_dims: List[Tuple[int]]) -> List[Tuple[int]]:
        raise NotImplementedError(f"{self.__class__.__name__} does not provide output_dims(...)")


def new_logdet(x: torch.Tensor) -> torch.Tensor:
    """float64[B] device accumulator the kernels add into (order-independent to fp32 rounding)."""
    return torch.zeros(x.shape[0], dtype=torch.float64, device=x.device)


def as_jac(acc: torch.Tensor) -> torch.Tensor:
    """The reference returns the log-det in the activation dtype (fp32)."""
    return acc.to(torch.float32)
