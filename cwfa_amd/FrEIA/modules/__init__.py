"""Invertible operators of the hot path, API-compatible with the reference's ``FrEIA.modules``
(same class names, constructor arguments, ``forward(x, c, rev, jac) -> (tuple, logdet)`` contract, ``output_dims``,
``state_dict`` keys) but executing on hand-written HIP kernels through ``cwfa_amd.ops`` / libcwfa_hip.so.

Out of scope (never referenced by the CWFA path): IRevNet*, Flatten, Reshape,
FixedLinearTransform, IResNetLayer, InvAuto*, OrthogonalTransform, HouseholderPerm, GaussianMixtureModel.
"""
from .base import InvertibleModule
from .coupling import (NICECouplingBlock, RNVPCouplingBlock, GLOWCouplingBlock, GINCouplingBlock,
                       AffineCouplingOneSided, ConditionalAffineTransform, AllInOneBlock)
from .transforms import (PermuteRandom, Fixed1x1Conv, Split, Concat, HaarDownsampling, HaarUpsampling, ActNorm,
                         Split1D, SplitChannel, Concat1d, ConcatChannel)

__all__ = ["InvertibleModule", "AllInOneBlock", "ActNorm", "NICECouplingBlock", "RNVPCouplingBlock",
           "GLOWCouplingBlock", "GINCouplingBlock", "AffineCouplingOneSided", "ConditionalAffineTransform",
           "PermuteRandom", "Fixed1x1Conv", "SplitChannel", "ConcatChannel", "Split", "Concat", "HaarDownsampling",
           "HaarUpsampling"]
