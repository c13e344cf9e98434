"""cwfa_amd -- MI355X-native implementation of CWFA's inverse (reconstruction) / forward-NLL hot path.

Python mirrors of the reference's modules (``FrEIA``, ``INN_utils``, ``networks``, ``unet``, hot-path part of ``CWFA``)
over hand-written HIP kernels in ``libcwfa_hip.so`` (C ABI: include/cwfa_hip.h).  There is no CPU or PyTorch fallback:
ops raise on non-HIP tensors and on a missing extension.

    import cwfa_amd; cwfa_amd.install()     # then the reference's main.py imports resolve to this package
"""
import sys

__version__ = "0.1.0"


def install():
    """Register this package's modules under the reference's top-level import names (FrEIA, INN_utils, networks, unet)
    so that code written against the reference (``import FrEIA.framework as Ff``, ``from networks import *``) runs on
    the HIP implementation unchanged.  Call before importing the reference's driver."""
    from . import FrEIA, INN_utils, networks, unet
    sys.modules["FrEIA"] = FrEIA
    sys.modules["FrEIA.framework"] = FrEIA.framework
    sys.modules["FrEIA.modules"] = FrEIA.modules
    sys.modules["INN_utils"] = INN_utils
    sys.modules["networks"] = networks
    sys.modules["unet"] = unet
    return FrEIA, INN_utils, networks, unet
