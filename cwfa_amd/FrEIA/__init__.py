"""HIP-backed drop-in for the subset of FrEIA that CWFA uses (see cwfa_amd.install())."""
from . import framework, modules

__all__ = ["framework", "modules"]
