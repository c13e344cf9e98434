import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load one fixture (inputs / weights / expected outputs produced by oracle/make_golden.py)."""
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def sd_of(fx, prefix="sd/"):
    import torch
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(prefix)}


def rel_err(a, b):
    """(max|a-b|/max|b|, ||a-b||/||b||) -- the two measures of SURVEY.md section 8d."""
    import torch
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    den_m = max(float(b.abs().max()), 1e-30)
    den_2 = max(float(b.norm()), 1e-30)
    return float((a - b).abs().max()) / den_m, float((a - b).norm()) / den_2


def assert_close(a, b, tol=1e-4, what=""):
    m, l2 = rel_err(a, b)
    assert m <= tol and l2 <= tol, f"{what}: max-rel {m:.3e}, l2-rel {l2:.3e} > {tol:g}"


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
