"""CPU, build container only: the committed recipe for the golden vectors (oracle/make_golden.py, run with no argument
as its header documents) still completes and regenerates every fixture under tests/golden bit-identically from the
imported reference.  Skipped where /root/reference does not exist (the GPU box)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")
def test_documented_regeneration_reproduces_the_fixtures(tmp_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", CWFA_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_golden.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    new = sorted(os.path.basename(p) for p in glob.glob(str(tmp_path / "*.npz")))
    old = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
    assert new == old, (set(new) ^ set(old))
    for name in new:
        a, b = np.load(tmp_path / name), np.load(os.path.join(ROOT, "tests", "golden", name))
        assert sorted(a.files) == sorted(b.files), name
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (name, k)
            same = np.array_equal(a[k], b[k], equal_nan=True) if a[k].dtype.kind == "f" else np.array_equal(a[k], b[k])
            assert same, (name, k)
