"""CPU: the drop-in boundary.  The C-ABI library loads, exports every symbol include/cwfa_hip.h declares, validates its
arguments without touching a GPU, and the product never routes through the oracle or a CPU fallback."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from cwfa_amd import build
    return build.build_all()


def _declared():
    src = open(os.path.join(ROOT, "include", "cwfa_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cwfa_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built_lib):
    names = _declared()
    assert len(names) >= 20
    h = ctypes.CDLL(built_lib)
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/cwfa_hip.h but not exported by libcwfa_hip.so"
    from cwfa_amd import _lib
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert _lib.lib().cwfa_version() == 100


def test_argument_validation_without_gpu(built_lib):
    """Error paths return codes + messages; nothing aborts, nothing is launched (no GPU here)."""
    from cwfa_amd import _lib
    L = _lib.lib()
    assert L.cwfa_haar1d_fwd_f32(None, None, None, 1, 4, 16, 64, 32, 32, None) == -1
    assert b"null" in L.cwfa_last_error()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.cwfa_haar1d_fwd_f32(p, p, p, 1, 3, 16, 64, 32, 32, None) == -2      # odd depth
    assert b"odd" in L.cwfa_last_error()
    assert L.cwfa_gather_f32(p, p, p, 1, 1, 1, 1, 7, 1, 1, None) == -1            # bad axis
    assert L.cwfa_conv2d_packed_floats(64, 64, 5) == -1                            # unsupported kernel size
    assert L.cwfa_conv2d_packed_floats(64, 64, 3) == 12 * 8 * 64 * 8               # Winograd F(2,3): 3 x 4 taps
    assert L.cwfa_set_option(b"winograd_min_cout", 1 << 20) == 0
    assert L.cwfa_conv2d_packed_floats(64, 64, 3) == 9 * 8 * 64 * 8                # direct kernel: 9 taps
    assert L.cwfa_set_option(b"winograd_min_cout", 1) == 0
    assert L.cwfa_set_option(b"no_such_option", 1) == -1
    assert L.cwfa_conv2d_f32(p, p, p, 1, 4, 8, 8, 4, 5, 256, 256, None, None) == -2
    # empty problems are accepted and do nothing
    assert L.cwfa_haar1d_fwd_f32(p, p, p, 0, 4, 16, 64, 32, 32, None) == 0
    assert L.cwfa_affine_f32(None, p, ctypes.byref(_lib.AffineStage()), 0, 0, 4, 4, 4, 0, 0, None, None, None) == 0


def test_couple_row_interleaving(built_lib):
    """cwfa_couple_rows: every source row of a [2n] coupling bank appears exactly once, s_j and t_j in the same 4-row lane
    group (n <= 32) / the same row of two adjacent 16-row m-tiles (n > 32)."""
    from cwfa_amd import _lib
    L = _lib.lib()
    assert L.cwfa_couple_rows(0, None) == -1 and L.cwfa_couple_rows(65, None) == -1
    for n in (1, 3, 24, 32, 33, 48, 64):
        total = L.cwfa_couple_rows(n, None)
        assert total == (64 if n <= 32 else 128)
        rows = (ctypes.c_int * total)()
        assert L.cwfa_couple_rows(n, rows) == total
        rows = list(rows)
        assert sorted(r for r in rows if r >= 0) == list(range(2 * n))
        for j in range(n):
            rs, rt = rows.index(j), rows.index(n + j)
            assert (rt == rs + 2 and rs % 4 < 2) if n <= 32 else (rt == rs + 16 and (rs // 16) % 2 == 0)
    assert L.cwfa_conv3x3_split_couple_f32(None, None, None, 1, 8, 8, 8, 0, None, None) == -1
    cp = _lib.Couple()
    cp.n = 65
    assert L.cwfa_conv3x3_split_couple_f32(None, None, None, 1, 8, 8, 8, 0, ctypes.byref(cp), None) == -2


def test_layout_and_two_source_options_are_validated(built_lib):
    """The options added for the sub-network maps are rejected where they do not apply (no launch happens: no GPU here)."""
    from cwfa_amd import _lib
    L = _lib.lib()
    buf = ctypes.create_string_buffer(256)
    p = ctypes.cast(buf, ctypes.c_void_p)
    o = _lib.ConvOpts()
    o.in_cat, o.in_cat_from, o.in_cat_c1 = p.value, 24, 24                 # in_cat_from must be a multiple of 16
    assert L.cwfa_conv2d_f32(p, p, p, 1, 72, 8, 8, 64, 1, 4608, 4096, ctypes.byref(o), None) == -1
    assert b"in_cat" in L.cwfa_last_error()
    o = _lib.ConvOpts()
    o.out_blocked8 = 1                                                       # blocked output: 1x1 banks only here
    assert L.cwfa_conv2d_f32(p, p, p, 1, 64, 8, 8, 64, 3, 4096, 4096, ctypes.byref(o), None) == -1
    o = _lib.ConvOpts()
    o.in_blocked8 = 1                                                        # blocked input is read by the split 3x3 kernel only
    assert L.cwfa_conv2d_f32(p, p, p, 1, 64, 8, 8, 64, 1, 4096, 4096, ctypes.byref(o), None) == -1
    assert L.cwfa_subnet_layer_split_f32(p, p, p, p, p, 1, 8, 8, 4096, 4096, 7, None) == -1        # layout not in 0..3
    assert L.cwfa_conv7x7_split_packed_bytes(65, 64) == -1 and L.cwfa_conv7x7_split_packed_bytes(64, 64) == 98 * 3 * 4 * 64 * 16
    assert L.cwfa_haar3d_fwd_f32(p, p, 1, 3, 8, 8, 1, 0.5, 192, None) == -2                         # odd depth


def test_ops_fail_loudly_on_cpu_tensors(built_lib):
    from cwfa_amd import ops
    from cwfa_amd.INN_utils import HaarTransform1D
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.haar1d(torch.zeros(1, 4, 2, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HaarTransform1D([(4, 2, 2)])((torch.zeros(1, 4, 2, 2),))


def test_missing_extension_is_fatal(tmp_path):
    code = ("import cwfa_amd._lib as L; L.LIB_PATH='/nonexistent/libcwfa_hip.so'\n"
            "try:\n    L.lib()\nexcept L.CwfaHipError as e:\n    print('LOUD', e)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert "LOUD" in out.stdout and "no CPU / PyTorch fallback" in out.stdout


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "cwfa_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports the oracle"
                assert "/root/reference" not in txt, f"{f} mentions the reference path"
    code = "import sys, cwfa_amd, cwfa_amd.CWFA, cwfa_amd.networks; print([m for m in sys.modules if m.split('.')[0]=='oracle'])"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert out.stdout.strip() == "[]", out.stdout + out.stderr
