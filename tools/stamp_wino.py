#!/usr/bin/env python3
"""s_memtime stamps of one block of the Winograd kernel (variant built with -DCWFA_EXP_STAMP): where a chunk's time goes.
    python tools/stamp_wino.py build | run"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "tools", "_variants")
if sys.argv[1] == "build":
    from cwfa_amd import build as b
    os.makedirs(VDIR, exist_ok=True)
    objs = []
    for s in b.SOURCES:
        o = os.path.join(VDIR, f"stamp_{s[:-4]}.o")
        subprocess.run([b.HIPCC, *b.FLAGS, "-DCWFA_EXP_STAMP", "-c", os.path.join(b.CSRC, s), "-o", o], check=True)
        objs.append(o)
    subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VDIR, "lib_stamp.so"), *objs], check=True)
else:
    import numpy as np, torch
    from cwfa_amd import _lib
    _lib.LIB_PATH = os.path.join(VDIR, "lib_stamp.so")
    from cwfa_amd import ops
    x = torch.randn(1, 256, 512, 512, device="cuda")
    pc = ops.pack_conv_weight(torch.randn(256, 256, 3, 3, device="cuda") * 0.05)
    alpha = torch.tensor([0.25], device="cuda")
    for _ in range(3):
        ops.conv2d(x, pc, act="prelu", prelu_alpha=alpha)
    torch.cuda.synchronize()
    h = C.CDLL(_lib.LIB_PATH)
    buf = (C.c_longlong * (8 * 40 * 5))()
    assert h.cwfa_debug_stamps(buf, 8 * 40 * 5) == 0
    t = np.array(buf, dtype=np.int64).reshape(8, 40, 5)[:, :32]
    t0 = t[:, 4:28]                       # steady-state chunks
    names = ["stage-before (early commit + prefetch)", "mfmas", "stage-after (late commit)", "barrier wait"]
    for w in range(8):
        d = np.diff(t0[w], axis=1).mean(axis=0)
        loop = np.diff(t0[w, :, 0]).mean()
        print(f"wave {w}: " + "  ".join(f"{n.split()[0]} {v:7.0f}" for n, v in zip(names, d)) + f"   chunk {loop:7.0f} ticks")
