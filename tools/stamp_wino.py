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
    CH, HH = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 512)
    x = torch.randn(1, CH, HH, HH, device="cuda")
    pc = ops.pack_conv_weight(torch.randn(CH, CH, 3, 3, device="cuda") * 0.05)
    alpha = torch.tensor([0.25], device="cuda")
    for _ in range(3):
        ops.conv2d(x, pc, act="prelu", prelu_alpha=alpha)
    torch.cuda.synchronize()
    h = C.CDLL(_lib.LIB_PATH)
    N = 8 * 40 * 10
    buf = (C.c_longlong * (N + 64))()
    assert h.cwfa_debug_stamps(buf, N + 64) == 0
    t = np.array(buf, dtype=np.int64)[:N].reshape(8, 40, 10)[:, 4:28]     # steady-state chunks
    # stamps 0..7: k-steps 0,6,..,42; 8: after the barrier (before step 46); 9: end of chunk
    for w in range(8):
        seg = np.concatenate([np.diff(t[w][:, :8], axis=1), (t[w][:, 8] - t[w][:, 7])[:, None], (t[w][:, 9] - t[w][:, 8])[:, None]], axis=1).mean(axis=0)
        print(f"wave {w}: " + " ".join(f"{v:6.0f}" for v in seg) + f"   chunk {np.diff(t[w][:, 0]).mean():7.0f}")
    l = np.array(buf, dtype=np.int64)[N:].reshape(8, 8)[:, :3]
    c = np.array(buf, dtype=np.int64)[:N].reshape(8, 40, 10)
    for w in (0, 4):
        print(f"block wave {w}: prologue {c[w,0,0]-l[w,0]}  mainloop {l[w,1]-l[w,0]}  epilogue {l[w,2]-l[w,1]}")
    print("        (6 k-steps each: s0-5 s6-11 ... s36-41 | s42-45+barrier | s46-47)")
    x = torch.randn(1, 64, 512, 512, device="cuda")
    pc3 = ops.pack_conv_weight(torch.randn(64, 64, 3, 3, device="cuda") * 0.05)
    pn = ops.pack_1x1_panel(torch.randn(64, 64, 1, 1, device="cuda") * 0.1)
    b = torch.zeros(64, device="cuda")
    for _ in range(3):
        ops.subnet_layer(x, pc3, b, pn, b)
    torch.cuda.synchronize()
    assert h.cwfa_debug_stamps(buf, N + 64) == 0
    l = np.array(buf, dtype=np.int64)[N:].reshape(8, 8)[:, :4]
    c = np.array(buf, dtype=np.int64)[:N].reshape(8, 40, 10)[:, :8]
    for w in (0, 4):
        d = np.diff(l[w])
        print(f"layer wave {w}: mainloop {d[0]} (prologue {c[w,0,0]-l[w,0]}, chunks {np.diff(c[w,:,0])})  panel {d[1]}  1x1+store {d[2]}  total {l[w,3]-l[w,0]}")
