#!/usr/bin/env python3
"""Build variants of cwfa_amd/csrc/conv3d_split.hip (-D knobs) and time the fused Conv3d 1->32->1 on the four Omega-net sizes.
    python tools/c3s_tune.py build      (here)        python tools/c3s_tune.py run [names]     (GPU box)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "tools", "_variants")
SRC = "conv3d_split.hip"
VARIANTS = {"p2s": ["-DC3S_PIPE=2", "-DC3S_STAGGER=1"], "p2s_prio1": ["-DC3S_PIPE=2", "-DC3S_STAGGER=1", "-DC3S_PRIO=1"], "p2_prio1": ["-DC3S_PIPE=2", "-DC3S_PRIO=1"], "stamp_prio": ["-DC3S_PIPE=2", "-DC3S_STAGGER=1", "-DC3S_PRIO=1", "-DC3S_STAMP=1"]}


def build():
    from cwfa_amd import build as b
    os.makedirs(VDIR, exist_ok=True)
    for name, defs in VARIANTS.items():
        objs = []
        for s in b.SOURCES:
            o = os.path.join(VDIR, f"c3s{name}_{s[:-4]}.o") if s == SRC else os.path.join(b.CSRC, s.replace(".hip", ".o"))
            if s == SRC:
                r = subprocess.run([b.HIPCC, *b.FLAGS, *b.EXTRA.get(s, []), *defs, "-c", os.path.join(b.CSRC, s), "-o", o],
                                   capture_output=True, text=True)
                assert r.returncode == 0, r.stderr[-3000:]
            objs.append(o)
        subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VDIR, f"libc3s_{name}.so"), *objs], check=True)
        print("built", name, flush=True)


def run_one(name):
    import torch
    from cwfa_amd import _lib
    _lib.LIB_PATH = os.path.join(VDIR, f"libc3s_{name}.so")
    from cwfa_amd import ops
    K = 32
    torch.manual_seed(0)
    w1, b1 = torch.randn(K, 1, 3, 3, 3, device="cuda") * 0.3, torch.randn(K, device="cuda")
    w2, b2 = torch.randn(1, K, 3, 3, 3, device="cuda") * 0.1, torch.randn(1, device="cuda")
    a = torch.tensor([0.25], device="cuda")
    ops.set_precision("split_bf16")
    res, tot = {}, 0.0
    for D in (48, 24, 12, 6):
        x = torch.randn(1, D, 512, 512, device="cuda")
        ops.set_precision("fp32")
        ref = ops.conv3d_1k1(x, w1, b1, a, w2, b2)
        ops.set_precision("split_bf16")
        y = ops.conv3d_1k1(x, w1, b1, a, w2, b2)
        err = float((y - ref).abs().max() / ref.abs().max())
        for _ in range(3):
            ops.conv3d_1k1(x, w1, b1, a, w2, b2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv3d_1k1(x, w1, b1, a, w2, b2)
        e1.record()
        torch.cuda.synchronize()
        res[f"D{D}"] = round(e0.elapsed_time(e1) / 20, 4)
        res[f"err{D}"] = float(f"{err:.2e}")
        tot += res[f"D{D}"]
    res["total_ms"] = round(tot, 4)
    print(json.dumps({"variant": name, "ms": res}), flush=True)
    if name.startswith("stamp"):
        import ctypes
        x = torch.randn(1, 48, 512, 512, device="cuda")
        ops.conv3d_1k1(x, w1, b1, a, w2, b2)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 128)()
        _lib.lib().cwfa_dbg_c3s_stamps(buf)
        v = list(buf)
        for wv in range(2):
            for st in range(4):
                row = v[(wv * 4 + st) * 16:(wv * 4 + st) * 16 + 11]
                print("wave", wv * 4, "step", 10 + st, [row[i] - row[0] for i in range(11)], "abs0", row[0] - v[0], flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "run":
        for name in (sys.argv[2:] or VARIANTS):
            subprocess.run([sys.executable, __file__, "one", name])
    else:
        run_one(sys.argv[2])
