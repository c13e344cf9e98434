#!/usr/bin/env python3
"""Variants (-D knobs of conv2d.hip) of the experimental split fused layer, timed on 64 ch @512^2.
    python tools/sl_tune.py build | run"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "tools", "_variants")
VARIANTS = {"base": [], "skip1x1": ["-DCWFA_EXP_SL_SKIP1X1"], "skip1x1_nobound": ["-DCWFA_EXP_SL_SKIP1X1", "-DCWFA_EXP_SL_NOBOUNDARY"]}


def build():
    from cwfa_amd import build as b
    os.makedirs(VDIR, exist_ok=True)
    for name, defs in VARIANTS.items():
        objs = []
        for s in b.SOURCES:
            o = os.path.join(VDIR, f"sl{name}_{s[:-4]}.o") if s == "conv2d.hip" else os.path.join(VDIR, f"slcommon_{s[:-4]}.o")
            if s == "conv2d.hip" or not os.path.exists(o) or os.path.getmtime(o) < os.path.getmtime(os.path.join(b.CSRC, s)):
                r = subprocess.run([b.HIPCC, *b.FLAGS, *(defs if s == "conv2d.hip" else []), "-c", os.path.join(b.CSRC, s), "-o", o],
                                   capture_output=True, text=True)
                assert r.returncode == 0, r.stderr[-3000:]
            objs.append(o)
        subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VDIR, f"libsl_{name}.so"), *objs], check=True)


def run_one(name):
    import torch
    from cwfa_amd import _lib
    _lib.LIB_PATH = os.path.join(VDIR, f"libsl_{name}.so")
    from cwfa_amd import ops
    x = torch.randn(1, 64, 512, 512, device="cuda")
    w3, b3 = torch.randn(64, 64, 3, 3, device="cuda") / 24, torch.randn(64, device="cuda") * 0.1
    w1, b1 = torch.randn(64, 64, 1, 1, device="cuda") / 8, torch.randn(64, device="cuda") * 0.1
    pn, pb = ops.pack_1x1_panel(w1), ops.pack_split_layer_weight(w3)
    f = lambda: ops.subnet_layer(x, pb, b3, pn, b1)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"variant": name, "layer_us": round(1e3 * e0.elapsed_time(e1) / 20, 1)}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "run":
        for name in (sys.argv[2:] or VARIANTS):
            subprocess.run([sys.executable, __file__, "one", name])
    else:
        run_one(sys.argv[2])
