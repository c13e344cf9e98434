#!/usr/bin/env python3
"""Ablation builds of the weight-gradient kernel (-D knobs of cwfa_amd/csrc/conv_bwd.hip), timed on 64->64 3x3 @512^2.
    python tools/wg_tune.py build     (here)        python tools/wg_tune.py run      (GPU box)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "tools", "_variants")
VARIANTS = {"base": [], "nostage": ["-DCWFA_EXP_WG_NOSTAGE"], "nomfma": ["-DCWFA_EXP_WG_NOMFMA"],
            "nostage_nolds": ["-DCWFA_EXP_WG_NOSTAGE", "-DCWFA_EXP_WG_NOLDS"],
            "burst": ["-DCWFA_EXP_WG_BURST"], "rows": []}


def build():
    from cwfa_amd import build as b
    os.makedirs(VDIR, exist_ok=True)
    for name, defs in VARIANTS.items():
        objs = []
        for s in b.SOURCES:
            if s == "conv_bwd.hip":
                o = os.path.join(VDIR, f"wg_{name}.o")
                r = subprocess.run([b.HIPCC, *b.FLAGS, *defs, "-c", os.path.join(b.CSRC, s), "-o", o], capture_output=True, text=True)
                assert r.returncode == 0, r.stderr[-2000:]
            else:
                o = os.path.join(b.CSRC, s.replace(".hip", ".o"))
            objs.append(o)
        subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VDIR, f"libwg_{name}.so"), *objs], check=True)


def run_one(name):
    import torch
    from cwfa_amd import _lib
    _lib.LIB_PATH = os.path.join(VDIR, f"libwg_{name}.so")
    from cwfa_amd import ops
    if name != "rows":
        ops.set_option("wgrad_rows", 0)             # the knobs below belong to the first (register-staged) form
    x = torch.randn(1, 64, 512, 512, device="cuda")
    dy = torch.randn(1, 64, 512, 512, device="cuda")
    for _ in range(3):
        ops.conv2d_wgrad(x, dy, 3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d_wgrad(x, dy, 3)
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"variant": name, "us": round(e0.elapsed_time(e1) / 20 * 1e3, 1)}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "run":
        for name in VARIANTS:
            subprocess.run([sys.executable, __file__, "one", name])
    else:
        run_one(sys.argv[2])
