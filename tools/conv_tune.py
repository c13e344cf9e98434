#!/usr/bin/env python3
"""Build conv2d variants (-D knobs of cwfa_amd/csrc/conv2d.hip) and time them on the shapes of the hot path.
    python tools/conv_tune.py build            # here (no GPU): writes tools/_variants/lib_<name>.so
    python tools/conv_tune.py run              # on the GPU box: interleaved rounds in ONE process per variant"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "tools", "_variants")
VARIANTS = {     # name -> (source the -D flags apply to, flags)
    "base": ("conv2d.hip", []),
}
SHAPES = [  # (Cin, Cout, H, W, ks)
    (512, 512, 256, 256, 3), (1024, 1024, 128, 128, 3),
]


def build():
    from cwfa_amd import build as b
    os.makedirs(VDIR, exist_ok=True)
    for name, (vsrc, defs) in VARIANTS.items():
        objs = []
        for s in b.SOURCES:
            o = os.path.join(VDIR, f"{name}_{s[:-4]}.o") if s == vsrc else os.path.join(VDIR, f"common_{s[:-4]}.o")
            if s == vsrc or not os.path.exists(o) or os.path.getmtime(o) < os.path.getmtime(os.path.join(b.CSRC, s)):
                r = subprocess.run([b.HIPCC, *b.FLAGS, *(defs if s == vsrc else []), "-c", os.path.join(b.CSRC, s), "-o", o],
                                   capture_output=True, text=True)
                assert r.returncode == 0, r.stderr[-3000:]
            objs.append(o)
        subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VDIR, f"lib_{name}.so"),
                        *objs], check=True)


def run_one(name):
    import torch
    from cwfa_amd import _lib
    _lib.LIB_PATH = os.path.join(VDIR, f"lib_{name}.so")
    from cwfa_amd import ops
    ops.set_option("winograd_2d", int(os.environ.get("CWFA_TUNE_2D", "1")))
    ops.set_option("split_bf16", int(os.environ.get("CWFA_TUNE_SPLIT", "0")))
    res = {}
    for (cin, cout, H, W, ks) in SHAPES:
        x = torch.randn(1, cin, H, W, device="cuda")
        w = torch.randn(cout, cin, ks, ks, device="cuda") * 0.05
        pc = ops.pack_conv_weight(w)
        out = torch.empty(1, cout, H, W, device="cuda")
        for _ in range(3):
            ops.conv2d(x, pc, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            ops.conv2d(x, pc, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[f"{cin}->{cout} k{ks} @{H}"] = round(2.0 * cin * cout * ks * ks * H * W / ms / 1e9, 1)
    print(json.dumps({"variant": name, "TFLOPs": res}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "run":
        for name in (sys.argv[2:] or VARIANTS):
            subprocess.run([sys.executable, __file__, "one", name])
    else:
        run_one(sys.argv[2])
