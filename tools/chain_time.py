#!/usr/bin/env python3
"""Time the fused chain kernels (inverse and forward) at the four pyramid shapes of the 512x512x96 configuration, with and
without column permutations, optionally for -D variants of elementwise.hip built into tools/_variants.
    python tools/chain_time.py build [name=-DFOO ...]      python tools/chain_time.py run [name ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "tools", "_variants")
SRC = "elementwise.hip"


def build(variants):
    from cwfa_amd import build as b
    b.build_all()
    os.makedirs(VDIR, exist_ok=True)
    for name, defs in variants.items():
        objs = []
        for s in b.SOURCES:
            o = os.path.join(b.CSRC, s.replace(".hip", ".o"))
            if s == SRC:
                o = os.path.join(VDIR, f"ch_{name}.o")
                r = subprocess.run([b.HIPCC, *b.FLAGS, *defs, "-c", os.path.join(b.CSRC, s), "-o", o], capture_output=True, text=True)
                assert r.returncode == 0, r.stderr[-3000:]
            objs.append(o)
        subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VDIR, f"libch_{name}.so"), *objs], check=True)
        print("built", name, defs)


def run_one(name):
    import torch
    from cwfa_amd import _lib
    if name != "default":
        _lib.LIB_PATH = os.path.join(VDIR, f"libch_{name}.so")
    from cwfa_amd import ops
    res = {"variant": name}
    g = torch.Generator().manual_seed(3)
    H = W = 512
    for C_ in (48, 24, 12, 6):
        for axes in ((1, 2, 1, 3, 1),):
            perms = [torch.randperm({1: C_, 2: H, 3: W}[ax], generator=g).cuda() for ax in axes]
            st = [ops.stage(0.3 * torch.randn(1, C_, H, W, device="cuda"), torch.randn(1, C_, H, W, device="cuda"), perm=p, axis=ax)
                  for p, ax in zip(perms, axes)]
            low = torch.randn(1, C_, H, W, device="cuda")
            x = torch.randn(1, 2 * C_, H, W, device="cuda")
            tabs = ops.chain_tables([(p, ax) for p, ax in zip(perms, axes)], None, C_, H, W, low.device)
            for tag, f, nb in (("inv", lambda: ops.chain_inv(None, low, st, tables=tabs), 13), ("fwd", lambda: ops.chain_fwd(x, st, tables=tabs), 14)):
                for _ in range(3): f()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): f()
                e1.record(); torch.cuda.synchronize()
                us = 1e3 * e0.elapsed_time(e1) / 20
                res[f"{tag}_C{C_}_{'col' if 3 in axes else 'row'}"] = (round(us, 1), round(4.0 * nb * C_ * H * W / us / 1e3))   # us, GB/s
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build({a.split("=", 1)[0]: [d for d in a.split("=", 1)[1].split(",") if d] for a in sys.argv[2:]})
    elif sys.argv[1] == "run":
        for name in (sys.argv[2:] or ["default"]):
            subprocess.run([sys.executable, __file__, "one", name])
    else:
        run_one(sys.argv[2])
