#!/usr/bin/env python3
"""The split-bf16 fused sub-network layer (csrc/conv_split_layer.hip): accuracy against fp64 and timing at 64 ch @512^2,
optionally for -D variants of the kernel source.
    python tools/sl_tune.py build [name=-DFOO,-DBAR ...]     (here, no GPU needed)
    python tools/sl_tune.py run [name ...]                   (on the GPU box)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "tools", "_variants")
SRC = "conv_split_layer.hip"


def build(variants):
    from cwfa_amd import build as b
    b.build_all()
    os.makedirs(VDIR, exist_ok=True)
    for name, defs in variants.items():
        objs = []
        for s in b.SOURCES:
            o = os.path.join(b.CSRC, s.replace(".hip", ".o"))
            if s == SRC:
                o = os.path.join(VDIR, f"sl_{name}.o")
                r = subprocess.run([b.HIPCC, *b.FLAGS, *defs, "-c", os.path.join(b.CSRC, s), "-o", o], capture_output=True, text=True)
                assert r.returncode == 0, r.stderr[-3000:]
            objs.append(o)
        subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VDIR, f"libsl_{name}.so"), *objs], check=True)
        print("built", name, defs)


def run_one(name):
    import torch
    from cwfa_amd import _lib
    if name != "default":
        _lib.LIB_PATH = os.path.join(VDIR, f"libsl_{name}.so")
    from cwfa_amd import ops
    F = torch.nn.functional
    res = {"variant": name}
    g = torch.Generator().manual_seed(5)
    for tag, (B, H, W) in {"small": (2, 50, 70), "tile": (1, 16, 32)}.items():
        x = torch.randn(B, 64, H, W, generator=g)
        w3, b3 = torch.randn(64, 64, 3, 3, generator=g) / 24, torch.randn(64, generator=g) * 0.1
        w1, b1 = torch.randn(64, 64, 1, 1, generator=g) / 8, torch.randn(64, generator=g) * 0.1
        xd = x.double()
        ref = F.elu(F.conv2d(F.elu(F.conv2d(xd, w3.double(), b3.double(), padding=1)), w1.double(), b1.double()) + xd)
        y = ops.subnet_layer(x.cuda(), ops.pack_split_layer_weight(w3.cuda(), w1.cuda()), b3.cuda(), None, b1.cuda()).cpu().double()
        res["err_" + tag] = float((y - ref).abs().max() / ref.abs().max())
    for B in (1, 4, 10):
        x = torch.randn(B, 64, 512, 512, device="cuda")
        w3, b3 = torch.randn(64, 64, 3, 3, device="cuda") / 24, torch.randn(64, device="cuda") * 0.1
        w1, b1 = torch.randn(64, 64, 1, 1, device="cuda") / 8, torch.randn(64, device="cuda") * 0.1
        pb = ops.pack_split_layer_weight(w3, w1)
        pa, pn = ops.pack_conv_weight(w3), ops.pack_1x1_panel(w1)
        ya = ops.subnet_layer(x, pa, b3, pn, b1)
        yb = ops.subnet_layer(x, pb, b3, None, b1)
        res[f"vs_fp32_layer_B{B}"] = float((ya - yb).abs().max() / ya.abs().max())
        for mode, f in (("split", lambda: ops.subnet_layer(x, pb, b3, None, b1)), ("fp32", lambda: ops.subnet_layer(x, pa, b3, pn, b1)),
                        ("split_l1", lambda: ops.subnet_layer(x, pb, b3, None, b1, layout=1)), ("split_l2", lambda: ops.subnet_layer(x, pb, b3, None, b1, layout=2)),
                        ("split_l3", lambda: ops.subnet_layer(x, pb, b3, None, b1, layout=3))):
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            res[f"{mode}_us_B{B}"] = round(1e3 * e0.elapsed_time(e1) / 20, 1)
        if B == 1:
            ops.set_option("split_products", 1)
            f = lambda: ops.subnet_layer(x, pb, b3, None, b1)
            yc = f()
            res["bf16_vs_fp32"] = float((ya - yc).abs().max() / ya.abs().max())
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            res["bf16_us_B1"] = round(1e3 * e0.elapsed_time(e1) / 20, 1)
            ops.set_option("split_products", 6)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build({a.split("=", 1)[0]: [d for d in a.split("=", 1)[1].split(",") if d] for a in sys.argv[2:]})
    elif sys.argv[1] == "run":
        for name in (sys.argv[2:] or ["default"]):
            subprocess.run([sys.executable, __file__, "one", name])
    else:
        run_one(sys.argv[2])
