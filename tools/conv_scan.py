#!/usr/bin/env python3
"""Time conv2d on a list of shapes with the shipped library (fixed-cost vs per-chunk cost analysis)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwfa_amd import ops

def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

res = {}
for (cin, cout, H, ks) in [(64,64,512,3),(128,64,512,3),(256,64,512,3),(512,64,512,3),(64,128,512,3),(256,128,512,3),(8,64,512,3),(64,64,256,3),(64,64,1024,3)]:
    x = torch.randn(1, cin, H, H, device="cuda"); w = torch.randn(cout, cin, ks, ks, device="cuda") * .05
    pc = ops.pack_conv_weight(w); out = torch.empty(1, cout, H, H, device="cuda")
    ms = t(lambda: ops.conv2d(x, pc, out=out))
    res[f"{cin}->{cout} k{ks} @{H}"] = (round(ms*1e3,1), round(2.0*cin*cout*ks*ks*H*H/ms/1e9,1))
x = torch.randn(1, 64, 512, 512, device="cuda")
w3 = torch.randn(64,64,3,3, device="cuda")/24; w1 = torch.randn(64,64,1,1, device="cuda")/8; b = torch.zeros(64, device="cuda")
pc3, pn = ops.pack_conv_weight(w3), ops.pack_1x1_panel(w1)
ms = t(lambda: ops.subnet_layer(x, pc3, b, pn, b)); res["layer64 @512"] = (round(ms*1e3,1), round(2.0*64*64*10*512*512/ms/1e9,1))
ms = t(lambda: ops.conv2d(x, pc3, bias=b, act="elu")); res["64->64 elu @512"] = (round(ms*1e3,1), round(2.0*64*64*9*512*512/ms/1e9,1))
print(json.dumps(res))
