#!/usr/bin/env python3
"""Time the fused Conv3d 1->32->1 of the four condition nets (512 x 512 x {48, 24, 12, 6}) on the fp32 MFMA kernel, the
split-bf16 kernel and its bf16 form (GPU box).  Prints one JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cwfa_amd import ops

K = 32
torch.manual_seed(0)
w1, b1 = torch.randn(K, 1, 3, 3, 3, device="cuda") * 0.3, torch.randn(K, device="cuda")
w2, b2 = torch.randn(1, K, 3, 3, 3, device="cuda") * 0.1, torch.randn(1, device="cuda")
a = torch.tensor([0.25], device="cuda")
out = {}
for mode in ("fp32", "split_bf16", "bf16"):
    ops.set_precision(mode)
    res, tot = {}, 0.0
    for D in (48, 24, 12, 6):
        x = torch.randn(1, D, 512, 512, device="cuda")
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
        tot += res[f"D{D}"]
    res["total_ms"] = round(tot, 4)
    out[mode] = res
ops.set_precision("fp32")
print(json.dumps(out))
