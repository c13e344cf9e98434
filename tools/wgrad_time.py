#!/usr/bin/env python3
"""3x3 weight gradient: fp32 MFMA rows form vs the split-bf16 form (option "wgrad_split") at the training shapes (GPU box)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwfa_amd import ops

res = {}
KS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for (ci, co, s) in ((64, 64, 512), (256, 256, 512), (512, 512, 256), (1024, 1024, 128), (12, 256, 512), (512, 256, 512), (64, 96, 512)):
    x = torch.randn(1, ci, s, s, device="cuda")
    dy = torch.randn(1, co, s, s, device="cuda")
    row = {}
    for mode in (0, 1):
        ops.set_option("wgrad_split", mode)
        for _ in range(2):
            ops.conv2d_wgrad(x, dy, KS, want_bias=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv2d_wgrad(x, dy, KS, want_bias=True)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        row["split" if mode else "fp32"] = [round(us, 1), round(2.0 * co * ci * KS * KS * s * s / us / 1e6, 1)]      # us, TF/s algorithmic
    res[f"{ci}->{co}@{s}"] = row
ops.set_option("wgrad_split", 0)
print(json.dumps(res))
