#!/usr/bin/env python3
"""3x3 convolutions with <= 48 outputs at 512 x 512 (the condition nets' 2-D part and the coarse steps' output convolutions): fp32
Winograd kernel vs the split-bf16 kernel's narrow tilings.  GPU box; one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwfa_amd import ops
res = {}
for cin, cout in ((64, 96), (64, 48), (64, 32), (64, 24), (64, 12), (64, 6), (29, 24), (29, 12), (29, 6), (24, 24), (12, 12), (6, 6)):
    x = torch.randn(1, cin, 512, 512, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
    b = torch.randn(cout, device="cuda")
    row = {}
    ops.SPLIT_3X3_NARROW_MAX = 48          # every bank on the split kernel's tilings (the library's rule keeps some on Winograd)
    for mode in ("fp32", "split_bf16"):
        ops.set_precision(mode)
        pc = ops.pack_conv_weight(w)
        f = lambda: ops.conv2d(x, pc, bias=b)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        row[mode] = round(1e3 * e0.elapsed_time(e1) / 20, 1)
    res[f"{cin}->{cout}"] = row
ops.set_precision("fp32")
print(json.dumps(res))
