#!/usr/bin/env python3
"""Does the row / channel scatter of the 11 coefficient streams limit the chain kernel?  chain_inv at C = 48, 512 x 512 with (a)
identity permutations, (b) channel permutations only, (c) channel + row permutations (the real pattern).  GPU box."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwfa_amd import ops
g = torch.Generator().manual_seed(3)
H = W = 512
res = {}
for C_ in (48, 24):
    s_t = [(0.3 * torch.randn(1, C_, H, W, device="cuda"), torch.randn(1, C_, H, W, device="cuda")) for _ in range(5)]
    low = torch.randn(1, C_, H, W, device="cuda")
    for tag, axes, ident in (("identity", (1, 1, 1, 1, 1), True), ("chan", (1, 1, 1, 1, 1), False), ("chan+row", (1, 2, 1, 2, 1), False)):
        perms = [(torch.arange({1: C_, 2: H}[ax]) if ident else torch.randperm({1: C_, 2: H}[ax], generator=g)).cuda() for ax in axes]
        st = [ops.stage(s, t, perm=p, axis=ax) for (s, t), p, ax in zip(s_t, perms, axes)]
        tabs = ops.chain_tables([(p, ax) for p, ax in zip(perms, axes)], None, C_, H, W, low.device)
        f = lambda: ops.chain_inv(None, low, st, tables=tabs)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 20
        res[f"C{C_}_{tag}"] = (round(us, 1), round(4.0 * 13 * C_ * H * W / us / 1e3))
print(json.dumps(res))
