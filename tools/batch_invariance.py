#!/usr/bin/env python3
"""Are per-sample results independent of the batch they run in?  Forward pass of every flow step (+ its condition net) on 8 volumes at
once vs the same volumes in two batches of 4; reports the first tensors that differ bitwise.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cwfa_amd import CWFA, ops

torch.manual_seed(0); np.random.seed(0)
D, S = 96, 512
conv_inn, cond_nets = CWFA.build_networks(D, S, 5, with_lrnn=False, device="cuda")
ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else "split_bf16")
g = torch.Generator().manual_seed(3)
B = 8
views = torch.randn(B, 29, S, S, generator=g).cuda()
gt = torch.randn(B, D, S, S, generator=g).cuda()
means = [(0.1 * torch.randn(B, D // 2 ** (n + 1), S, S, generator=g)).cuda() for n in range(4)]


def run(sl):
    out = {}
    x = gt[sl]
    with torch.no_grad():
        for n, gi in enumerate(conv_inn):
            om = cond_nets[n](views[sl])[-1]
            out[f"omega{n}"] = om
            (z, low), ld = gi(x, c=[om, means[n][sl]])
            out[f"z{n}"], out[f"low{n}"], out[f"logdet{n}"] = z, low, ld
            x = low
    return out

full = run(slice(0, 8))
halves = [run(slice(0, 4)), run(slice(4, 8))]
for k in full:
    got = torch.cat([h[k] for h in halves], 0)
    same = torch.equal(got, full[k])
    d = (got.double() - full[k].double()).abs().max().item()
    print(f"{k:10s} {'identical' if same else 'DIFFERENT'}  max|d| = {d:.3e}  (max|ref| = {full[k].abs().max().item():.3e})")
