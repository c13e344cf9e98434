#!/usr/bin/env python3
"""training.train_iteration at 512x512x96 for rocprofv3: python3 tools/train_prof.py <fp32|split_bf16> [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cwfa_amd import CWFA, ops, training

torch.manual_seed(0); np.random.seed(0)
dev = torch.device("cuda")
D, S = 96, 512
conv_inn, cond_nets = CWFA.build_networks(D, S, 5, device=dev)
gen = torch.Generator().manual_seed(17)
gt = torch.randn(1, D, S, S, generator=gen).to(dev)
views = torch.randn(1, 29, S, S, generator=gen).to(dev)
means = [(0.1 * torch.randn(1, D // 2 ** (n + 1), S, S, generator=gen)).to(dev) for n in range(4)]
ops.set_precision(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
one = lambda: training.train_iteration(conv_inn, cond_nets, gt, views, means)   # noqa: E731
one(); one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    one()
torch.cuda.synchronize()
print("ms per iteration", (time.perf_counter() - t0) / n * 1e3)
