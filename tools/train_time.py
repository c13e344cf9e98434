#!/usr/bin/env python3
"""Training iteration over the whole pyramid at 512x512x96 (GPU box): training.train_iteration in fp32 and split precision, and
the same iteration written as the reference writes it (modules + torch autograd, CWFA.py:865-1027).  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F
from cwfa_amd import CWFA, ops, training

torch.manual_seed(0); np.random.seed(0)
dev = torch.device("cuda")
D, S = 96, 512
conv_inn, cond_nets = CWFA.build_networks(D, S, 5, device=dev)
gen = torch.Generator().manual_seed(17)
gt = torch.randn(1, D, S, S, generator=gen).to(dev)
views = torch.randn(1, 29, S, S, generator=gen).to(dev)
means = [(0.1 * torch.randn(1, D // 2 ** (n + 1), S, S, generator=gen)).to(dev) for n in range(4)]


def reference_style():
    """One iteration as run_CWFA does it: forward through the modules, loss with torch ops, full_loss.backward()."""
    return float(training.train_iteration_autograd(conv_inn, cond_nets, gt, views, means)["losses"][0])


def timeit(fn, n=3):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = {}
for mode in ("fp32", "split_bf16"):
    ops.set_precision(mode)
    out[mode] = {"manual_ms": round(timeit(lambda: training.train_iteration(conv_inn, cond_nets, gt, views, means)), 2),
                 "autograd_ms": round(timeit(reference_style), 2)}
    torch.cuda.empty_cache()
ops.set_precision("fp32")
print(json.dumps(out))
