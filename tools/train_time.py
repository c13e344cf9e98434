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
    gt_cache = [gt]
    with torch.no_grad():
        for _ in range(4):
            y = ops.haar1d(gt_cache[-1], False)
            gt_cache.append(y[:, :y.shape[1] // 2].contiguous())
    for m in list(conv_inn) + list(cond_nets):
        for p in m.parameters():
            p.grad = None
    up = cond_nets[4](views, means[3])[-1]
    F.mse_loss(gt_cache[4], up).backward()
    up = up.detach()
    for n in range(3, -1, -1):
        cond = [cond_nets[n](views)[-1].float(), means[n]]
        z = CWFA.sample_z_truncated((1,) + tuple(conv_inn[n].global_out_shapes[0]), device=dev, temperature=0)
        xhat, _ = conv_inn[n]([z, up], c=cond, rev=True)
        full = F.mse_loss(gt_cache[n], xhat) * 0.40984
        Z, ld = conv_inn[n](gt_cache[n], c=cond)
        full = full + (0.5 * torch.norm(Z[0]) ** 2 - ld.mean()) / xhat.numel() * (1 - 0.40984)
        full.backward()
        up = xhat.detach()
    return float(full)


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
