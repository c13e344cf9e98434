#!/usr/bin/env python3
"""Which torch operators still run inside one inverse pass (GPU box): aten op counts with the Python line that called them."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from cwfa_amd import CWFA
torch.manual_seed(0); np.random.seed(0)
dev = torch.device("cuda")
conv_inn, cond_nets = CWFA.build_networks(96, 512, 5, with_lrnn=True, device=dev)
g = torch.Generator().manual_seed(1)
cond_input = torch.randn(1, 29, 512, 512, generator=g).to(dev)
mean_cache = [(0.1 * torch.randn(1, 96 // 2 ** (n + 1), 512, 512, generator=g)).to(dev) for n in range(4)]
step = lambda: CWFA.inverse_pass(conv_inn, cond_nets, cond_input, mean_cache)
with torch.no_grad():
    step(); step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::"):
        st = e.stack or []
        fr = next((s for s in st if "cwfa_amd" in s), "?")
        cnt[(e.name, fr[-80:])] += 1
for (n, f), c in cnt.most_common(60):
    print(f"{c:4d} {n:28s} {f}")
