#!/usr/bin/env python3
"""Full-size (512x512x96, config 3) GPU-vs-oracle error, per scale (GPU box): the numbers quoted in DESIGN.md section 3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cwfa_amd import CWFA
from oracle import cwfa_oracle as O
torch.manual_seed(0); np.random.seed(0)
conv_inn, cond_nets = CWFA.build_networks(96, 512, 5, with_lrnn=True, device="cuda")
enc = cond_nets[-1]
enc.net.deconv[1].drop_out = 0
for cn in enc.net.conv3d:
    cn.drop_prob = 0.0
g = torch.Generator().manual_seed(1)
ci = torch.randn(1, 29, 512, 512, generator=g)
mc = [0.1 * torch.randn(1, 96 // 2 ** (n + 1), 512, 512, generator=g) for n in range(4)]
with torch.no_grad():
    out = CWFA.inverse_pass(conv_inn, cond_nets, ci.cuda(), [m.cuda() for m in mc]).cpu()
cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}
steps = []
for n, gi in enumerate(conv_inn):
    axes = {i: (m.axis if hasattr(m, "axis") else 1) for i, m in enumerate(gi.module_list) if hasattr(m, "perm")}
    steps.append({"inn": cpu(gi.state_dict()), "omega": cpu(cond_nets[n].state_dict()), "axes": axes})
with torch.no_grad():
    ref = O.inverse_pass(steps, None, ci, mc, lrnn_sd=cpu(enc.state_dict()), lrnn_train=True)[-1]
d = (out - ref)
print(f"full-size config-3 inverse: max|d|/max|ref| = {float(d.abs().max() / ref.abs().max()):.3e}, L2-rel = {float(d.norm() / ref.norm()):.3e}")
from cwfa_amd import ops
for mode in ("split_bf16", "bf16"):
    ops.set_precision(mode)
    try:
        with torch.no_grad():
            o2 = CWFA.inverse_pass(conv_inn, cond_nets, ci.cuda(), [m.cuda() for m in mc]).cpu()
    finally:
        ops.set_precision("fp32")
    d2 = o2 - ref
    print(f"  precision mode {mode}: max|d|/max|ref| = {float(d2.abs().max() / ref.abs().max()):.3e}, L2-rel = {float(d2.norm() / ref.norm()):.3e}")
