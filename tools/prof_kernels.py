#!/usr/bin/env python3
"""Run the two dominant kernels a few times in isolation (for rocprofv3 --pmc / --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwfa_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 64, 512, 512, generator=g).cuda()
w3 = (torch.randn(64, 64, 3, 3, generator=g) / 24).cuda()
w1 = (torch.randn(64, 64, 1, 1, generator=g) / 8).cuda()
b = torch.zeros(64).cuda()
pc3, pn = ops.pack_conv_weight(w3), ops.pack_1x1_panel(w1)
xu = torch.randn(1, 256, 512, 512, generator=g).cuda()
wu = (torch.randn(256, 256, 3, 3, generator=g) / 48).cuda()
pcu = ops.pack_conv_weight(wu)
alpha = torch.tensor([0.25]).cuda()
sc, sh = torch.ones(256).cuda(), torch.zeros(256).cuda()
for _ in range(reps):
    ops.subnet_layer(x, pc3, b, pn, b)
    ops.conv2d(x, pc3, bias=b, act="elu")
    ops.conv2d(xu, pcu, act="prelu", prelu_alpha=alpha)
    ops.conv2d(xu, pcu, act="prelu", prelu_alpha=alpha, in_scale=sc, in_shift=sh)
torch.cuda.synchronize()
