#!/usr/bin/env python3
"""Random-shape stress of the fused sub-network layer and of a CAT step round trip (GPU box)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cwfa_amd import ops, networks as N
random.seed(3); torch.manual_seed(3); np.random.seed(3)
F = torch.nn.functional
worst = 0.0
for it in range(25):
    B = random.choice([1, 2]); H = random.randint(1, 45); W = random.randint(1, 150)
    x = torch.randn(B, 64, H, W)
    w3, b3 = torch.randn(64, 64, 3, 3) / 24, torch.randn(64) * 0.1
    w1, b1 = torch.randn(64, 64, 1, 1) / 8, torch.randn(64) * 0.1
    xd = x.double()
    ref = F.elu(F.conv2d(F.elu(F.conv2d(xd, w3.double(), b3.double(), padding=1)), w1.double(), b1.double()) + xd)
    y = ops.subnet_layer(x.cuda(), ops.pack_conv_weight(w3.cuda()), b3.cuda(), ops.pack_1x1_panel(w1.cuda()), b1.cuda())
    e = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
    worst = max(worst, e)
    assert e < 5e-6, (B, H, W, e)
print("25 random fused layers ok, worst max-rel", worst)
worst = 0.0
for it in range(6):
    H = random.choice([8, 24, 40]); W = random.choice([16, 96, 130]); D = random.choice([8, 12, 16])
    cn, inns = N.conditional_wavelet_flow([D, H, W], [1, 29, H, W], N.wavelet_flow_subnetwork2D,
                                          lambda: N.cond_network(29, D // 2, 1, 3, [], 4), n_internal_ch=8, n_down_steps=1,
                                          use_permutations=True, block_type="CAT", n_blocks=4)
    g = inns[0].eval().cuda()
    with torch.no_grad():
        for p_ in g.parameters():
            if p_.is_floating_point():
                p_.add_(0.05 * torch.randn_like(p_))
    x = torch.randn(2, D, H, W).cuda()
    c = [torch.randn(2, D // 2, H, W).cuda(), 0.1 * torch.randn(2, D // 2, H, W).cuda()]
    with torch.no_grad():
        (z, low), jf = g(x, c=c, rev=False)
        xr, jr = g([z, low], c=c, rev=True)
        xr = xr[0] if isinstance(xr, (tuple, list)) else xr
    e = float((xr - x).abs().max() / x.abs().max())
    j = float((jf + jr).abs().max() / (jf.abs().max() + 1e-9))
    worst = max(worst, e)
    assert e < 2e-5 and j < 1e-5, (D, H, W, e, j)
print("6 random CAT steps: forward->inverse round trip ok, worst max-rel", worst)
