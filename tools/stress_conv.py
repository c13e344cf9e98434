#!/usr/bin/env python3
"""Random-shape stress of the default conv kernels against an fp64 torch-CPU reference (GPU box)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwfa_amd import ops
random.seed(7); torch.manual_seed(7)
F = torch.nn.functional
worst = 0.0
for it in range(60):
    ks = random.choice([1, 3, 3, 3, 7])
    B = random.choice([1, 2, 3]); Cin = random.choice([1, 3, 8, 29, 64, 65, 130]); Cout = random.choice([1, 6, 32, 33, 64, 96, 128, 200])
    if ks == 7: Cin, Cout = min(Cin, 29), min(Cout, 64)
    H = random.randint(1, 37); W = random.randint(1, 140)
    x = torch.randn(B, Cin, H, W); w = torch.randn(Cout, Cin, ks, ks) / (Cin * ks * ks) ** 0.5; b = torch.randn(Cout)
    pro = random.random() < 0.5; res = random.random() < 0.4; act = random.choice([None, "elu", "prelu"])
    alpha = torch.tensor([0.3])
    sc, sh, add = torch.rand(B, Cin) + 0.5, torch.randn(B, Cin), torch.randn(B, Cin, H, W)
    r = torch.randn(B, Cout, H, W)
    xin = x.double() * sc.double().view(B, -1, 1, 1) + sh.double().view(B, -1, 1, 1) + add.double() if pro else x.double()
    ref = F.conv2d(xin, w.double(), b.double(), padding=ks // 2)
    if act == "elu": ref = F.elu(ref)
    if act == "prelu": ref = F.prelu(ref, alpha.double())
    if res: ref = F.prelu(ref + r.double(), alpha.double())
    pc = ops.pack_conv_weight(w.cuda())
    kw = dict(bias=b.cuda(), act=act, prelu_alpha=alpha.cuda())
    if pro: kw.update(in_scale=sc.cuda(), in_shift=sh.cuda(), in_add=add.cuda())
    if res: kw.update(residual=r.cuda(), act2="prelu")
    y = ops.conv2d(x.cuda(), pc, **kw)
    e = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
    worst = max(worst, e)
    assert e < 5e-6, (ks, B, Cin, Cout, H, W, pro, res, act, e)
print("60 random convs ok, worst max-rel", worst)
