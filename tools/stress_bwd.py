#!/usr/bin/env python3
"""Random-shape stress of the backward kernels against float64 autograd on the CPU (GPU box): python tools/stress_bwd.py [n]
Weight gradients (1x1 / 3x3 both forms / 7x7), Conv3d stage, chain backward; prints the worst relative error per family."""
import math
import os
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import rel_err  # noqa: E402
from cwfa_amd import ops  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rnd = random.Random(7)
    g = torch.Generator().manual_seed(7)
    worst = {}
    for it in range(n):
        ks = rnd.choice([1, 3, 3, 3, 7])
        B, Cin, Cout = rnd.randint(1, 3), rnd.choice([1, 3, 29, 48, 64, 65, 96, 130]), rnd.choice([1, 6, 24, 64, 70, 96, 128])
        H = rnd.randint(1, 40)
        W = rnd.choice([rnd.randint(1, 70), 4 * rnd.randint(1, 24)])
        if ks == 7:
            Cin, Cout = min(Cin, 64), min(Cout, 64)
        x = torch.randn(B, Cin, H, W, generator=g)
        dy = torch.randn(B, Cout, H, W, generator=g)
        w = torch.zeros(Cout, Cin, ks, ks, dtype=torch.float64, requires_grad=True)
        (F.conv2d(x.double(), w, padding=ks // 2) * dy.double()).sum().backward()
        got, gb = ops.conv2d_wgrad(x.cuda(), dy.cuda(), ks, want_bias=True)
        e = max(max(rel_err(got, w.grad)), max(rel_err(gb, dy.double().sum((0, 2, 3)))))
        key = f"wgrad k{ks}" + (" rows" if ks == 3 and W % 4 == 0 else "")
        worst[key] = max(worst.get(key, (0, None)), (e, (B, Cin, Cout, H, W)))
    for it in range(max(n // 3, 4)):
        B, D, H, K = rnd.randint(1, 2), rnd.randint(1, 9), rnd.randint(1, 12), rnd.choice([1, 5, 32])
        W = rnd.choice([rnd.randint(1, 70), 4 * rnd.randint(1, 20)])
        x, dy = torch.randn(B, D, H, W, generator=g), torch.randn(B, D, H, W, generator=g)
        w1, b1 = torch.randn(K, 1, 3, 3, 3, generator=g) * 0.3, torch.randn(K, generator=g) * 0.1
        w2, b2, al = torch.randn(1, K, 3, 3, 3, generator=g) * 0.3, torch.randn(1, generator=g), torch.tensor([0.2])
        lv = [t.double().requires_grad_() for t in (x, w1, b1, w2, b2, al)]
        vol = lv[0].permute(0, 2, 3, 1).unsqueeze(1)
        y = F.conv3d(F.prelu(F.conv3d(vol, lv[1], lv[2], padding=1), lv[5]), lv[3], lv[4], padding=1)[:, 0].permute(0, 3, 1, 2)
        (y * dy.double()).sum().backward()
        outs = ops.conv3d_1k1_backward(x.cuda(), dy.cuda(), w1.cuda(), b1.cuda(), al.cuda(), w2.cuda())
        e = max(max(rel_err(o, r.grad)) for o, r in zip(outs, (lv[0], lv[1], lv[2], lv[3], lv[4], lv[5])))
        key = "conv3d stage" + (" (aligned rows)" if W % 4 == 0 else "")
        worst[key] = max(worst.get(key, (0, None)), (e, (B, D, H, W, K)))
    for k, (e, shape) in sorted(worst.items()):
        print(f"{k:28s} worst max-rel / L2-rel error {e:.2e} at {shape}", flush=True)
    assert all(e <= 1e-4 for e, _ in worst.values())


if __name__ == "__main__":
    main()
