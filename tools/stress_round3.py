#!/usr/bin/env python3
"""Random shapes through round 3's other new kernels against float64 torch: the composed first layer (with / without its fused first
map, both output layouts), the split Conv3d 1 -> K -> 1, the split weight gradient (3x3 / 1x1), the tape form of the layer.  GPU box."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwfa_amd import ops
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 5); torch.manual_seed(9)
F = torch.nn.functional
ops.set_precision("split_bf16")
worst = {}
def rel(a, b): return float((a.cpu().double() - b).abs().max() / b.abs().max())
def note(k, e, bound, ctx):
    worst[k] = max(worst.get(k, 0.0), e)
    assert e < bound, (k, e, ctx)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    B = random.choice([1, 2]); H = random.randint(1, 70); W = random.randint(1, 100)
    # composed first layer
    cin = random.randint(1, 31)
    u = torch.randn(B, cin, H, W)
    w0, b0 = torch.randn(64, cin, 1, 1) / cin ** 0.5, torch.randn(64) * 0.3
    w3, b3 = torch.randn(64, 64, 3, 3) / 24, torch.randn(64) * 0.1
    w1, b1 = torch.randn(64, 64, 1, 1) / 8, torch.randn(64) * 0.1
    x64 = F.conv2d(u.double(), w0.double(), b0.double())
    ref = F.elu(F.conv2d(F.elu(F.conv2d(x64, w3.double(), b3.double(), padding=1)), w1.double(), b1.double()) + x64)
    pc = ops.pack_first_layer_weight(w0.cuda(), b0.cuda(), w3.cuda(), w1.cuda(), short=False)
    pcx = ops.pack_first_layer_weight(w0.cuda(), b0.cuda(), w3.cuda(), w1.cuda())          # short form where cin + 1 <= 16
    u1 = ops.with_ones(u.cuda())
    note("first layer, fused map", rel(ops.subnet_layer_first(u1, None, pc, b3.cuda(), b1.cuda()), ref), 6e-6, (B, cin, H, W))
    note("first layer, fused map, default image", rel(ops.subnet_layer_first(u1, None, pcx, b3.cuda(), b1.cuda(), layout=random.choice([0, 2]) * 0), ref), 6e-6, (B, cin, H, W))
    x = ops.conv2d(u.cuda(), ops.pack_conv_weight(w0.cuda()), bias=b0.cuda())
    note("first layer, map from memory", rel(ops.subnet_layer_first(u1, x, pc, b3.cuda(), b1.cuda()), ref), 6e-6, (B, cin, H, W))
    # tape form
    xx = torch.randn(B, 64, H, W)
    href = F.elu(F.conv2d(xx.double(), w3.double(), b3.double(), padding=1))
    yref = F.elu(F.conv2d(href, w1.double(), b1.double()) + xx.double())
    y, h = ops.subnet_layer(xx.cuda(), ops.pack_split_layer_weight(w3.cuda(), w1.cuda()), b3.cuda(), None, b1.cuda(), want_hidden=True)
    note("tape layer y", rel(y, yref), 5e-6, (B, H, W)); note("tape layer h", rel(h, href), 5e-6, (B, H, W))
    # split weight gradient
    ks = random.choice([1, 3]); ci = random.choice([3, 29, 64, 70, 130]); co = random.choice([6, 48, 64, 96, 130])
    W4 = max(4, W // 4 * 4)
    xa = torch.randn(B, ci, H, W4); dy = torch.randn(B, co, H, W4)
    wz = torch.zeros(co, ci, ks, ks, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xa.double(), wz, padding=ks // 2) * dy.double()).sum().backward()
    gw, gb = ops.conv2d_wgrad(xa.cuda(), dy.cuda(), ks, want_bias=True)
    note(f"wgrad {ks}x{ks}", rel(gw, wz.grad), 5e-6, (B, ci, co, H, W4)); note("wgrad bias", rel(gb, dy.double().sum((0, 2, 3))), 1e-4, (B, co, H, W4))
    # split Conv3d
    D = random.randint(1, 20); K = random.choice([4, 8, 32])
    v = torch.randn(B, D, H, W); k1, c1 = torch.randn(K, 1, 3, 3, 3) / 5, torch.randn(K) * 0.1
    k2, c2 = torch.randn(1, K, 3, 3, 3) / (27 * K) ** 0.5, torch.randn(1) * 0.1
    al = torch.tensor([0.25])
    hid = F.conv3d(v.double().permute(0, 2, 3, 1).unsqueeze(1), k1.double(), c1.double(), padding=1)    # the reference convolves over (H, W, depth): networks.py:236-241
    hid = torch.where(hid > 0, hid, 0.25 * hid)
    r3 = F.conv3d(hid, k2.double(), c2.double(), padding=1)[:, 0].permute(0, 3, 1, 2)
    note("conv3d 1-K-1", rel(ops.conv3d_1k1(v.cuda(), k1.cuda(), c1.cuda(), al.cuda(), k2.cuda(), c2.cuda()), r3), 5e-6, (B, D, H, W, K))
ops.set_precision("fp32")
print({k: "%.2e" % v for k, v in worst.items()})
