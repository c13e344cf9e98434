#!/usr/bin/env python3
"""Random 3x3 convolutions on the split-bf16 kernel over every tiling (16 / 32 / 48 / 64 / 96 / 128 / 256 channels per block), odd and
even chunk counts (the skipped trailing steps), epilogues (bias, PReLU scalar / per channel, residual + second activation), blocked
inputs, statistics epilogue -- against float64 torch.  GPU box."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwfa_amd import ops
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 7); torch.manual_seed(3)
F = torch.nn.functional
ops.set_precision("split_bf16")
ops.SPLIT_3X3_NARROW_MAX = 48
worst = 0.0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for it in range(N):
    B = random.choice([1, 2])
    Cin = random.choice([3, 6, 8, 15, 16, 17, 29, 32, 33, 48, 64, 80])
    Cout = random.choice([5, 12, 16, 17, 24, 32, 33, 48, 50, 64, 65, 96, 97, 128, 130, 180, 256, 260])
    H = random.randint(1, 50); W = random.randint(1, 70)
    if Cout <= 48 and Cin < 29:
        Cin = random.choice([29, 32, 48, 64])            # (smaller banks stay on the fp32 Winograd kernel)
    x = torch.randn(B, Cin, H, W); w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5; b = torch.randn(Cout)
    mode = random.choice(["bias", "prelu", "prelu_pc", "res_prelu", "res"])
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    kw = {}
    if mode == "prelu":
        a = torch.tensor([0.2]); ref = torch.where(ref > 0, ref, 0.2 * ref); kw = dict(act="prelu", prelu_alpha=a.cuda())
    elif mode == "prelu_pc":
        a = torch.rand(Cout) + 0.1; a[::3] = 1.0
        ref = torch.where(ref > 0, ref, ref * a.double().view(1, -1, 1, 1)); kw = dict(act="prelu", prelu_alpha=a.cuda())
    elif mode in ("res_prelu", "res"):
        r = torch.randn(B, Cout, H, W); ref = ref + r.double(); kw = dict(residual=r.cuda())
        if mode == "res_prelu":
            a = torch.tensor([0.3]); ref = torch.where(ref > 0, ref, 0.3 * ref); kw.update(act2="prelu", prelu_alpha=a.cuda())
    pc = ops.pack_conv_weight(w.cuda())
    assert pc.split, (Cin, Cout)
    stats = None
    if mode in ("bias", "prelu") and random.random() < 0.5 and ops.conv_writes_stats(pc, kw.get("act"), None, None, False):
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device="cuda"); kw["out_stats"] = stats
    y = ops.conv2d(x.cuda(), pc, bias=b.cuda(), **kw)
    e = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
    worst = max(worst, e)
    assert e < 5e-6, (it, mode, B, Cin, Cout, H, W, e)
    if stats is not None:
        s1 = ref.sum((0, 2, 3)); s2 = (ref * ref).sum((0, 2, 3))
        got = stats.cpu().view(Cout, 2)
        assert float((got[:, 0] - s1).abs().max() / s1.abs().max().clamp_min(1e-9)) < 1e-4 and float((got[:, 1] - s2).abs().max() / s2.abs().max()) < 1e-4, (it, "stats", Cin, Cout, H, W)
ops.set_precision("fp32")
print(N, "random split 3x3 convolutions over all tilings ok, worst max-rel", worst)
