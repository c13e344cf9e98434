#!/usr/bin/env python3
"""fp32-MFMA 1x1 kernels vs the opt-in split-bf16 GEMM (GPU box): time and error on the UNet's up-convolutions and 1x1s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwfa_amd import ops

def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for (cin, co, H, tr) in [(1024, 512, 128, True), (512, 256, 256, True), (256, 256, 512, False), (64, 128, 512, False)]:
    x = torch.randn(1, cin, H, H, device="cuda")
    w = (torch.randn(cin, co, 2, 2, device="cuda") if tr else torch.randn(co, cin, 1, 1, device="cuda")) / cin ** 0.5
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    cout = 4 * co if tr else co
    fl = 2.0 * cin * cout * H * H
    res = {}
    for split in (0, 1):
        ops.set_option("split_bf16", split)
        pc = ops.pack_conv_weight(w, transposed=tr)
        y = ops.conv2d(x, pc, in_scale=sc, in_shift=sh)
        ms = t(lambda: ops.conv2d(x, pc, in_scale=sc, in_shift=sh))
        res[split] = (ms, y)
    ops.set_option("split_bf16", 0)
    d = float((res[1][1] - res[0][1]).abs().max() / res[0][1].abs().max())
    print(f"{'convT' if tr else '1x1'} {cin}->{cout} @{H}: fp32 {res[0][0]:.3f} ms ({fl/res[0][0]/1e9:.0f} TF)  split {res[1][0]:.3f} ms ({fl/res[1][0]/1e9:.0f} TF)  |split-fp32|/max {d:.1e}", flush=True)

alpha = torch.tensor([0.25], device="cuda")
for (cin, cout, H) in [(256, 256, 512), (512, 512, 256), (1024, 1024, 128)]:
    x = torch.randn(1, cin, H, H, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    fl = 2.0 * cin * cout * 9 * H * H
    res = {}
    for split in (0, 2):
        ops.set_option("split_bf16", split)
        pc = ops.pack_conv_weight(w)
        f = lambda: ops.conv2d(x, pc, act="prelu", prelu_alpha=alpha, in_scale=sc, in_shift=sh)
        y = f()
        res[split] = (t(f), y)
    ops.set_option("split_bf16", 0)
    d = float((res[2][1] - res[0][1]).abs().max() / res[0][1].abs().max())
    print(f"3x3 {cin}->{cout} @{H} (BN-on-load, PReLU): fp32 Winograd {res[0][0]:.3f} ms ({fl/res[0][0]/1e9:.0f} TF)  split {res[2][0]:.3f} ms ({fl/res[2][0]/1e9:.0f} TF)  |diff|/max {d:.1e}", flush=True)

x = torch.randn(1, 64, 512, 512, device="cuda")
w3, b3 = torch.randn(64, 64, 3, 3, device="cuda") / 24, torch.randn(64, device="cuda") * 0.1
w1, b1 = torch.randn(64, 64, 1, 1, device="cuda") / 8, torch.randn(64, device="cuda") * 0.1
pn = ops.pack_1x1_panel(w1)
pa, pb = ops.pack_conv_weight(w3), ops.pack_split_layer_weight(w3, w1)
ya, yb = ops.subnet_layer(x, pa, b3, pn, b1), ops.subnet_layer(x, pb, b3, pn, b1)
ta, tb = t(lambda: ops.subnet_layer(x, pa, b3, pn, b1), 20), t(lambda: ops.subnet_layer(x, pb, b3, pn, b1), 20)
fl = 2.0 * 64 * 64 * 10 * 512 * 512
print(f"fused layer 64ch @512: fp32 Winograd {1e3*ta:.1f} us ({fl/ta/1e9:.0f} TF)  split {1e3*tb:.1f} us ({fl/tb/1e9:.0f} TF)  |diff|/max {float((ya-yb).abs().max()/ya.abs().max()):.1e}", flush=True)
