#!/usr/bin/env python3
"""Cost of the load-side prologue on the UNet 3x3 convs (GPU box): plain vs BN-affine vs affine + skip add."""
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

alpha = torch.tensor([0.25], device="cuda")
for two_d in (1, 0):
  ops.set_option("winograd_2d", two_d)
  print("winograd_2d =", two_d)
  for (cin, cout, H) in [(256, 256, 512), (512, 512, 256), (1024, 1024, 128)]:
      x = torch.randn(1, cin, H, H, device="cuda"); a = torch.randn(1, cin, H, H, device="cuda")
      pc = ops.pack_conv_weight(torch.randn(cout, cin, 3, 3, device="cuda") * 0.02)
      sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
      out = torch.empty(1, cout, H, H, device="cuda")
      fl = 2.0 * cin * cout * 9 * H * H
      r = {"plain": t(lambda: ops.conv2d(x, pc, act="prelu", prelu_alpha=alpha, out=out)),
           "affine": t(lambda: ops.conv2d(x, pc, act="prelu", prelu_alpha=alpha, in_scale=sc, in_shift=sh, out=out)),
           "affine+add": t(lambda: ops.conv2d(x, pc, act="prelu", prelu_alpha=alpha, in_scale=sc, in_shift=sh, in_add=a, out=out))}
      print(cin, cout, H, {k: (round(v, 3), round(fl / v / 1e9, 1)) for k, v in r.items()}, flush=True)
