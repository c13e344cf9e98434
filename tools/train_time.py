#!/usr/bin/env python3
"""Time the training step (forward + backward + gradient exchange stub) of the finest CAT flow step at BASELINE.json
configs[3]'s per-rank shape (512x512x96 volumes) and the weight-gradient kernel on its own (GPU box):
    python tools/train_time.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cwfa_amd import CWFA, ops, training  # noqa: E402


def ev_time(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    torch.manual_seed(0)
    for (cin, cout, ks) in [(64, 64, 3), (64, 64, 1), (64, 96, 3), (96, 64, 1)]:
        x = torch.randn(B, cin, 512, 512, device="cuda")
        dy = torch.randn(B, cout, 512, 512, device="cuda")
        ms = ev_time(lambda: ops.conv2d_wgrad(x, dy, ks))
        fl = 2.0 * B * cin * cout * ks * ks * 512 * 512
        print(f"wgrad {cin}->{cout} k{ks} @512 B{B}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TF/s", flush=True)
    conv_inn, cond_nets = CWFA.build_networks(96, 512, 2, with_lrnn=False, device="cuda")
    g = conv_inn[0].train()
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, 96, 512, 512, generator=gen).cuda()
    c = [torch.randn(B, 48, 512, 512, generator=gen).cuda(), 0.1 * torch.randn(B, 48, 512, 512, generator=gen).cuda()]
    params = [p for p in g.parameters() if p.requires_grad]

    def step():
        for p in params:
            p.grad = None
        training.nll_backward(g, x, c)

    cn = cond_nets[0].eval()
    views = torch.randn(B, 29, 512, 512, generator=gen).cuda()
    low = torch.randn(B, 48, 512, 512, generator=gen).cuda()
    cparams = [p for p in cn.parameters() if p.requires_grad]

    def full_step():                 # the reference's default training step of a flow step, condition net included
        for p in params + cparams:
            p.grad = None
        omega, tape = training.cond_forward_train(cn, views)
        out = training.step_backward(g, x, [omega, c[1]], low=low, want_cond_grads=True)
        training.cond_backward(tape, out["cond_grads"][0])

    def cond_only():
        for p in cparams:
            p.grad = None
        omega, tape = training.cond_forward_train(cn, views)
        training.cond_backward(tape, c[0])

    ms_c = ev_time(cond_only, reps=3, warm=2)
    ms_f = ev_time(full_step, reps=3, warm=2)
    print(f"condition net (29 views -> 48 channels, Conv3d K=32) forward + backward B{B}: {ms_c:.2f} ms; full default training step "
          f"(inverse + forward + backward of the flow step and its condition net): {ms_f:.2f} ms = {B/ms_f*1e3:.2f} volumes/s", flush=True)
    with torch.no_grad():
        fwd = ev_time(lambda: CWFA.nll_step(g, x, c), reps=5)
    ms = ev_time(step, reps=5)
    n_par = sum(p.numel() for p in params if p.grad is not None)
    print(f"flow step 0 (48 flow channels, 5 CAT blocks) @512x512x96 B{B}: inference forward {fwd:.2f} ms, training step "
          f"(forward with tape + backward) {ms:.2f} ms = {B/ms*1e3:.2f} volumes/s; {n_par/1e6:.2f} M parameters with gradients; "
          f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)


if __name__ == "__main__":
    main()


def lrnn_time():
    """LRNN without the mean-volume branch (1x1 conv + UNet 256/512/1024) at 512x512: train-mode forward + backward, B = 1."""
    from cwfa_amd import networks as N
    torch.manual_seed(0)
    enc = N.Encoder(29, 6, 5, 64, True).cuda().train()
    lr = enc.net
    lr.deconv[1].drop_out = 0
    gen = torch.Generator().manual_seed(5)
    views = torch.randn(1, 29, 512, 512, generator=gen).cuda()
    gt = torch.randn(1, 6, 512, 512, generator=gen).cuda()
    params = [p for p in lr.parameters() if p.requires_grad]

    for cn in lr.conv3d:
        cn.drop_prob = 0.0
    mean = (0.1 * torch.randn(1, 6, 512, 512, generator=gen)).cuda()

    def step():
        for p in params:
            p.grad = None
        training.lrnn_step_backward(enc, views, mean, gt)

    with torch.no_grad():
        fwd = ev_time(lambda: lr(views, mean), reps=3, warm=2)
    ms = ev_time(step, reps=3, warm=2)
    n_par = sum(p.numel() for p in params if p.grad is not None)
    print(f"LRNN (UNet 256/512/1024 + mean-volume branch) @512x512 B1: inference forward {fwd:.2f} ms, training step (forward with tape + backward, L2 loss) "
          f"{ms:.2f} ms; {n_par/1e6:.1f} M parameters with gradients; peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "lrnn":
    lrnn_time()
