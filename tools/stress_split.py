import sys; sys.path.insert(0, "/root/repo")
import torch, random
from cwfa_amd import ops
random.seed(1); torch.manual_seed(1)
F = torch.nn.functional
worst = 0
for it in range(30):
    ks = random.choice([1, 3])
    B = random.choice([1, 2, 3]); Cin = random.choice([5, 16, 17, 48, 100, 130]); Cout = random.choice([192, 200, 256, 300, 513])
    H = random.randint(1, 40); W = random.randint(1, 90)
    x = torch.randn(B, Cin, H, W); w = torch.randn(Cout, Cin, ks, ks) / (Cin * ks * ks) ** 0.5; b = torch.randn(Cout)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=ks // 2)
    ops.set_option("split_bf16", random.choice([2, 3]))
    pc = ops.pack_conv_weight(w.cuda()); assert pc.split
    y = ops.conv2d(x.cuda(), pc, bias=b.cuda())
    ops.set_option("split_bf16", 0)
    e = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
    worst = max(worst, e)
    assert e < 5e-6, (ks, B, Cin, Cout, H, W, e)
print("30 random split convs ok, worst max-rel", worst)
