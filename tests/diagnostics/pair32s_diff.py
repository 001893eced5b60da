"""where the streaming 3-tap step (pair32s.hip) and the general fused step disagree: error per position block and per channel"""
import sys, os
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import satools_amd
from satools_amd import ops, packing, _lib

B, C, k, dev = 1, 32, 3, "cuda"
T, d = int(sys.argv[1]) if len(sys.argv) > 1 else 449, int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = torch.Generator().manual_seed(1)
x = torch.randn(B, C, T, generator=g).to(dev)
w1f, w2f = torch.randn(C, C, k, generator=g) * 0.06, torch.randn(C, C, k, generator=g) * 0.06
b1, b2 = (torch.randn(C, generator=g) * 0.1).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
w1, w2 = packing.pack_conv_weight_f16x3(w1f.to(dev)), packing.pack_conv_weight_f16x3(w2f.to(dev))
xs = ops.act_split(x, 0.1)
out = {}
for opt in (0, 1):
    _lib.check(_lib.lib().sat_conv_set_option(b"pair32s", opt), "opt")
    y = torch.zeros(B, C, T, device=dev)
    ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=xs, planes_residual=True, out=y)
    out[opt] = y.cpu().double()
    print(opt, _lib.lib().sat_last_dispatch_name().decode())
xd = x.double().cpu()
t1 = F.conv1d(F.leaky_relu(xd, 0.1), w1f.double(), b1.double().cpu(), dilation=d, padding=d)
y64 = xd + F.conv1d(F.leaky_relu(t1, 0.1), w2f.double(), b2.double().cpu(), padding=1)
for opt in (0, 1):
    e = (out[opt] - y64).abs()[0]
    print("opt", opt, "max err", e.max().item())
    print("  per channel:", " ".join(f"{v:.1e}" for v in e.max(1).values.tolist()))
    print("  per 16 positions:", " ".join(f"{e[:, i:i + 16].max().item():.0e}" for i in range(0, T, 16)))
