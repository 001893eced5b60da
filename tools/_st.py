import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satools_amd
from satools_amd import ops, packing, _lib
B, T = 32, 249
cin, cout = 4096, 1024
x = torch.randn(B, cin, T, device="cuda"); w = torch.randn(cout, cin, 1, device="cuda") / cin ** 0.5
wp = packing.pack_conv_weight_f16x3(w); xs = ops.act_split(x, 1.0); bias = torch.randn(cout, device="cuda")
for _ in range(5): ops.conv1d(x, wp, cout, 1, bias=bias, mode=1, x_split=xs)
torch.cuda.synchronize()
for d in (-74.0, -70.0, -71.0, -72.0, -73.0):
    for _ in range(2):
        ops.conv1d(x, wp, cout, 1, bias=bias, mode=1, x_split=xs, accum_div=d)
    torch.cuda.synchronize()
print("-74 full; -70 no fragment reads; -71 no DMA issue; -72 no barrier; -73 neither reads nor DMA (MFMAs + barrier only)")
