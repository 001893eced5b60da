import sys, torch
sys.path.insert(0, "/root/repo")
import satools_amd
from satools_amd import ops, packing, _lib
import torch.nn.functional as F
dev = "cuda"
torch.manual_seed(0)
for B, C, T, k, dil in ((2, 256, 333, 11, 5), (3, 256, 333, 11, 5), (3, 256, 1250, 11, 5), (4, 256, 333, 11, 5), (1, 256, 333, 11, 5), (3, 128, 700, 3, 5)):
    x = torch.randn(B, C, T, device=dev); w = torch.randn(C, C, k, device=dev) * (k * C) ** -0.5; b = torch.randn(C, device=dev); r = torch.randn(B, C, T, device=dev)
    wp = packing.pack_conv_weight_f16x3(w); xs = ops.act_split(x, 0.1); rs = ops.act_split(r, 0.1)
    ref = F.conv1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), padding=dil * (k - 1) // 2, dilation=dil) + r.double()
    for order in ("res_first", "plain_first"):
        if order == "plain_first":
            ys0 = ops.split_like(B, C, T, dev).zero_()
            ops.conv1d(x, wp, C, k, bias=b, dilation=dil, pad_left=dil * (k - 1) // 2, mode=1, x_split=xs, y_split=ys0, y_split_slope=0.1, no_y=True)
        ys = ops.split_like(B, C, T, dev).zero_()
        y = ops.conv1d(x, wp, C, k, bias=b, dilation=dil, pad_left=dil * (k - 1) // 2, mode=1, x_split=xs, y_split=ys, y_split_slope=0.1, res_split=rs, res_split_slope=0.1)
        e = (y.double() - ref).abs()
        bad = (e > 1e-3).nonzero()
        print(B, C, T, k, dil, order, "max err %.2e" % e.max().item(), "bad", len(bad), bad[:3].tolist(), bad[-2:].tolist(), _lib.lib().sat_last_dispatch_name().decode())
