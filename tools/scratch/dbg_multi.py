import sys, torch
sys.path.insert(0, "/root/repo")
import satools_amd
from satools_amd import ops, packing, _lib
dev = "cuda"
torch.manual_seed(0)
for B, C, T in ((4, 256, 1250), (32, 256, 1250), (3, 256, 333), (4, 128, 700)):
    ks, dils = (3, 7, 11), (1, 3, 5)
    x = torch.randn(B, C, T, device=dev); xs = ops.act_split(x, 0.1)
    ws = [packing.pack_conv_weight_f16x3(torch.randn(C, C, k, device=dev) * (k * C) ** -0.5) for k in ks]
    bs = [torch.randn(C, device=dev) for k in ks]
    def jobs(ys):
        return [(x, ws[j], C, k, dict(bias=bs[j], dilation=dils[j], pad_left=dils[j] * (k - 1) // 2, mode=1, x_split=xs, y_split_slope=0.1, y_split=ys[j], no_y=True)) for j, k in enumerate(ks)]
    for trial in range(3):
        ya = [ops.split_like(B, C, T, dev).zero_() for _ in ks]; yb = [ops.split_like(B, C, T, dev).zero_() for _ in ks]
        for (xx, w, c, k, kw) in jobs(ya):
            ops.conv1d(xx, w, c, k, **kw)
        ops.conv1d_multi(jobs(yb))
        torch.cuda.synchronize()
        for j in range(3):
            a, b = ops.unsplit(ya[j]), ops.unsplit(yb[j])
            d = (a - b).abs()
            bad = (d > 0).nonzero()
            print(B, C, T, "trial", trial, "job", j, "mismatches", len(bad), "max %.2e" % d.max().item(), bad[:4].tolist(), bad[-2:].tolist())
