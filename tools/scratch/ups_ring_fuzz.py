"""edge lengths of the stride-4 upsampler on the ring against the 64 x 256 tile (planes out): max abs difference per case"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import satools_amd
from satools_amd import ops, packing, _lib
dev = "cuda"
torch.manual_seed(0)
worst = 0.0
for cin in (256, 128, 192, 64 * 5):
    cout, k, u = cin // 2, 8, 4
    if not packing.upsample_grouped_supported(cin, cout, k, u, 2):
        print("skip", cin); continue
    w = torch.randn(cin, cout, k, device=dev) * (2.0 / (cin * k)) ** 0.5
    b = torch.randn(cout, device=dev)
    wc, ks, pl = packing.convtranspose_as_phase_conv(w, u, 2)
    wg, _, _ = packing.convtranspose_as_phase_conv(w, u, 2, grouped=True)
    wpt, wpg = packing.pack_conv_weight_f16x3(wc, up=u), packing.pack_conv_weight_f16x3(wg, up=u)
    zt = packing.convtranspose_zero_taps(k, u, 2)
    for T in (1, 2, 3, 15, 16, 17, 63, 64, 65, 159, 160, 161, 162, 319, 320, 321, 480, 481, 1250):
        for B in (1, 3):
            x = torch.randn(B, cin, T, device=dev)
            xs = ops.act_split(x, 0.1)
            a = torch.full((B, cout // 16, 2, 2, T * u, 8), 9.0, dtype=torch.float16, device=dev)
            g = torch.full((B, cout // 16, 2, 2, T * u, 8), 9.0, dtype=torch.float16, device=dev)
            ops.conv1d(x, wpt, cout, ks, bias=b, pad_left=pl, up=u, mode=1, x_split=xs, y_split=a, y_split_slope=0.1, no_y=True)
            ops.conv1d(x, wpg, cout, ks, bias=b, pad_left=pl, up=u, mode=1, x_split=xs, y_split=g, y_split_slope=0.1, no_y=True, up_grouped=True, up_zero_taps=zt)
            d = (ops.unsplit(a) - ops.unsplit(g)).abs().max().item()
            worst = max(worst, d)
            if d > 4e-6 or not torch.isfinite(ops.unsplit(g)).all():
                print("MISMATCH", cin, T, B, d)
print("worst abs difference", worst)
