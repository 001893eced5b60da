"""the thin upsamplers (64 -> 32 at T 20000, 32 -> 16 at T 40000, batch 32): streaming kernel (csrc/ups2.hip) against the polyphase conv tile"""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import satools_amd
from satools_amd import ops, packing

B, dev = 32, "cuda"


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin, T in ((64, 20000), (32, 40000)):
    cout = cin // 2
    x = torch.randn(B, cin, T, device=dev)
    w = torch.randn(cin, cout, 4, device=dev) * 0.8 / np.sqrt(cin * 2)
    b = torch.randn(cout, device=dev) * 0.1
    wc, kp, pl = packing.convtranspose_as_phase_conv(w, 2, 1)
    wp = packing.pack_conv_weight_f16x3(wc, up=2)
    xs = ops.act_split(x, 0.1)
    ys = ops.split_like(B, cout, 2 * T, dev)
    t_old = timed(lambda: ops.conv1d(x, wp, cout, kp, bias=b, pad_left=pl, up=2, mode=1, x_split=xs, y_split=ys, y_split_slope=0.1, no_y=True))
    t_new = timed(lambda: ops.upsample2(xs, wp, b, B, cin, T, y_split=ys))
    mb = B * (cin * T + cout * 2 * T) * 4 / 1e6
    print(f"C {cin} -> {cout}, T {T}: polyphase conv tile {t_old:7.1f} us   streaming kernel {t_new:7.1f} us   ({mb:.0f} MB: {mb / t_new:.2f} TB/s)")
