"""the generator's split-f16 conv tile (conv_lean.hip) on the thick stages, batch 32: plain grid against the balanced grid
(sat_conv_set_option "lean_balance")"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import satools_amd
from satools_amd import ops, packing, _lib

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


shapes = [(256, 1250), (128, 5000), (64, 20000)] + [(256, int(t)) for t in sys.argv[1:]]
for C, T in shapes:
    x = torch.randn(B, C, T, device=dev)
    xs = ops.act_split(x, 0.1)
    ys = ops.split_like(B, C, T, dev)
    for k in (3, 7, 11):
        w = packing.pack_conv_weight_f16x3(torch.randn(C, C, k, device=dev) * (k * C) ** -0.5)
        b = torch.randn(C, device=dev)
        t = []
        for v in (0, 1):
            _lib.check(_lib.lib().sat_conv_set_option(b"lean_balance", v), "opt")
            t.append(timed(lambda: ops.conv1d(x, w, C, k, bias=b, dilation=5, pad_left=5 * (k - 1) // 2, mode=1, x_split=xs, y_split=ys, y_split_slope=0.1, no_y=True)))
        fl = 2 * B * C * C * k * T
        print(f"C {C:3d} T {T:5d} k {k:2d}: plain {t[0]:6.1f} us ({fl / t[0] / 1e6:5.0f} TF/s)   balanced {t[1]:6.1f} us ({fl / t[1] / 1e6:5.0f} TF/s)")
