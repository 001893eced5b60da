"""the 3-tap ResBlock1 step of the 32-channel stage (T 40000, batch 32): streaming kernel (csrc/pair32s.hip) against the general fused step"""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import satools_amd
from satools_amd import ops, packing, _lib

B, dev = 32, "cuda"
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
C = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = 40000 * 32 // C


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


x = torch.randn(B, C, T, device=dev)
pk = packing.pack_conv_weight_f16x3
w1, w2 = pk(torch.randn(C, C, k, device=dev) * 0.6 / np.sqrt(C * k)), pk(torch.randn(C, C, k, device=dev) * 0.6 / np.sqrt(C * k))
b1, b2 = torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
xs = ops.act_split(x, 0.1)
ys = ops.split_like(B, C, T, dev)
acc = torch.randn(B, C, T, device=dev)
for d in (1, 3, 5):
    row = []
    for opt in (0, 8, 4):
        _lib.check(_lib.lib().sat_conv_set_option(b"pair32s", int(opt > 0)), "set_option")
        _lib.check(_lib.lib().sat_conv_set_option(b"pair32w", int(opt > 0)), "set_option")
        _lib.check(_lib.lib().sat_conv_set_option(b"pair64w", int(opt > 0)), "set_option")
        _lib.check(_lib.lib().sat_conv_set_option(b"pair32s_waves", opt), "set_option")
        t_planes = timed(lambda: ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=xs, y_split=ys, y_split_slope=0.1, planes_residual=True, no_y=True))
        t_f32 = timed(lambda: ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=xs, planes_residual=True, out=acc))
        row.append((t_planes, t_f32))
    mb = 2 * B * C * T * 4 / 1e6
    print(f"C {C} k {k} dilation {d}: planes -> planes: general {row[0][0]:6.1f}, 8 waves {row[1][0]:6.1f}, 2 x 4 waves {row[2][0]:6.1f} us ({mb / row[2][0]:.2f} TB/s)   "
          f"planes -> f32: {row[0][1]:6.1f}, {row[1][1]:6.1f}, {row[2][1]:6.1f} us")
