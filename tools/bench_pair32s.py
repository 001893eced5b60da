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
T = 32000 * 32 // C      # the stage's length at batch 32 x 250 frames


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
def setopt(name, v):
    _lib.check(_lib.lib().sat_conv_set_option(name, v), "set_option")


for d in (1, 3, 5):
    row, outs = [], []
    for opt in (0, 8, 4):
        setopt(b"pair32s", int(opt > 0))
        setopt(b"pair32w", int(opt > 0))
        setopt(b"pair64w", int(opt > 0))
        setopt(b"pair32s_waves", opt)
        t_planes = timed(lambda: ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=xs, y_split=ys, y_split_slope=0.1, planes_residual=True, no_y=True))
        y1 = ys.clone()
        acc2 = acc.clone()
        t_f32 = timed(lambda: ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=xs, planes_residual=True, out=acc2))
        acc3 = acc.clone()
        ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=xs, planes_residual=True, out=acc3, accum=True)
        t_acc = timed(lambda: ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=xs, planes_residual=True, out=acc2, accum=True))
        row.append((t_planes, t_f32, t_acc))
        outs.append((y1, acc3))
    same = all(torch.equal(outs[1][i], outs[2][i]) for i in (0, 1))
    mb = 2 * B * C * T * 4 / 1e6
    names = ("general", "8 waves", "2 x 4 waves")
    print(f"C {C} k {k} dilation {d} ({mb:.0f} MB in + out); streaming forms bit-identical: {same}")
    for nm, (a, b_, c) in zip(names, row):
        print(f"    {nm:24s} planes -> planes {a:6.1f} us ({mb / a:.2f} TB/s)   planes -> f32 {b_:6.1f}   planes -> f32 accumulated {c:6.1f}")
