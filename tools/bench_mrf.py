"""micro-benchmark of the fused MRF block (csrc/mrf.hip) against the launches it replaces, on the generator's thin stage
shapes (B = 32):  python tools/bench_mrf.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import satools_amd
from satools_amd import ops, packing

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda"


def timed(f, n=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for C, T in [(16, 80000)]:
    x = torch.randn(B, C, T, device=dev)
    xs = ops.act_split(x, 0.1)
    pk = packing.pack_conv_weight_f16x3
    branches = []
    for k in (3, 7, 11):
        steps = []
        for i in range(3):
            steps.append((pk(torch.randn(C, C, k, device=dev) * 0.7 / np.sqrt(C * k)), torch.randn(C, device=dev) * 0.1,
                          pk(torch.randn(C, C, k, device=dev) * 0.7 / np.sqrt(C * k)), torch.randn(C, device=dev) * 0.1))
        branches.append((k, steps))
    acc = torch.empty(B, C, T, device=dev)
    bufs = [ops.split_like(B, C, T, dev) for _ in range(2)]

    def unfused(sel=branches):
        nb = len(sel)
        for j, (k, steps) in enumerate(sel):
            cur = xs
            for i, (w1, b1, w2, b2) in enumerate(steps):
                if i < 2:
                    ops.resblock_pair(x, w1, b1, w2, b2, k, 2 * i + 1, x_split=cur, y_split=bufs[i], y_split_slope=0.1,
                                      planes_residual=True, no_y=True, out=acc)
                    cur = bufs[i]
                else:
                    ops.resblock_pair(x, w1, b1, w2, b2, k, 2 * i + 1, x_split=cur, planes_residual=True, out=acc, accum=j > 0,
                                      accum_div=3.0 if j == nb - 1 else 0.0)

    out = torch.empty(B, C, T, device=dev)
    flop = sum(2 * 2.0 * B * C * C * k * T * 3 for k in (3, 7, 11))
    tu = timed(unfused)
    tf = timed(lambda: ops.resblock_mrf(xs, B, C, T, branches, out=out, out_div=3.0))
    tp = timed(lambda: ops.resblock_mrf(xs, B, C, T, branches, out=out, out_div=3.0, residual_from_planes=True))
    print(f"C={C} T={T} B={B}: nine launches {tu:8.1f} us ({flop / tu / 1e6:6.1f} TFLOP/s)   one launch {tf:8.1f} us ({flop / tf / 1e6:6.1f} TFLOP/s)"
          f"   residual from planes {tp:8.1f} us")
    for k, steps in branches:
        f1 = 2 * 2.0 * B * C * C * k * T * 3
        tu1 = timed(lambda: unfused([(k, steps)]))
        tf1 = timed(lambda: ops.resblock_mrf(xs, B, C, T, [(k, steps)], out=out))
        print(f"   branch k={k:2d}: three launches {tu1:8.1f} us   one launch {tf1:8.1f} us ({f1 / tf1 / 1e6:6.1f} TFLOP/s)")
