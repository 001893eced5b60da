"""pair32s planes -> planes with the SAME buffers every launch against a rotation over NB buffer sets (the 256 MB memory-side cache
cannot hold what the previous launches moved): python tools/scratch/pair32s_cold.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from satools_amd import ops, packing
B, C, T, k, dev = 32, 32, 32000, 3, "cuda"
NB = 6
pk = packing.pack_conv_weight_f16x3
w1, w2 = pk(torch.randn(C, C, k, device=dev) * 0.6 / np.sqrt(C * k)), pk(torch.randn(C, C, k, device=dev) * 0.6 / np.sqrt(C * k))
b1, b2 = torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
x = torch.randn(B, C, T, device=dev)
xs = [ops.act_split(x, 0.1) for _ in range(NB)]
ys = [ops.split_like(B, C, T, dev) for _ in range(NB)]


def run(i, j, d):
    ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=xs[i], y_split=ys[j], y_split_slope=0.1, planes_residual=True, no_y=True)


def timed(f, n=24):
    for i in range(6):
        f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        f(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for d in (1, 3, 5):
    same = timed(lambda i: run(0, 0, d))
    rot = timed(lambda i: run(i % NB, i % NB, d))
    chain = timed(lambda i: run(i % 2, (i + 1) % 2, d) if False else ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=(xs[0], ys[0])[i % 2], y_split=(ys[0], xs[0])[i % 2],
                                                                                       y_split_slope=0.1, planes_residual=True, no_y=True))
    print(f"dilation {d}: same buffers {same:6.1f} us   rotation over {NB} sets {rot:6.1f} us   ping-pong (reads what the last launch wrote) {chain:6.1f} us")
