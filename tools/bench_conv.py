"""micro-benchmark of the fused conv kernel on the generator's layer shapes (run on the GPU box)"""
import sys
import time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import satools_amd
from satools_amd import ops, packing

B = 32
SHAPES = [  # (C, T) per stage
    (256, 1250), (128, 5000), (64, 20000), (32, 40000), (16, 80000)]
MODE = 1 if ('f16' in sys.argv or 'split' in sys.argv) else 0
SPLIT = 'split' in sys.argv      # split-plane input and output (conv2 of a pair: residual + both outputs)
only = [int(a) for a in sys.argv[1:] if a.isdigit()] or list(range(5))
dev = "cuda"
rows = []
for si in only:
    C, T = SHAPES[si]
    x = torch.randn(B, C, T, device=dev)
    res = torch.randn(B, C, T, device=dev)
    out = torch.empty(B, C, T, device=dev)
    for k in (3, 7, 11):
        for d in (1, 5):
            w = (packing.pack_conv_weight_f16x3 if MODE else packing.pack_conv_weight)(torch.randn(C, C, k, device=dev) * 0.05)
            b = torch.randn(C, device=dev)
            pl = (k * d - d) // 2
            f = lambda: ops.conv1d(x, w, C, k, bias=b, dilation=d, pad_left=pl, in_lrelu=0.1, res=res, out=out, mode=MODE)
            if SPLIT:
                xs = ops.act_split(x, 0.1)
                ys = ops.split_like(B, C, T, dev)
                f = lambda: ops.conv1d(x, w, C, k, bias=b, dilation=d, pad_left=pl, res=res, out=out, mode=1,
                                       x_split=xs, y_split=ys, y_split_slope=0.1)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                f()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            flop = 2.0 * B * C * C * k * T
            gb = 3.0 * B * C * T * 4 / 1e9
            print(f"stage{si} C={C:4d} T={T:6d} k={k:2d} d={d}: {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s  ({flop/us/1e6/157.3*100:5.1f}% of f32 MFMA)  {gb / (us*1e-6) / 1e3:6.2f} TB/s")
