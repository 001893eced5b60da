"""micro-benchmark of one ResBlock1 step (conv1 -> conv2 + residual) on the generator's stage shapes with
split-plane activations: the fused pair kernel vs two launches of the planes conv kernel
  python tools/bench_pair.py [f8] [stage indices]"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import satools_amd
from satools_amd import ops, packing

B = 32
SHAPES = [(256, 1250), (128, 5000), (64, 20000), (32, 40000), (16, 80000)]
F8 = 'f8' in sys.argv
only = [int(a) for a in sys.argv[1:] if a.isdigit()] or [3, 4]
dev = "cuda"


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


for si in only:
    C, T = SHAPES[si]
    x = torch.randn(B, C, T, device=dev)
    out = torch.empty(B, C, T, device=dev)
    fmt = 1 if F8 else 0
    mode = 2 if F8 else 1
    pk = packing.pack_conv_weight_f16f8 if F8 else packing.pack_conv_weight_f16x3
    xs = ops.act_split(x, 0.1, fmt=fmt)
    t1s = ops.split_like(B, C, T, dev)
    ys = ops.split_like(B, C, T, dev)
    for k in (3, 7, 11):
        for d in (1, 5):
            w1, w2 = pk(torch.randn(C, C, k, device=dev) * 0.05), pk(torch.randn(C, C, k, device=dev) * 0.05)
            b1, b2 = torch.randn(C, device=dev), torch.randn(C, device=dev)

            def unfused():
                ops.conv1d(x, w1, C, k, bias=b1, dilation=d, pad_left=(k * d - d) // 2, mode=mode, x_split=xs, y_split=t1s,
                           y_split_slope=0.1, no_y=True, out=out)
                ops.conv1d(x, w2, C, k, bias=b2, pad_left=(k - 1) // 2, res=x, mode=mode, x_split=t1s, y_split=ys,
                           y_split_slope=0.1, out=out)

            tu = timed(unfused)
            tf = float('nan')
            if C <= 32 and not F8:
                tf = timed(lambda: ops.resblock_pair(x, w1, b1, w2, b2, k, d, out=out, x_split=xs, y_split=ys, y_split_slope=0.1))
            flop = 2 * 2.0 * B * C * C * k * T
            print(f"stage{si} C={C:4d} T={T:6d} k={k:2d} d={d}: two convs {tu:7.1f} us ({flop / tu / 1e6:6.1f} TFLOP/s)   fused pair {tf:7.1f} us")
