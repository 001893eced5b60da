"""conv_ring16.hip experiment variants (option "convring" = 1 + 16 * VAR) on the two thick generator stages."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satools_amd  # noqa: E402,F401
from satools_amd import ops, packing, _lib  # noqa: E402

B, dev = 32, "cuda"


def timed(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def opt(v):
    _lib.check(_lib.lib().sat_conv_set_option(b"convring", v), "opt")


variants = [int(v) for v in os.environ.get("VARS", "0,1,2,3,4,9,11").split(",")]
for C, T in ((256, 1250), (128, 5000)):
    x = torch.randn(B, C, T, device=dev)
    xs = ops.act_split(x, 0.1)
    ys = ops.split_like(B, C, T, dev)
    for k in (3, 7, 11):
        w = packing.pack_conv_weight_f16x3(torch.randn(C, C, k, device=dev) * (k * C) ** -0.5)
        b = torch.randn(C, device=dev)
        f = lambda: ops.conv1d(x, w, C, k, bias=b, dilation=1, pad_left=(k - 1) // 2, mode=1, x_split=xs, y_split=ys, y_split_slope=0.1, no_y=True)
        f2 = lambda: ops.conv1d(x, w, C, k, bias=b, dilation=1, pad_left=(k - 1) // 2, mode=1, x_split=xs, y_split=ys, y_split_slope=0.1, no_y=True, res_split=xs, res_split_slope=0.1)
        opt(0)
        line = f"C {C:3d} k {k:2d}: lean {timed(f):6.1f} |"
        for rep in range(2):
            for v in variants:
                opt(1 + 16 * v)
                line += f" v{v} {timed(f):6.1f}/{timed(f2):6.1f}"
            line += " |"
        print(line, flush=True)
opt(0)
