"""time the 1x1 conv on split planes (the Linear layers of the wav2vec2 tag, TDNNF linearA/B) with the 128 x 128
GEMM kernel (k1_gemm = 1) and the LDS-DMA ring GEMM (k1_gemm = 2): B=32 utterances x 249 frames"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satools_amd
from satools_amd import ops, packing, _lib

B, T = 32, 249


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


for cin, cout in ((1024, 1024), (1024, 3072), (1024, 4096), (4096, 1024), (128, 1024), (1024, 3280), (512, 1024)):
    x = torch.randn(B, cin, T, device="cuda")
    w = torch.randn(cout, cin, 1, device="cuda") / cin ** 0.5
    wp = packing.pack_conv_weight_f16x3(w)
    xs = ops.act_split(x, 1.0)
    bias = torch.randn(cout, device="cuda")
    ref = None
    for opt in (1, 2):
        _lib.check(_lib.lib().sat_conv_set_option(b"k1_gemm", opt), "set_option")
        y = ops.conv1d(x, wp, cout, 1, bias=bias, mode=1, x_split=xs)
        us = timed(lambda: ops.conv1d(x, wp, cout, 1, bias=bias, mode=1, x_split=xs))
        if ref is None:
            ref = y
        fl = 2.0 * cin * cout * B * T
        print(f"{cin:5d} -> {cout:5d}  k1_gemm={opt}  {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s useful   max diff vs k1_gemm=1 {float((y - ref).abs().max()):.2e}  bit-identical {bool(torch.equal(y, ref))}")
_lib.check(_lib.lib().sat_conv_set_option(b"k1_gemm", 2), "set_option")
