"""time the 1x1 conv on split planes (the Linear layers of the wav2vec2 tag, TDNNF linearA/B) with the 128 x 128
GEMM kernel (k1_gemm = 1), the LDS-DMA ring GEMM (k1_gemm = 2) and the ring on the 16x16x32 MFMA shape (k1_gemm = 3): B=32 utterances x 249 frames"""
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


def run(cin, cout, opt, variant, x, xs, wp, bias, res, scale, shift):
    _lib.check(_lib.lib().sat_conv_set_option(b"k1_gemm", opt), "set_option")
    kw = dict(bias=bias, mode=1, x_split=xs)
    if variant == "gelu_planes":
        ys = ops.split_like(B, cout, T, x.device)
        kw.update(gelu=True, y_split=ys)
        f = lambda: ops.conv1d(x, wp, cout, 1, **kw)
        y = f()
        # the planes written next to y must be split(y): feed them to a second product and compare with the f32 input
        w2 = torch.eye(cout, device=x.device)[:128].reshape(128, cout, 1).contiguous()
        w2p = packing.pack_conv_weight_f16x3(w2)
        back = ops.conv1d(y, w2p, 128, 1, mode=1, x_split=ys)
        assert float((back - y[:, :128]).abs().max()) < 1e-5 * float(y.abs().max()), "planes store"
        return f, y
    if variant == "res":
        kw.update(res=res)
    if variant == "bn_relu_postres":
        kw.update(post_res=res, ch_scale=scale, ch_shift=shift, relu=True)
    f = lambda: ops.conv1d(x, wp, cout, 1, **kw)
    return f, f()


for cin, cout in ((1024, 1024), (1024, 3072), (1024, 4096), (4096, 1024), (128, 1024), (1024, 3280), (512, 1024)):
    x = torch.randn(B, cin, T, device="cuda")
    w = torch.randn(cout, cin, 1, device="cuda") / cin ** 0.5
    wp = packing.pack_conv_weight_f16x3(w)
    xs = ops.act_split(x, 1.0)
    bias = torch.randn(cout, device="cuda")
    res = torch.randn(B, cout, T, device="cuda")
    scale, shift = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
    exact = torch.einsum("oc,bct->bot", w[:, :, 0].double(), x.double()) + bias.double()[None, :, None]
    for variant in ("plain", "res", "gelu_planes", "bn_relu_postres"):
        ref = None
        for opt in (1, 2, 3):
            f, y = run(cin, cout, opt, variant, x, xs, wp, bias, res, scale, shift)
            us = timed(f)
            if ref is None:
                ref = y
            fl = 2.0 * cin * cout * B * T
            same = bool(torch.equal(y, ref))
            if y.dtype == torch.float32:
                d = float((y - ref).abs().max())
                e = f"  err vs f64 {float((y.double() - exact).abs().max()):.2e}" if variant == "plain" else ""
            else:
                d, e = float("nan"), ""
            print(f"{cin:5d} -> {cout:5d} {variant:16s} k1_gemm={opt}  {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s useful   "
                  f"max diff vs k1_gemm=1 {d:.2e}  bit-identical {same}{e}")
_lib.check(_lib.lib().sat_conv_set_option(b"k1_gemm", 3), "set_option")
