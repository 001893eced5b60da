"""The persistent multi-job ring GEMM (gemm_walk16.hip, option "gemm_walk") against gemm_f16x3_ring16_kernel: same bits, time per
launch on the wav2vec2 encoder's shapes (batch 32 x 249 frames), q | k | v as one launch against three."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satools_amd  # noqa: E402,F401
from satools_amd import ops, packing, _lib  # noqa: E402

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


def opt(v):
    _lib.check(_lib.lib().sat_conv_set_option(b"gemm_walk", v), "opt")


torch.manual_seed(0)
for B, T in ((32, 249), (3, 100), (2, 257)):
    for cin, cout, kind in ((1024, 1024, "plain"), (1024, 4096, "gelu_planes"), (4096, 1024, "res"), (1024, 1024, "qkv")):
        x = torch.randn(B, cin, T, device=dev)
        xs = ops.act_split(x, 1.0)
        nj = 3 if kind == "qkv" else 1
        ws = [packing.pack_conv_weight_f16x3(torch.randn(cout, cin, 1, device=dev) * cin ** -0.5) for _ in range(nj)]
        bs = [torch.randn(cout, device=dev) for _ in range(nj)]
        res = torch.randn(B, cout, T, device=dev)
        tp = (T + 63) // 64 * 64

        def run(multi):
            outs = []
            if kind == "qkv":
                qs, ks = ops.split_like(B, cout, T, dev).zero_(), ops.split_like(B, cout, T, dev).zero_()
                v = torch.zeros(B, cout, tp, device=dev)
                jobs = [(x, ws[0], cout, 1, dict(bias=bs[0], mode=1, x_split=xs, y_split=qs, y_split_slope=1.0, no_y=True)),
                        (x, ws[1], cout, 1, dict(bias=bs[1], mode=1, x_split=xs, y_split=ks, y_split_slope=1.0, no_y=True)),
                        (x, ws[2], cout, 1, dict(bias=bs[2], mode=1, x_split=xs, out=v[:, :, :T]))]
                if multi:
                    f = lambda: ops.conv1d_multi(jobs)
                else:
                    def f():
                        for (xx, w, c, k, kw) in jobs:
                            ops.conv1d(xx, w, c, k, **kw)
                f()
                return f, (qs, ks, v)
            if kind == "plain":
                f = lambda: ops.conv1d(x, ws[0], cout, 1, bias=bs[0], mode=1, x_split=xs)
                return f, (f(),)
            if kind == "res":
                f = lambda: ops.conv1d(x, ws[0], cout, 1, bias=bs[0], mode=1, x_split=xs, res=res)
                return f, (f(),)
            ys = ops.split_like(B, cout, T, dev).zero_()
            f = lambda: ops.conv1d(x, ws[0], cout, 1, bias=bs[0], gelu=True, mode=1, x_split=xs, y_split=ys, y_split_slope=1.0, no_y=True)
            f()
            return f, (ys,)

        opt(0)
        f0, o0 = run(False)
        n0 = _lib.lib().sat_last_dispatch_name().decode()
        t0 = timed(f0)
        opt(3)
        f1, o1 = run(True)
        n1 = _lib.lib().sat_last_dispatch_name().decode()
        t1 = timed(f1)
        same = all(torch.equal(a, b) for a, b in zip(o0, o1))
        fl = 2.0 * B * T * cin * cout * nj
        print(f"B {B:2d} T {T:3d} {cin:4d} -> {cout:4d} {kind:12s}: {n0[:28]:28s} {t0:7.1f} us ({fl / t0 / 1e6:4.0f} TF/s)   {n1[:28]:28s} {t1:7.1f} us ({fl / t1 / 1e6:4.0f} TF/s)   same bits: {same}", flush=True)
opt(1)

# what the epilogues cost (diagnostic option bits: + 8 no K loop, + 16 no epilogue; results are wrong)
if os.environ.get("EPI"):
    B, T = 32, 249
    for cin, cout, kind in ((1024, 4096, "gelu_planes"), (1024, 4096, "planes"), (1024, 1024, "qkv")):
        x = torch.randn(B, cin, T, device=dev)
        xs = ops.act_split(x, 1.0)
        nj = 3 if kind == "qkv" else 1
        ws = [packing.pack_conv_weight_f16x3(torch.randn(cout, cin, 1, device=dev) * cin ** -0.5) for _ in range(nj)]
        bs = [torch.randn(cout, device=dev) for _ in range(nj)]
        ys = [ops.split_like(B, cout, T, dev) for _ in range(nj)]
        jobs = [(x, ws[j], cout, 1, dict(bias=bs[j], mode=1, x_split=xs, y_split=ys[j], y_split_slope=1.0, no_y=True, gelu=(kind == "gelu_planes"))) for j in range(nj)]
        f = (lambda: ops.conv1d_multi(jobs)) if nj > 1 else (lambda: ops.conv1d(*jobs[0][:4], **jobs[0][4]))
        t = {}
        for bits, what in ((3, "full"), (3 + 16, "no epilogue"), (3 + 8, "no K loop")):
            opt(bits)
            t[what] = timed(f)
        print(f"{cin} -> {cout} {kind:12s}: full {t['full']:6.1f} us   without the epilogues {t['no epilogue']:6.1f} us   without the K loops {t['no K loop']:6.1f} us", flush=True)
    opt(1)
