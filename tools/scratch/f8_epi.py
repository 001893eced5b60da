"""what the epilogues and the K loops of the F8 three-branch launches cost (diagnostic bits of option convring: 4 = no epilogue, 2 = no K loop; results are wrong)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import satools_amd
from satools_amd import ops, packing, _lib
B, dev = 32, "cuda"
def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for C, T in ((256, 1250), (128, 5000)):
    x = torch.randn(B, C, T, device=dev)
    xs = ops.act_split(x, 0.1); xs8 = ops.planes_f8_sidecar(xs)
    ks = (3, 7, 11)
    w8 = [packing.pack_conv_weight_f16f8r(torch.randn(C, C, k, device=dev) * (k * C) ** -0.5) for k in ks]
    bs = [torch.randn(C, device=dev) for _ in ks]
    ys = [ops.split_like(B, C, T, dev) for _ in ks]; y8 = [ops.sidecar_like(B, C, T, dev) for _ in ks]
    for kind in ("conv1", "conv2"):
        jobs = [(x, w8[j], C, k, dict(bias=bs[j], dilation=1, pad_left=(k - 1) // 2, mode=3, x_split=xs, x_split8=xs8, y_split_slope=0.1, y_split=ys[j], y_split8=y8[j], no_y=True,
                                      **(dict(res_split=xs, res_split_slope=0.1) if kind == "conv2" else dict(y_split_hi_only=True)))) for j, k in enumerate(ks)]
        t = {}
        for bits, what in ((1, "full"), (5, "no epilogue"), (3, "no K loop")):
            _lib.check(_lib.lib().sat_conv_set_option(b"convring", bits), "opt")
            t[what] = timed(lambda: ops.conv1d_multi(jobs))
        print(f"C {C} {kind}: full {t['full']:6.1f} us   without the epilogues {t['no epilogue']:6.1f} us   without the K loops {t['no K loop']:6.1f} us", flush=True)
_lib.check(_lib.lib().sat_conv_set_option(b"convring", 1), "opt")
