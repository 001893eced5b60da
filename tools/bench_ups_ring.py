"""The stride-4 upsamplers (ConvTranspose1d k 8, stride 4, padding 2: 256 -> 128 and 128 -> 64 channels) planes to planes: the LDS-transposed
epilogue of the 64 x 256 tile against the LDS-DMA ring with rows grouped by phase (sat_conv1d_desc.up_grouped), with and without the
zero tap slots skipped.  Checks the three against each other and against float64 first."""
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


def forms(C, k, u, w, b):
    pad = (k - u) // 2
    out = {}
    for name, grouped in (("tile", False), ("ring", True)):
        wc, ks, pl = packing.convtranspose_as_phase_conv(w, u, pad, grouped=grouped)
        out[name] = (packing.pack_conv_weight_f16x3(wc, up=u), ks, pl)
    return out, packing.convtranspose_zero_taps(k, u, pad)


def run(x, xs, hs, f, C, u, b, grouped, zero_taps=0):
    wp, ks, pl = f
    return ops.conv1d(x, wp, C // 2, ks, bias=b, pad_left=pl, up=u, mode=1, x_split=xs, y_split=hs, y_split_slope=0.1, no_y=True,
                      up_grouped=grouped, up_zero_taps=zero_taps)


torch.manual_seed(0)
for C, T, B in ((256, 133, 2), (128, 700, 3), (256, 1250, 32), (128, 5000, 32)):
    u, k = 4, 8
    x = torch.randn(B, C, T, device=dev)
    w = torch.randn(C, C // 2, k, device=dev) * (2.0 / (C * k)) ** 0.5
    b = torch.randn(C // 2, device=dev)
    xs = ops.act_split(x, 0.1)
    f, zt = forms(C, k, u, w, b)
    res = {}
    for name, grouped, z in (("tile", False, 0), ("ring", True, 0), ("ring_skip", True, zt)):
        hs = ops.split_like(B, C // 2, T * u, dev)
        run(x, xs, hs, f["tile" if not grouped else "ring"], C, u, b, grouped, z)
        res[name] = (hs, _lib.lib().sat_last_dispatch_name().decode())
    if B <= 3:
        ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv_transpose1d(torch.nn.functional.leaky_relu(x.double(), 0.1), w.double(), b.double(), stride=u, padding=(k - u) // 2), 0.1)
        back = {n: ops.unsplit(h).double() for n, (h, _) in res.items()}
        print(f"check C {C} T {T}: " + "  ".join(f"{n} vs f64 {(back[n] - ref).abs().max().item():.2e} [{res[n][1]}]" for n in back) +
              f"   ring == ring_skip: {bool(torch.equal(res['ring'][0], res['ring_skip'][0]))}   zero-tap mask {zt:#x}")
    else:
        print(f"C {C} T {T}: ring == ring_skip: {bool(torch.equal(res['ring'][0], res['ring_skip'][0]))}")
        hs = ops.split_like(B, C // 2, T * u, dev)
        fl = 2 * B * C * (C // 2) * k * T
        # (diagnostic bits of option convring: 2 = no K loop, 4 = no epilogue; results are wrong)
        for name, grouped, z, bits in (("tile", False, 0, 1), ("ring", True, 0, 1), ("ring_skip", True, zt, 1), ("skip, no epilogue", True, zt, 5), ("skip, no K loop", True, zt, 3), ("skip, neither", True, zt, 7)):
            _lib.check(_lib.lib().sat_conv_set_option(b"convring", bits), "opt")
            t = timed(lambda: run(x, xs, hs, f["tile" if not grouped else "ring"], C, u, b, grouped, z))
            print(f"   {name:18s} {t:7.1f} us   {fl / t / 1e6:5.0f} TF/s on the true taps ({fl / t / 1e6 / 833:.2f})", flush=True)
_lib.check(_lib.lib().sat_conv_set_option(b"convring", 1), "opt")
