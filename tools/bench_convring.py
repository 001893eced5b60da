"""The LDS-DMA ring conv on the 16x16x32 shape (conv_ring16.hip, option "convring") against the register-staged conv tile
(conv_lean.hip) on the generator's thick stages at batch 32: time per launch, and the two outputs against each other
(f32 output and split planes, with the residual taken from planes and the MRF accumulation on) and against float64."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satools_amd  # noqa: E402,F401
from satools_amd import ops, packing, _lib  # noqa: E402

B, dev = 32, "cuda"


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
    _lib.check(_lib.lib().sat_conv_set_option(b"convring", v), "opt")


def check_small():
    torch.manual_seed(0)
    for C, T, k, dil in ((256, 333, 11, 5), (128, 700, 7, 3), (256, 160, 3, 1), (192, 401, 3, 5), (128, 97, 11, 1), (512, 200, 7, 1)):
        x = torch.randn(2, C, T, device=dev)
        w = torch.randn(C, C, k, device=dev) * (k * C) ** -0.5
        b = torch.randn(C, device=dev)
        r = torch.randn(2, C, T, device=dev)
        wp = packing.pack_conv_weight_f16x3(w)
        xs = ops.act_split(x, 0.1)
        rs = ops.act_split(r, 0.1)
        ref = torch.nn.functional.conv1d(torch.nn.functional.leaky_relu(x.double(), 0.1), w.double(), b.double(), padding=dil * (k - 1) // 2, dilation=dil) + r.double()
        outs = []
        for v in (0, 1):
            opt(33 * v)
            ys = ops.split_like(2, C, T, dev)
            y = ops.conv1d(x, wp, C, k, bias=b, dilation=dil, pad_left=dil * (k - 1) // 2, mode=1, x_split=xs, y_split=ys, y_split_slope=0.1,
                           res_split=rs, res_split_slope=0.1)
            name = _lib.lib().sat_last_dispatch_name().decode()
            outs.append((y, ys, name))
        e0 = (outs[0][0].double() - ref).abs().max().item()
        e1 = (outs[1][0].double() - ref).abs().max().item()
        d = (outs[0][0] - outs[1][0]).abs().max().item()
        # the planes written next to y must be split(lrelu(y)): feed them to an identity product and compare with the f32 output
        ne = min(C, 128)
        eye = packing.pack_conv_weight_f16x3(torch.eye(C, device=dev)[:ne].reshape(ne, C, 1).contiguous())
        opt(0)
        back = ops.conv1d(outs[1][0], eye, ne, 1, mode=1, x_split=outs[1][1])
        pl = (back - torch.nn.functional.leaky_relu(outs[1][0][:, :ne], 0.1)).abs().max().item()
        print(f"check C {C} T {T} k {k} dil {dil}: lean-vs-f64 {e0:.2e}  ring-vs-f64 {e1:.2e}  lean-vs-ring {d:.2e}  planes-vs-own-f32 {pl:.2e}  [{outs[1][2]}]")
    opt(0)


check_small()
shapes = [(256, 1250), (128, 5000)] + [(int(c), int(t)) for c, t in (a.split(":") for a in sys.argv[1:])]
for C, T in shapes:
    x = torch.randn(B, C, T, device=dev)
    xs = ops.act_split(x, 0.1)
    ys = ops.split_like(B, C, T, dev)
    for k in (3, 7, 11):
        w = packing.pack_conv_weight_f16x3(torch.randn(C, C, k, device=dev) * (k * C) ** -0.5)
        b = torch.randn(C, device=dev)
        for dil in (1, 5):
            t = []
            for v in (0, 1):
                opt(v)
                t.append(timed(lambda: ops.conv1d(x, w, C, k, bias=b, dilation=dil, pad_left=dil * (k - 1) // 2, mode=1, x_split=xs, y_split=ys, y_split_slope=0.1, no_y=True)))
            # conv2 of a step: residual from planes
            opt(1)
            t2 = timed(lambda: ops.conv1d(x, w, C, k, bias=b, dilation=dil, pad_left=dil * (k - 1) // 2, mode=1, x_split=xs, y_split=ys, y_split_slope=0.1, no_y=True, res_split=xs, res_split_slope=0.1))
            fl = 2 * B * C * C * k * T
            print(f"C {C:3d} T {T:5d} k {k:2d} dil {dil}: lean {t[0]:6.1f} us ({fl / t[0] / 1e6:4.0f} TF/s)   ring {t[1]:6.1f} us ({fl / t[1] / 1e6:4.0f} TF/s, {fl / t[1] / 1e6 / 833:.2f})   ring+res {t2:6.1f} us", flush=True)
opt(0)

# where the fixed cost of a launch sits (diagnostic bits of the option; results are wrong)
if os.environ.get("CONVRING_ABLATE"):
    C, T = 256, 1250
    x = torch.randn(B, C, T, device=dev)
    xs = ops.act_split(x, 0.1)
    ys = ops.split_like(B, C, T, dev)
    for k in (3, 11):
        w = packing.pack_conv_weight_f16x3(torch.randn(C, C, k, device=dev) * (k * C) ** -0.5)
        b = torch.randn(C, device=dev)
        for bits, what in ((1, "full"), (1 + 8, "no stores"), (1 + 4, "no epilogue"), (1 + 2, "no K loop"), (1 + 2 + 4, "prologue only"), (1 + 2 + 8, "no loop, no stores")):
            opt(bits)
            t = timed(lambda: ops.conv1d(x, w, C, k, bias=b, dilation=1, pad_left=(k - 1) // 2, mode=1, x_split=xs, y_split=ys, y_split_slope=0.1, no_y=True))
            print(f"ablate k {k:2d} {what:20s}: {t:6.1f} us", flush=True)
    opt(0)
