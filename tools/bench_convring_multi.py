"""One launch for the three branches of an MRF block (sat_conv1d_multi_f32, conv_ring16.hip) against three launches of the
ring kernel and of the register-staged tile, on the two thick generator stages; results compared bit for bit."""
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


torch.manual_seed(0)
for C, T in ((256, 1250), (128, 5000)):
    x = torch.randn(B, C, T, device=dev)
    xs = ops.act_split(x, 0.1)
    ks = (3, 7, 11)
    ws = [packing.pack_conv_weight_f16x3(torch.randn(C, C, k, device=dev) * (k * C) ** -0.5) for k in ks]
    bs = [torch.randn(C, device=dev) for _ in ks]
    for dil, kind in ((1, "conv1 (planes out)"), (5, "conv1 dil 5"), (1, "conv2 (+res, planes out)"), (1, "conv2 last (+res, MRF accumulate)")):
        def jobs(ys, acc):
            out = []
            for j, k in enumerate(ks):
                kw = dict(bias=bs[j], dilation=dil, pad_left=dil * (k - 1) // 2, mode=1, x_split=xs, y_split_slope=0.1)
                if kind.startswith("conv1"):
                    kw.update(y_split=ys[j], no_y=True)
                elif "last" in kind:
                    kw.update(res_split=xs, res_split_slope=0.1, out=acc, accum=j > 0, accum_div=3.0 if j == 2 else 0.0, y_split=ys[2] if j == 2 else None)
                else:
                    kw.update(y_split=ys[j], no_y=True, res_split=xs, res_split_slope=0.1)
                out.append((x, ws[j], C, k, kw))
            return out

        def single(ys, acc):
            for (xx, w, c, k, kw) in jobs(ys, acc):
                ops.conv1d(xx, w, c, k, **kw)

        def multi(ys, acc):
            ops.conv1d_multi(jobs(ys, acc))

        res = {}
        for name, v, f in (("lean x3", 0, single), ("ring x3", 1, single), ("ring multi", 1, multi)):
            opt(v)
            ys = [ops.split_like(B, C, T, dev).zero_() for _ in ks]
            acc = torch.zeros(B, C, T, device=dev)
            f(ys, acc)
            torch.cuda.synchronize()
            res[name] = ([y.clone() for y in ys], acc.clone(), timed(lambda: f(ys, acc)))
        same = all(torch.equal(a, b) for a, b in zip(res["ring x3"][0], res["ring multi"][0])) and torch.equal(res["ring x3"][1], res["ring multi"][1])
        dl = (res["lean x3"][1] - res["ring multi"][1]).abs().max().item()
        print(f"C {C} T {T} {kind:34s}: lean x3 {res['lean x3'][2]:6.1f} us   ring x3 {res['ring x3'][2]:6.1f} us   ring multi {res['ring multi'][2]:6.1f} us"
              f"   multi == singles: {same}   |lean - ring| (f32 sum) {dl:.1e}", flush=True)
opt(0)

# cycle stamps of the multi launch (diagnostic instantiation)
if os.environ.get("STAMPS"):
    l = _lib.lib()
    opt(1)
    for C, T in ((256, 1250), (128, 5000)):
        x = torch.randn(B, C, T, device=dev)
        xs = ops.act_split(x, 0.1)
        ks = (3, 7, 11)
        ws = [packing.pack_conv_weight_f16x3(torch.randn(C, C, k, device=dev) * (k * C) ** -0.5) for k in ks]
        bs = [torch.randn(C, device=dev) for _ in ks]
        ys = [ops.split_like(B, C, T, dev) for _ in ks]
        jobs = [(x, ws[j], C, k, dict(bias=bs[j], dilation=1, pad_left=(k - 1) // 2, mode=1, x_split=xs, y_split_slope=0.1, y_split=ys[j], no_y=True)) for j, k in enumerate(ks)]
        for _ in range(5):
            ops.conv1d_multi(jobs)
        buf = torch.zeros(512 * 2 * 8, dtype=torch.int64, device=dev)
        l.sat_convring_debug_stamps(buf.data_ptr())
        for _ in range(3):
            ops.conv1d_multi(jobs)
        torch.cuda.synchronize()
        l.sat_convring_debug_stamps(None)
        r = buf.view(512, 2, 8).cpu().double()
        r = r[r[:, 0, 4] > 0]
        for h, name in ((0, "early"), (1, "late ")):
            q = r[:, h]
            clk = (q[:, 4] / q[:, 5]).median().item() * 0.1
            print(f"multi C {C} {name}: blocks {len(q)} tiles {q[0, 7]:.0f}  first-operand waits {q[:, 0].median():7.0f}  loops {q[:, 1].median():8.0f}  step-head waits {q[:, 2].median():8.0f}"
                  f"  epilogues {q[:, 3].median():7.0f}  kernel {q[:, 4].median():8.0f} cyc = {q[:, 5].median() / 100:6.1f} us at {clk:4.2f} GHz", flush=True)
    opt(0)

# what the epilogues cost in the three-branch launch (diagnostic bit 4 of the option: no epilogue; results are wrong)
if os.environ.get("EPI"):
    for C, T in ((256, 1250), (128, 5000)):
        x = torch.randn(B, C, T, device=dev)
        xs = ops.act_split(x, 0.1)
        ks = (3, 7, 11)
        ws = [packing.pack_conv_weight_f16x3(torch.randn(C, C, k, device=dev) * (k * C) ** -0.5) for k in ks]
        bs = [torch.randn(C, device=dev) for k in ks]
        ys = [ops.split_like(B, C, T, dev) for _ in ks]
        for kind in ("conv1", "conv2"):
            jobs = [(x, ws[j], C, k, dict(bias=bs[j], dilation=1, pad_left=(k - 1) // 2, mode=1, x_split=xs, y_split_slope=0.1, y_split=ys[j], no_y=True,
                                          **(dict(res_split=xs, res_split_slope=0.1) if kind == "conv2" else {}))) for j, k in enumerate(ks)]
            t = {}
            for bits, what in ((1, "full"), (5, "no epilogue"), (3, "no K loop")):
                opt(bits)
                t[what] = timed(lambda: ops.conv1d_multi(jobs))
            print(f"C {C} {kind}: full {t['full']:6.1f} us   without the epilogues {t['no epilogue']:6.1f} us   without the K loops {t['no K loop']:6.1f} us", flush=True)
    opt(1)
