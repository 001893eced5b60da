"""SAT_CONV_F16F8R against SAT_CONV_F16X3 on the LDS-DMA ring kernel (csrc/conv_ring16.hip): the three-branch launches of the two
thick generator stages (one launch = the i-th conv of the 3 / 7 / 11-tap MRF branches), then the whole generator both ways,
interleaved.  python tools/bench_convring_f8.py [gen]"""
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


torch.manual_seed(0)
_lib.check(_lib.lib().sat_conv_set_option(b"convring_wr", int(os.environ.get("WR", "0"))), "convring_wr")      # 3: 128 x 320 tiles at C = 256 too
if "gen" not in sys.argv[1:]:
    for C, T in ((256, 1250), (128, 5000)):
        x = torch.randn(B, C, T, device=dev)
        xs = ops.act_split(x, 0.1)
        xs8 = ops.planes_f8_sidecar(xs)
        ks = (3, 7, 11)
        wf = [torch.randn(C, C, k, device=dev) * (k * C) ** -0.5 for k in ks]
        w3 = [packing.pack_conv_weight_f16x3(w) for w in wf]
        w8 = [packing.pack_conv_weight_f16f8r(w) for w in wf]
        bs = [torch.randn(C, device=dev) for _ in ks]
        for dil, kind in ((1, "conv1 (planes out)"), (5, "conv1 dil 5"), (1, "conv2 (+res, planes out)"), (1, "conv2 last (+res, MRF accumulate)")):
            def jobs(mode, ys, y8s, acc, only=None):
                out = []
                for j, k in enumerate(ks):
                    if only is not None and j != only:
                        continue
                    kw = dict(bias=bs[j], dilation=dil, pad_left=dil * (k - 1) // 2, mode=mode, x_split=xs, y_split_slope=0.1)
                    if mode == 3:
                        kw["x_split8"] = xs8
                    if kind.startswith("conv1"):
                        kw.update(y_split=ys[j], no_y=True)
                        if mode == 3:
                            kw.update(y_split8=y8s[j], y_split_hi_only=True)
                    elif "last" in kind:
                        kw.update(res_split=xs, res_split_slope=0.1, out=acc, accum=j > 0, accum_div=3.0 if j == 2 else 0.0, y_split=ys[2] if j == 2 else None)
                    else:
                        kw.update(y_split=ys[j], no_y=True, res_split=xs, res_split_slope=0.1)
                        if mode == 3:
                            kw.update(y_split8=y8s[j])
                    out.append((x, (w8 if mode == 3 else w3)[j], C, k, kw))
                return out

            ys = [ops.split_like(B, C, T, dev).zero_() for _ in ks]
            y8s = [ops.sidecar_like(B, C, T, dev).zero_() for _ in ks]
            acc = torch.zeros(B, C, T, device=dev)
            t = {}
            for rnd in range(2):
                for mode in (1, 3):
                    t[mode] = timed(lambda: ops.conv1d_multi(jobs(mode, ys, y8s, acc)))
            t11 = {mode: timed(lambda: ops.conv1d_multi(jobs(mode, ys, y8s, acc, only=2))) for mode in (1, 3)}
            flop = 2.0 * B * C * C * sum(ks) * T
            print(f"C {C} T {T} {kind:34s}: f16x3 {t[1]:6.1f} us ({flop / t[1] / 1e6:5.0f} TF/s)   f16f8r {t[3]:6.1f} us ({flop / t[3] / 1e6:5.0f} TF/s)   x{t[1] / t[3]:.2f}"
                  f"     k = 11 alone: {t11[1]:6.1f} -> {t11[3]:6.1f} us", flush=True)

# the whole generator, interleaved rounds
model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1")
model.to(dev)
g = model.hifigan
x = torch.randn(32, g.imput_dim, 250, device=dev)
res, outs = {"f16x3": [], "f16f8r": []}, {}
for rnd in range(6):
    for prec in ("f16x3", "f16f8r"):
        g.precision = prec
        g.invalidate()
        for _ in range(3):
            y = g(x)[0]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            y = g(x)[0]
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            res[prec].append(e0.elapsed_time(e1) / 8)
        outs[prec] = y
for prec, ts in res.items():
    print(f"generator {prec}: ms per forward, five rounds: " + " ".join(f"{t:.2f}" for t in ts) + f"   median {sorted(ts)[2]:.2f}")
d = (outs["f16x3"] - outs["f16f8r"]).double()
print(f"f16f8r vs f16x3 waveform: rms {float(d.pow(2).mean().sqrt()):.3e}  max {float(d.abs().max()):.3e}  (signal rms {float(outs['f16x3'].double().pow(2).mean().sqrt()):.3f})")
