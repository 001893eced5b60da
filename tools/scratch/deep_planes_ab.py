"""TDNNF linearB (1024 x 3 -> 128, planes in / planes out, 32 x 252 frames) with one and with two stages of loads in flight
(sat_conv_set_option("deep_planes")), bits compared; then get_bn of the fbank tag either way"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import satools_amd
from satools_amd import ops, packing, _lib, synthetic
B, T, H, Bn, dev = 32, 252, 1024, 128, "cuda"


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


def setopt(v):
    _lib.check(_lib.lib().sat_conv_set_option(b"deep_planes", v), "set_option")


torch.manual_seed(0)
x = torch.randn(B, H, T, device=dev).relu()
wB = packing.pack_conv_weight_f16x3(torch.randn(Bn, H, 3, device=dev) * 0.02)
bB = torch.randn(Bn, device=dev) * 0.1
xs = ops.act_split(x, 1.0)
z = torch.empty(B, Bn, T - 2, device=dev)
outs = []
for rnd in range(3):
    for v in (0, 1):
        setopt(v)
        zs = ops.split_like(B, Bn, T - 2, dev)
        t = timed(lambda: ops.conv1d(x, wB, Bn, 3, bias=bB, pad_left=0, pad_right=0, mode=1, x_split=xs, y_split=zs, no_y=True, out=z))
        outs.append(zs.clone())
        print(f"deep_planes={v}: linearB {t:6.1f} us")
print("bit-identical:", all(torch.equal(outs[0], o) for o in outs[1:]))
for Hc in (512, 48 * 16, 16, 32):        # stage counts 16, 24, odd ones (1 chunk = 0.5 stage is not served: S = 2 needs an even chunk count), 1
    xx = torch.randn(3, Hc, 77, device=dev)
    ww = packing.pack_conv_weight_f16x3(torch.randn(Bn, Hc, 3, device=dev) * 0.05)
    xxs = ops.act_split(xx, 1.0)
    r = []
    for v in (0, 1):
        setopt(v)
        zs = ops.split_like(3, Bn, 75, dev)
        zz = ops.conv1d(xx, ww, Bn, 3, bias=bB, pad_left=0, pad_right=0, mode=1, x_split=xxs, y_split=zs)
        r.append((zz.clone(), zs.clone()))
    print(f"C_in {Hc}: f32 and planes bit-identical: {torch.equal(r[0][0], r[1][0]) and torch.equal(r[0][1], r[1][1])}")
model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1")
model.to(dev)
model.eval()
model.bn_extractor.vq_tie_sigmas = 0.0
wav = synthetic.harm_batch(list(range(32))).to(dev)
res = []
for rnd in range(3):
    for v in (0, 1):
        setopt(v)
        t = timed(lambda: model.get_bn(wav), n=20) / 1e3
        res.append(model.get_bn(wav).clone())
        print(f"deep_planes={v}: get_bn {t:6.3f} ms")
print("get_bn bit-identical:", all(torch.equal(res[0], o) for o in res[1:]))
