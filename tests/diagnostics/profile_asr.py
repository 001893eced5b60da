"""per-stage wall time of the ASR forward (batch 32 x 5 s): run on the GPU box"""
import sys
import torch
sys.path.insert(0, ".")
import satools_amd
from satools_amd import synthetic, ops

model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1").to("cuda")
bx = model.bn_extractor
wav = synthetic.harm_batch(list(range(32))).to("cuda")


def timed(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


feats = bx.features((wav * 32768))
full = bx._prepare_full(feats.device)
layers = bx._stack_layers()


def stack(x):
    xs = None
    for lay, c in zip(layers, bx._cache):
        x, xs = bx._tdnnf_layer(lay, c, x, xs)
    return x


x = stack(feats)
print("stack incl. VQ layer  %.3f ms" % timed(lambda: stack(feats)))
xp = ops.pad_replicate(x, 4, 4, interleave_right=True)


def after(x):
    xs = None
    for lay, c in full["after"]:
        x, xs = bx._tdnnf_layer(lay, c, x, xs)
    return x, xs


xa, xas = after(xp)
print("tdnnfs_after  %.3f ms" % timed(lambda: after(xp)), xa.shape)
(lay, c), (w, b, odim) = full["prefinal"][0], full["out"][0]
h, _ = bx._tdnnf_layer(lay, c, xa, xas)
print("prefinal  %.3f ms" % timed(lambda: bx._tdnnf_layer(lay, c, xa, xas)))
y = ops.conv1d(h, w, odim, 1, bias=b)
print("output affine  %.3f ms" % timed(lambda: ops.conv1d(h, w, odim, 1, bias=b)))
print("log_softmax  %.3f ms" % timed(lambda: ops.log_softmax_channels_(y)))
print("forward  %.3f ms" % timed(lambda: bx(wav.clone())))
