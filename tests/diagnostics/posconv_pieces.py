"""wav2vec2 positional conv (grouped, 128 taps) as 12 chained 11-tap split-f16 convs vs the exact-f32 kernel"""
import sys
import torch
sys.path.insert(0, ".")
import satools_amd
from satools_amd import ops, packing

B, C, T, G, K = 32, 1024, 249, 16, 128
g = torch.Generator().manual_seed(0)
x = torch.randn(B, C, T, generator=g).cuda()
w = (torch.randn(C, C // G, K, generator=g) / (C // G * K) ** 0.5).cuda()
b = torch.randn(C, generator=g).cuda()
wp = packing.pack_conv_weight(w, groups=G)


def ref():
    return ops.conv1d(x, wp, C, K, bias=b, pad_left=64, pad_right=63, groups=G, gelu=True, post_res=x)


pieces = []
for s in range(12):
    ws = torch.zeros(C, C // G, 11, device="cuda")
    n = min(11, K - 11 * s)
    ws[:, :, :n] = w[:, :, 11 * s:11 * s + n]
    pieces.append(packing.pack_conv_weight_f16x3(ws, groups=G))


def chained():
    y = None
    for s, wsp in enumerate(pieces):
        pl = 64 - 11 * s
        y = ops.conv1d(x, wsp, C, 11, bias=b if s == 0 else None, pad_left=pl, pad_right=10 - pl, groups=G, mode=1,
                       res=y, gelu=(s == 11), post_res=None, out=y)
    return y + x


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


r, c = ref(), chained()
print("max abs diff", float((r - c).abs().max()), "ref %.3f ms" % timed(ref), "chained %.3f ms" % timed(chained))
