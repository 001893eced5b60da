"""what a K-split of linearB and a planes-only linearA could gain (T = 252 frames, batch 32): the SAME kernels on shapes that have the
blocks and the bytes of the candidates — linearB with C_in / 4 and 4 x the batch = the main pass of a four-way K split;
linearA without its f32 store and bypass from f32 (the planes-only form also reads the bypass as planes: the same bytes)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from satools_amd import ops, packing
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


pk = packing.pack_conv_weight_f16x3
for ks in (1, 2, 4, 8):
    Hc, Bc = H // ks, B * ks
    x = torch.randn(Bc, Hc, T, device=dev).relu()
    wB = pk(torch.randn(Bn, Hc, 3, device=dev) * 0.02)
    bB = torch.zeros(Bn, device=dev)
    xs = ops.act_split(x, 1.0)
    z = torch.empty(Bc, Bn, T - 2, device=dev)
    zs = ops.split_like(Bc, Bn, T - 2, dev)
    t_planes = timed(lambda: ops.conv1d(x, wB, Bn, 3, bias=bB, pad_left=0, pad_right=0, mode=1, x_split=xs, y_split=zs, no_y=True, out=z))
    t_f32 = timed(lambda: ops.conv1d(x, wB, Bn, 3, bias=bB, pad_left=0, pad_right=0, mode=1, x_split=xs, out=z))
    # the reduction: ks partial f32 tensors -> planes (bytes of an act_split over ks x the tensor)
    zz = torch.randn(B * ks, Bn, T - 2, device=dev)
    zs2 = ops.split_like(B * ks, Bn, T - 2, dev)
    t_red = timed(lambda: ops.act_split(zz, 1.0, out=zs2)) if ks > 1 else 0.0
    print(f"linearB K / {ks} x batch {Bc}: planes out {t_planes:6.1f} us, f32 partials out {t_f32:6.1f} us, reduction pass ~{t_red:5.1f} us")
x = torch.randn(B, H, T, device=dev).relu()
wA = pk(torch.randn(H, Bn, 1, device=dev) * 0.05)
bA, sc, sh = torch.zeros(H, device=dev), torch.ones(H, device=dev), torch.zeros(H, device=dev)
z = torch.randn(B, Bn, T - 2, device=dev)
zs = ops.act_split(z, 1.0)
ys = ops.split_like(B, H, T - 2, dev)
yb = torch.empty(B, H, T - 2, device=dev)
kw = dict(bias=bA, ch_scale=sc, ch_shift=sh, relu=True, mode=1, x_split=zs)
t_now = timed(lambda: ops.conv1d(z, wA, H, 1, res=x, res_scale=0.66, res_toff=1, y_split=ys, out=yb, **kw))
t_noy = timed(lambda: ops.conv1d(z, wA, H, 1, res=x, res_scale=0.66, res_toff=1, y_split=ys, no_y=True, out=yb, **kw))
t_nores = timed(lambda: ops.conv1d(z, wA, H, 1, y_split=ys, no_y=True, out=yb, **kw))
print(f"linearA: f32 + planes out, f32 bypass {t_now:6.1f} us;  planes only out, f32 bypass {t_noy:6.1f} us;  planes only out, no bypass {t_nores:6.1f} us")
