"""micro-benchmark of the TDNNF layer shapes (linearB 1024x3 -> 128, linearA 128 -> 1024 + BN + ReLU + bypass)"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import satools_amd
from satools_amd import ops, packing

B, T, H, Bn = 32, 534, 1024, 128
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


x = torch.randn(B, H, T, device=dev).relu()
wB = torch.randn(Bn, H, 3, device=dev) * 0.02
wA = torch.randn(H, Bn, 1, device=dev) * 0.05
bB, bA = torch.zeros(Bn, device=dev), torch.zeros(H, device=dev)
sc, sh = torch.ones(H, device=dev), torch.zeros(H, device=dev)
z = torch.randn(B, Bn, T - 2, device=dev)
for name, mode, pack in (("f32", 0, packing.pack_conv_weight), ("f16x3", 1, packing.pack_conv_weight_f16x3)):
    pB, pA = pack(wB), pack(wA)
    tB = timed(lambda: ops.conv1d(x, pB, Bn, 3, bias=bB, pad_left=0, pad_right=0, mode=mode))
    tA = timed(lambda: ops.conv1d(z, pA, H, 1, bias=bA, ch_scale=sc, ch_shift=sh, relu=True, res=x, res_scale=0.66, res_toff=1, mode=mode))
    print(f"{name:6s}: linearB {tB:7.1f} us   linearA {tA:7.1f} us")
pB, pA = packing.pack_conv_weight_f16x3(wB), packing.pack_conv_weight_f16x3(wA)
xs = ops.act_split(x, 1.0)
zs = ops.split_like(B, Bn, T - 2, dev)
ys = ops.split_like(B, H, T - 2, dev)
tB = timed(lambda: ops.conv1d(x, pB, Bn, 3, bias=bB, pad_left=0, pad_right=0, mode=1, x_split=xs, y_split=zs, no_y=True, out=z))
yb = torch.empty(B, H, T - 2, device=dev)
tA = timed(lambda: ops.conv1d(z, pA, H, 1, bias=bA, ch_scale=sc, ch_shift=sh, relu=True, res=x, res_scale=0.66, res_toff=1, mode=1,
                              x_split=zs, y_split=ys, out=yb))
print(f"planes: linearB {tB:7.1f} us   linearA {tA:7.1f} us (f32 + planes out)")
