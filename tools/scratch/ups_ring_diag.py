import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import satools_amd
from satools_amd import ops, packing, _lib
dev = "cuda"
torch.manual_seed(0)
C, T, B, u, k = 256, 133, 1, 4, 8
x = torch.randn(B, C, T, device=dev)
w = torch.randn(C, C // 2, k, device=dev) * (2.0 / (C * k)) ** 0.5
b = torch.randn(C // 2, device=dev)
xs = ops.act_split(x, 0.1)
wc, ks, pl = packing.convtranspose_as_phase_conv(w, u, 2, grouped=True)
wp = packing.pack_conv_weight_f16x3(wc, up=u)
# (1) the grouped rows as a plain 512-row conv: ring against the tile kernel, f32 outputs
b512 = torch.randn(512, device=dev)
outs = []
for v in (0, 33):
    _lib.check(_lib.lib().sat_conv_set_option(b"convring", v), "opt")
    ys = ops.split_like(B, 512, T, dev)
    y = ops.conv1d(x, wp, 512, ks, bias=b512, pad_left=pl, mode=1, x_split=xs, y_split=ys, y_split_slope=0.1)
    outs.append((y.clone(), ops.unsplit(ys).clone(), _lib.lib().sat_last_dispatch_name().decode()))
_lib.check(_lib.lib().sat_conv_set_option(b"convring", 1), "opt")
print("plain 512-row conv: ring vs tile f32", (outs[0][0] - outs[1][0]).abs().max().item(), " planes", (outs[0][1] - outs[1][1]).abs().max().item(), outs[0][2], outs[1][2])
ref_rows = outs[0][0]          # [B, 512, T]: row (g*4 + r)*16 + c
# (2) the upsampler form
hs = ops.split_like(B, C // 2, T * u, dev)
ops.conv1d(x, wp, C // 2, ks, bias=b, pad_left=pl, up=u, mode=1, x_split=xs, y_split=hs, y_split_slope=0.1, no_y=True, up_grouped=True)
got = ops.unsplit(hs)          # [B, 128, 4T] = lrelu(y)
want = ref_rows - b512[None, :, None]
want = want.reshape(B, 8, 4, 16, T).permute(0, 1, 3, 4, 2).reshape(B, 128, 4 * T) + b[None, :, None]
want = torch.nn.functional.leaky_relu(want, 0.1)
err = (got - want).abs()
print("ups form vs rearranged plain rows: max", err.max().item())
e = err.reshape(B, 8, 16, T, 4).amax(dim=(0, 3))      # [g][c][phase]
print("per (group, phase) max err:\n", e.amax(dim=1))
print("per channel-in-group max err:", e.amax(dim=(0, 2)))
print("per q max err (first 40):", err.reshape(B, 128, T, 4).amax(dim=(0, 1, 3))[:40])
torch.set_printoptions(precision=4, linewidth=200)
print("got ", got[0, 0, :16])
print("want", want[0, 0, :16])
print("got ", got[0, 5, 64:80])
print("want", want[0, 5, 64:80])
hs2 = ops.split_like(B, C // 2, T * u, dev)
ops.conv1d(x, wp, C // 2, ks, bias=b, pad_left=pl, up=u, mode=1, x_split=xs, y_split=hs2, y_split_slope=0.1, no_y=True, up_grouped=True)
print("deterministic:", bool(torch.equal(hs, hs2)))
# is a wrong value some other phase's right value?
g4 = got.reshape(B, 128, T, 4); w4 = want.reshape(B, 128, T, 4)
for r in range(4):
    print("phase", r, "got vs want phase r':", [f"{(g4[..., r] - w4[..., r2]).abs().max().item():.3f}" for r2 in range(4)])
# without the bias and activation differences: zero bias
b0 = torch.zeros(C // 2, device=dev)
hs3 = ops.split_like(B, C // 2, T * u, dev)
ops.conv1d(x, wp, C // 2, ks, bias=b0, pad_left=pl, up=u, mode=1, x_split=xs, y_split=hs3, y_split_slope=1.0, no_y=True, up_grouped=True)
w0 = (ref_rows - b512[None, :, None]).reshape(B, 8, 4, 16, T).permute(0, 1, 3, 4, 2).reshape(B, 128, 4 * T)
g0 = ops.unsplit(hs3)
print("zero bias, no activation: max err", (g0 - w0).abs().max().item(), " per phase", (g0 - w0).abs().reshape(B, 128, T, 4).amax(dim=(0, 1, 2)))
