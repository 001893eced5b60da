"""micro-benchmark of the five polyphase upsamplers (split planes in, f32 out) and the split pass that follows"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import satools_amd
from satools_amd import ops, packing

B = 32
STAGES = [(512, 250, 5, 11), (256, 1250, 4, 8), (128, 5000, 4, 8), (64, 20000, 2, 4), (32, 40000, 2, 4)]   # C_in, T_in, u, k
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


tot = 0.0
for C, T, u, k in STAGES:
    x = torch.randn(B, C, T, device=dev)
    w = torch.randn(C, C // 2, k, device=dev) * 0.05          # ConvTranspose1d weight [C_in, C_out, k]
    wc, ks, pl = packing.convtranspose_as_phase_conv(w, u, (k - u) // 2)
    wp = packing.pack_conv_weight_f16x3(wc, up=u)
    b = torch.zeros(C // 2, device=dev)
    xs = ops.act_split(x, 0.1)
    out = torch.empty(B, C // 2, T * u, device=dev)
    t_up = timed(lambda: ops.conv1d(x, wp, C // 2, ks, bias=b, pad_left=pl, up=u, mode=1, x_split=xs, out=out))
    hs = ops.split_like(B, C // 2, T * u, dev)
    t_sp = timed(lambda: ops.act_split(out, 0.1, out=hs))
    gb = (B * C * T * 4 + B * C // 2 * T * u * 4) / 1e9
    print(f"ups C {C}->{C//2} T {T}->{T*u} (u={u}, {ks} taps): conv {t_up:7.1f} us ({gb / t_up * 1e3:5.2f} TB/s)   split pass {t_sp:6.1f} us")
    tot += t_up + t_sp
print(f"total {tot:.1f} us")
