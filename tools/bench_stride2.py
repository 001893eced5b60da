"""the wav2vec2 feature extractor's stride-2 3-tap convs (512 -> 512 on [even | odd] phase planes, 32 utterances):
two-tap polyphase form (one zero tap) on conv1d_f16x3_planes_kernel against the one-product wrapped form on the ring
GEMM (sat_conv1d_desc.x_wrap_channels)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satools_amd
from satools_amd import ops, packing
from satools_amd.wav2vec2 import _polyphase_stride2_weight

B, C = 32, 512


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
    return e0.elapsed_time(e1) / n * 1e3


for Tq in (7999, 3999, 1999, 999):
    Th = Tq + 1
    ph = torch.randn(B, 2 * C, Th, device="cuda")
    xs = ops.act_split(ph, 1.0)
    w = torch.randn(C, C, 3, device="cuda") / (3 * C) ** 0.5
    b = torch.randn(C, device="cuda")
    wc, kp = _polyphase_stride2_weight(w)
    wp = packing.pack_conv_weight_f16x3(wc)
    ww = packing.pack_conv_weight_f16x3(torch.cat([w[:, :, 0], w[:, :, 1], w[:, :, 2]], 1).unsqueeze(-1).contiguous())
    f1 = lambda: ops.conv1d(ph, wp, C, kp, bias=b, pad_left=0, pad_right=0, t_out=Tq, mode=1, x_split=xs)
    f2 = lambda: ops.conv1d(ph, ww, C, 1, bias=b, t_out=Tq, mode=1, x_split=xs, x_wrap_channels=2 * C, c_in=3 * C)
    y1, y2 = f1(), f2()
    u1, u2 = timed(f1), timed(f2)
    fl = 2.0 * C * 3 * C * B * Tq
    print(f"T_q={Tq:5d}: polyphase 2-tap {u1:8.1f} us ({fl / u1 / 1e6:6.1f} TFLOP/s useful)   wrapped GEMM {u2:8.1f} us ({fl / u2 / 1e6:6.1f})   "
          f"x{u1 / u2:.2f}   max diff {float((y1 - y2).abs().max()):.1e}")
