"""time the fused self-attention kernel alone: 32 utterances x 16 heads x 64 dims, 249 frames (the wav2vec2 tag's shape)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satools_amd
from satools_amd import ops

B, H, D = 32, 16, 64
for T in (249, 499):
    tp = (T + 63) // 64 * 64
    q, k = torch.randn(B, H * D, T, device="cuda"), torch.randn(B, H * D, T, device="cuda")
    v = torch.zeros(B, H * D, tp, device="cuda")
    v[:, :, :T] = torch.randn(B, H * D, T, device="cuda")
    qs, ks = ops.act_split(q, 1.0), ops.act_split(k, 1.0)
    f = lambda: ops.attention_fused(qs, ks, v, B, H, D, T, D ** -0.5)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 4.0 * B * H * T * T * D
    print(f"T={T}: {us:7.1f} us per launch, {fl / us / 1e6:6.1f} TFLOP/s useful")

    if "stamps" in sys.argv:
        from satools_amd import _lib
        buf = torch.zeros(56, dtype=torch.int64, device="cuda")
        _lib.lib().sat_attention_debug_stamps(buf.data_ptr())
        f()
        torch.cuda.synchronize()
        _lib.lib().sat_attention_debug_stamps(None)
        st = buf.cpu().numpy().reshape(7, 8)
        names = ["wait K_0 + Q, barrier", "S = K^T Q", "softmax", "wait V, barrier", "V f32 -> split image", "O += V P"]
        print("   phase cycles of block (0,0,0), first stage (min / max over waves):")
        for i, nme in enumerate(names):
            d = st[i + 1] - st[i]
            print(f"     {nme:28s} {int(d.min()):7d} {int(d.max()):7d}")
        print(f"     whole {int((st[6] - st[0]).max())} cycles")
