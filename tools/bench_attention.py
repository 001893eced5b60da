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
