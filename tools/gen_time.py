"""generator forward time (batch 32 x 250 frames, the library's default arithmetic), one line: python tools/gen_time.py [rounds]"""
import os
import sys
import zlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satools_amd

model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1")
model.to("cuda")
g = model.hifigan
torch.manual_seed(0)
x = torch.randn(32, g.imput_dim, 250, device="cuda")
for _ in range(4):
    y = g(x)[0]
ts = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    e0.record()
    for _ in range(8):
        y = g(x)[0]
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 8)
print("generator forward ms:", " ".join(f"{t:.3f}" for t in ts), f"  median {sorted(ts)[len(ts) // 2]:.3f}   {g.last_arithmetic}   crc32 {zlib.crc32(y.cpu().numpy().tobytes()):08x}")
