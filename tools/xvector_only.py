"""time the x-vector extractor (ECAPA-TDNN) on 5 s utterances at batch 1 and 32, both arithmetic settings"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import satools_amd
from satools_amd import synthetic, xvector

net = xvector.build()(num_speakers=10)
net.load_state_dict(synthetic.xvector_state(0, 10))
net = net.to("cuda")


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
    return e0.elapsed_time(e1) / n


ref = {}
for prec in ("f32", "f16x3"):
    net.precision = prec
    for B in (1, 32):
        wav = synthetic.harm_batch(list(range(B))).to("cuda")
        xv = net(wav)[1]
        if prec == "f32":
            ref[B] = xv
        print(f"{prec:6s} batch {B:2d}: {timed(lambda: net(wav)):.3f} ms   max |x-vector - f32 x-vector| {float((xv - ref[B]).abs().max()):.2e}")
