"""run only the generator forward (batch 32 x 250 frames) a few times: for rocprofv3 --kernel-trace --stats"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import satools_amd

model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1")
model.to("cuda")
g = model.hifigan
x = torch.randn(32, g.imput_dim, 250, device="cuda")
for _ in range(3):
    g(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    g(x)
e1.record()
torch.cuda.synchronize()
print("generator forward", e0.elapsed_time(e1) / 10, "ms")
