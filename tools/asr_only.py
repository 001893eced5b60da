"""time the full ASR forward (chain / xent log-likelihoods) of the fbank-tag bottleneck net on 32 x 5 s"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satools_amd
from satools_amd import synthetic
tag = sys.argv[1] if len(sys.argv) > 1 else "hifigan_bn_tdnnf_600h_vq_48_v1"
model = satools_amd.load_model("synthetic:" + tag); model.to("cuda"); model.eval()
net = model.bn_extractor
wav = synthetic.harm_batch(list(range(32))).to("cuda")
with torch.no_grad():
    for _ in range(3):
        out = net(wav)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = net(wav)
    e1.record(); torch.cuda.synchronize()
print(f"ASR forward, 32 x 5 s: {e0.elapsed_time(e1) / 10:.3f} ms; outputs {[tuple(o.shape) for o in out]}")
