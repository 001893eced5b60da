"""the fbank-tag bottleneck extractor (model.get_bn) alone on 32 x 5 s — run under rocprofv3 --kernel-trace --stats"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satools_amd
from satools_amd import synthetic

tag = sys.argv[1] if len(sys.argv) > 1 else "hifigan_bn_tdnnf_600h_vq_48_v1"
model = satools_amd.load_model("synthetic:" + tag)
model.to("cuda")
model.bn_extractor.vq_tie_sigmas = 0.0      # the extractor ALONE as configured (the near-tie guard would add its calibration and exact-f32 re-runs to the trace)
model.eval()
wav = synthetic.harm_batch(list(range(32))).to("cuda")
for _ in range(3):
    model.get_bn(wav)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    model.get_bn(wav)
e1.record()
torch.cuda.synchronize()
print(f"get_bn, batch 32 x 5 s: {e0.elapsed_time(e1) / 20:.3f} ms per batch")
