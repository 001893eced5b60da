"""time the wav2vec2-tag bottleneck extractor alone (batch 32 x 5 s)"""
import sys
import torch
sys.path.insert(0, ".")
import satools_amd
from satools_amd import synthetic

model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_wav2vec2_vq_48_v1")
model.to("cuda")
wav = synthetic.harm_batch(list(range(32))).to("cuda")
for _ in range(2):
    model.get_bn(wav)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    model.get_bn(wav)
e1.record()
torch.cuda.synchronize()
print("wav2vec2-tag get_bn %.2f ms" % (e0.elapsed_time(e1) / 5))
