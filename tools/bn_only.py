"""time the bottleneck extractor (fbank -> TDNNF -> VQ) and YAAPT alone (batch 32 x 5 s)"""
import os
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import satools_amd
from satools_amd import synthetic

model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1")
model.to("cuda")
model.bn_extractor.vq_tie_sigmas = 0.0      # the extractor ALONE as configured (the near-tie guard would add its calibration and exact-f32 re-runs to the trace)
wav = synthetic.harm_batch(list(range(32))).to("cuda")


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
    return e0.elapsed_time(e1) / n


print("get_bn  %.3f ms" % timed(lambda: model.get_bn(wav)))
print("get_f0  %.3f ms" % timed(lambda: model.get_f0(wav)))
print("ASR forward (chain + xent log-likelihoods, 3280 pdfs)  %.3f ms" % timed(lambda: model.bn_extractor(wav.clone())))
