"""step time of the headline workload with parts of the path taken out (4 convert() calls in flight):
what YAAPT and the bottleneck extractor cost on top of the generator"""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import satools_amd
from satools_amd import synthetic

model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1")
model.to("cuda")
model.eval()
seeds = list(range(32))
wav = synthetic.harm_batch(seeds).to("cuda")
f0 = model.get_f0(wav).clone()        # what convert() computes on-path (YAAPT), handed over in the second line
tg = synthetic.targets(model.spk, seeds)
streams = [torch.cuda.Stream() for _ in range(4)]
bn = model.get_bn(wav)
spk = model.get_spk_id(wav, tg)


def full(i):
    return model.convert(wav, target=tg)


def handed(i):
    model.set_f0(f0.clone())
    return model.convert(wav, target=tg)


def gen_only(i):
    return model._forward(f0.clone(), bn, spk)


for name, fn in (("convert (YAAPT + BN + generator)", full), ("F0 handed over (BN + generator)", handed), ("generator only", gen_only)):
    def step(i):
        with torch.cuda.stream(streams[i % 4]):
            return fn(i)
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(12):
        step(i)
    torch.cuda.synchronize()
    print(f"{name:36s} {(time.perf_counter() - t0) / 12 * 1e3:7.2f} ms/step")
