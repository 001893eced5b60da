"""machine time vs latency of a part of the path: calls per millisecond with 1, 2, 4, 8 streams in flight
python tools/overlap_probe.py [bn|f0|gen]"""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satools_amd
from satools_amd import synthetic

what = sys.argv[1] if len(sys.argv) > 1 else "bn"
model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1")
model.to("cuda")
model.eval()
seeds = list(range(32))
wav = synthetic.harm_batch(seeds).to("cuda")
tg = synthetic.targets(model.spk, seeds)
f0 = model.get_f0(wav).clone()
bn = model.get_bn(wav)
spk = model.get_spk_id(wav, tg)
fn = {"bn": lambda: model.get_bn(wav), "f0": lambda: model.get_f0(wav), "gen": lambda: model._forward(f0.clone(), bn, spk)}[what]
for ns in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    for i in range(2 * ns):
        with torch.cuda.stream(streams[i % ns]):
            fn()
    torch.cuda.synchronize()
    n = 8 * ns
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(streams[i % ns]):
            fn()
    torch.cuda.synchronize()
    print(f"{what}: {ns} streams in flight: {(time.perf_counter() - t0) / n * 1e3:7.3f} ms per call")
