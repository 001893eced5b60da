"""The shader clock the chip holds while the path runs: a one-wave probe kernel (sat_clock_probe) samples
(s_memtime, s_memrealtime) every 200 us on its own stream beside (a) nothing, (b) the headline convert() steps,
(c) wav2vec2-tag steps, 4 jobs in flight as in bench.py.  Prints the median / min / max clock per phase."""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import satools_amd
from satools_amd import _lib, synthetic
from satools_amd._lib import check, lib, ptr

dev = torch.device("cuda")
probe_stream = torch.cuda.Stream()
N, PERIOD = 1500, 200          # 0.3 s of samples


def probe():
    buf = torch.zeros(2 * N, dtype=torch.int64, device=dev)
    with torch.cuda.stream(probe_stream):
        check(lib().sat_clock_probe(ptr(buf), N, PERIOD, probe_stream.cuda_stream), "sat_clock_probe")
    return buf


def clocks(buf):
    a = buf.cpu().numpy().reshape(-1, 2)
    a = a[a[:, 1] > 0]
    d = np.diff(a, axis=0)
    ghz = d[:, 0] / d[:, 1] * 0.1
    return ghz


def report(name, ghz):
    print(f"{name:44s} clock GHz: median {np.median(ghz):.2f}  p10 {np.percentile(ghz, 10):.2f}  p90 {np.percentile(ghz, 90):.2f}  "
          f"min {ghz.min():.2f}  max {ghz.max():.2f}  ({len(ghz)} samples of {PERIOD} us)")


b = probe()
torch.cuda.synchronize()
report("idle chip (probe alone)", clocks(b))
for tag in ("hifigan_bn_tdnnf_600h_vq_48_v1", "hifigan_bn_tdnnf_wav2vec2_vq_48_v1"):
    model = satools_amd.load_model("synthetic:" + tag)
    model.to(dev)
    model.eval()
    seeds = list(range(32))
    wav = synthetic.harm_batch(seeds).to(dev)
    tg = synthetic.targets(model.spk, seeds)
    streams = [torch.cuda.Stream() for _ in range(4)]
    def step(i):
        with torch.cuda.stream(streams[i % 4]):
            model.convert(wav, target=tg)
    for i in range(8):
        step(i)
    torch.cuda.synchronize()
    n_steps = 60 if "600h" in tag else 20
    for i in range(n_steps // 2):         # load first, then start sampling inside the sustained part
        step(i)
    b = probe()
    for i in range(n_steps):
        step(i)
    torch.cuda.synchronize()
    report(f"{tag[:36]}: {n_steps} steps, 4 jobs", clocks(b))
    del model
    torch.cuda.empty_cache()
