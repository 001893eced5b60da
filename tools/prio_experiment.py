"""does HIP stream priority keep YAAPT's LDS-heavy FFT blocks out of the generator's way?  Four convert() jobs in flight;
YAAPT of each job on a side stream of the same / of the lowest priority, its F0 handed over (model.set_f0)"""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satools_amd
from satools_amd import synthetic

model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1")
model.to("cuda")
model.eval()
seeds = list(range(32))
wav = synthetic.harm_batch(seeds).to("cuda")
tg = synthetic.targets(model.spk, seeds)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range (least, greatest):", lo, hi)
NJ = 4


def run(name, job_prio, side_prio):
    jobs = [torch.cuda.Stream(priority=job_prio) for _ in range(NJ)]
    sides = [torch.cuda.Stream(priority=side_prio) for _ in range(NJ)] if side_prio is not None else None

    def step(i):
        js = jobs[i % NJ]
        if sides is None:
            with torch.cuda.stream(js):
                return model.convert(wav, target=tg)
        ss = sides[i % NJ]
        ss.wait_stream(js)
        with torch.cuda.stream(ss):
            f0 = model.get_f0(wav)
        js.wait_stream(ss)
        with torch.cuda.stream(js):
            f0.record_stream(js)
            model.set_f0(f0)
            return model.convert(wav, target=tg)
    for i in range(8):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(24):
        step(i)
    torch.cuda.synchronize()
    print(f"{name:60s} {(time.perf_counter() - t0) / 24 * 1e3:7.2f} ms/step")


for rep in range(2):
    run("convert() on the job stream (baseline)", 0, None)
    run("YAAPT on a side stream, same priority", 0, 0)
    run("YAAPT on a side stream, lowest priority", 0, lo)
    run("jobs at the highest priority, YAAPT side stream lowest", hi, lo)
