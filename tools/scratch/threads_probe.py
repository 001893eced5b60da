"""bench.py's loop (one thread, four streams round-robin) against one launching thread per job (what the reference's
jobs_per_compute_device are: separate processes, each blocking in its own convert())"""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.set_num_threads(1)
import satools_amd
from satools_amd import synthetic
model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1"); model.to("cuda"); model.eval()
B = 32
wav = synthetic.harm_batch(list(range(B))).cuda()
tg = synthetic.targets(model.spk, list(range(B)))
def single(jobs, steps):
    streams = [torch.cuda.Stream() for _ in range(jobs)]
    with torch.no_grad():
        for i in range(2 * jobs):
            with torch.cuda.stream(streams[i % jobs]): model.convert(wav, target=tg)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            with torch.cuda.stream(streams[i % jobs]): model.convert(wav, target=tg)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
def threaded(jobs, steps):
    streams = [torch.cuda.Stream() for _ in range(jobs)]
    def work(j, n):
        with torch.no_grad(), torch.cuda.stream(streams[j]):
            for _ in range(n): model.convert(wav, target=tg)
    def run(n):
        ts = [threading.Thread(target=work, args=(j, n)) for j in range(jobs)]
        for t in ts: t.start()
        for t in ts: t.join()
    run(2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(steps // jobs)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (steps // jobs * jobs) * 1e3
for rnd in range(3):
    for jobs in (4, 6):
        print(f"jobs {jobs}: one thread {single(jobs, 240):6.2f} ms per step   a thread per job {threaded(jobs, 240):6.2f} ms per step", flush=True)
