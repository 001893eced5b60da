"""what separates the batch job's 11.7 ms per batch from bench.py's 8.8?  four streams round-robin from one thread (bench.py's loop) with the
data plane's pieces added one at a time"""
import os, sys, time, tempfile
from concurrent.futures import ThreadPoolExecutor
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.set_num_threads(1)
import satools_amd
from satools_amd import ops, synthetic, pipeline as pl
model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1"); model.to("cuda"); model.eval()
B, n, J = 32, 80000, 4
w = synthetic.harm_batch(list(range(B)), n)
pcm = np.clip(np.rint(w.numpy().astype(np.float64) * 32768), -32768, 32767).astype(np.int16)
tg = synthetic.targets(model.spk, list(range(B)))
lens = [n] * B
d32 = w.cuda()
streams = [torch.cuda.Stream() for _ in range(J)]
pin_in = [[torch.empty(B, n, dtype=torch.int16, pin_memory=True) for _ in range(3)] for _ in range(J)]
pin_out = [[torch.empty(B, 1, n + 1, dtype=torch.int16, pin_memory=True) for _ in range(3)] for _ in range(J)]
tmp = tempfile.mkdtemp()
writers = ThreadPoolExecutor(4)
def write(host, ev, k):
    ev.synchronize()
    a = host.numpy()
    for i in range(B):
        pl.write_riff_pcm16(os.path.join(tmp, f"o{k % 8}_{i}.wav"), a[i, :, :n], 16000)
def loop(mode, steps=40):
    futs = []
    with torch.no_grad():
        for it in range(steps + 8):
            if it == 8:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            j, r = it % J, (it // J) % 3
            with torch.cuda.stream(streams[j]):
                if "h2d" in mode:
                    np.copyto(pin_in[j][r].numpy(), pcm)
                    x = ops.pcm16_to_f32(pin_in[j][r].to("cuda", non_blocking=True))
                else:
                    x = d32
                y = model.convert_padded(x, lens, tg) if "padded" in mode else model.convert(x, target=tg)
                if "d2h" in mode:
                    pin_out[j][r].copy_(ops.pcm16_from_f32(y), non_blocking=True)
                    ev = torch.cuda.Event(); ev.record()
                    if "write" in mode:
                        futs.append(writers.submit(write, pin_out[j][r], ev, it))
        for f in futs: f.result()
        torch.cuda.synchronize()
    print(f"{mode:32s} {(time.perf_counter() - t0) / steps * 1e3:7.2f} ms per batch", flush=True)
lens_eq = lens
lens_rg = [n - (i * 997) % 32000 for i in range(B)]
for m in ("convert", "padded", "padded ragged", "padded h2d d2h write", "padded ragged h2d d2h write", "padded"):
    lens = lens_rg if "ragged" in m else lens_eq
    loop(m)
lens = lens_eq

# ---- with reader threads like process_data's (8 file threads, one batch-assembling thread per job, two batches ahead)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from pipeline_toy import write_wav
paths = []
for i in range(B):
    p = os.path.join(tmp, f"in{i}.wav"); write_wav(p, w[i].numpy().astype(np.float64)); paths.append(p)
file_pool = ThreadPoolExecutor(8); readers = ThreadPoolExecutor(J)
def read_one(p):
    a, sr = pl.read_pcm16_mono(p)
    return {"utid": p, "pcm": a, "audio": None, "f0": None, "freq": sr}
def read_batch(serial):
    if serial:
        return pl.collate_pcm16([read_one(p) for p in paths])
    return pl.collate_pcm16(list(file_pool.map(read_one, paths)))
def loop2(mode, steps=40):
    futs = []
    pre = [[readers.submit(read_batch, "serial" in mode) for _ in range(2)] for _ in range(J)]
    with torch.no_grad():
        for it in range(steps + 8):
            if it == 8:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            j, r = it % J, (it // J) % 3
            audio = pre[j].pop(0).result()[0]
            pre[j].append(readers.submit(read_batch, "serial" in mode))
            with torch.cuda.stream(streams[j]):
                np.copyto(pin_in[j][r].numpy(), audio)
                x = ops.pcm16_to_f32(pin_in[j][r].to("cuda", non_blocking=True))
                y = model.convert_padded(x, lens, tg)
                pin_out[j][r].copy_(ops.pcm16_from_f32(y), non_blocking=True)
                ev = torch.cuda.Event(); ev.record()
                if "write" in mode:
                    futs.append(writers.submit(write, pin_out[j][r], ev, it))
        for f in futs: f.result()
        torch.cuda.synchronize()
    print(f"{mode:32s} {(time.perf_counter() - t0) / steps * 1e3:7.2f} ms per batch", flush=True)
for m in ("readers(pool)", "readers(pool) write", "readers(serial)", "readers(serial) write", "readers(pool) write"):
    loop2(m)
import sys as _s
for iv in (0.0005, 0.05):
    _s.setswitchinterval(iv)
    print("switch interval", iv)
    loop2("readers(pool) write")
