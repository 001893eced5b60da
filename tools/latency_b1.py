"""latency of one convert() call at small batches (serving one utterance at a time): wall time per call with a device synchronisation after
each, host time until the call returns, and the sum of its kernels' durations is in `tools/prof_trace_bench.sh`-style traces"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satools_amd
from satools_amd import synthetic
tag = sys.argv[1] if len(sys.argv) > 1 else "hifigan_bn_tdnnf_600h_vq_48_v1"
model = satools_amd.load_model("synthetic:" + tag); model.to("cuda"); model.eval()
for B, secs in ((1, 5), (1, 1), (1, 20), (4, 5), (32, 5)):
    n = 16000 * secs
    wav = synthetic.harm_batch(list(range(B)), n).to("cuda")
    tg = synthetic.targets(model.spk, list(range(B)))
    with torch.no_grad():
        for _ in range(5):
            model.convert(wav, target=tg)
        torch.cuda.synchronize()
        ts, hs = [], []
        for _ in range(30):
            t0 = time.perf_counter()
            model.convert(wav, target=tg)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0); hs.append(t1 - t0)
    ts.sort(); hs.sort()
    print(f"batch {B:2d} x {secs:2d} s: {ts[len(ts) // 2] * 1e3:6.2f} ms per call (min {ts[0] * 1e3:.2f}; call returns after {hs[len(hs) // 2] * 1e3:.2f} ms) = "
          f"{B * secs / ts[len(ts) // 2]:7.0f} x real-time", flush=True)
