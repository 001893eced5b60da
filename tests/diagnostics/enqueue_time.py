"""host time to ENQUEUE one convert() of 32 x 5 s (no synchronisation inside the timed region): the launch-path cost
that bounds the data plane once reading and writing are off the launching thread"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import torch
import satools_amd
from satools_amd import synthetic

model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1")
model.to("cuda")
model.eval()
seeds = list(range(32))
wav = synthetic.harm_batch(seeds).to("cuda")
targets = synthetic.targets(model.spk, seeds)
lens = [80000] * 32
streams = [torch.cuda.Stream() for _ in range(4)]
with torch.no_grad():
    for name, f in (("convert", lambda: model.convert(wav, target=targets)),
                    ("convert_padded", lambda: model.convert_padded(wav, lens, targets))):
        for s in streams:
            with torch.cuda.stream(s):
                f()
        torch.cuda.synchronize()
        ts = []
        for i in range(12):
            with torch.cuda.stream(streams[i % 4]):
                t0 = time.perf_counter()
                f()
                ts.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        print(name, "enqueue ms per call:", " ".join(f"{t:.1f}" for t in ts))
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(8):
        with torch.cuda.stream(streams[i % 4]):
            model.convert_padded(wav, lens, targets)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
