"""host-side costs of one batch of the anonymize data plane (32 x 5 s): file read + decode + collate, page-locked
staging + H2D, D2H, PCM16 encode + file write.  Run on the GPU box from the repo root."""
import os
import sys
import tempfile
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import torch
import satools_amd
from satools_amd import pipeline as pl, synthetic
from pipeline_toy import write_wav


def t(f, n=10):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3


with tempfile.TemporaryDirectory() as tmp:
    paths = []
    for i in range(32):
        p = os.path.join(tmp, f"u{i}.wav")
        write_wav(p, synthetic.harm_batch([i], 80000)[0].numpy().astype(np.float64))
        paths.append(p)

    def read():
        return pl.collate_fn([{"utid": str(i), "audio": pl.load_wav_from_scp(p)[0], "f0": None, "freq": 16000} for i, p in enumerate(paths)])

    audio = read()[0]
    print("read + decode + collate of 32 files: %.2f ms" % t(read))
    pin = torch.empty(audio.numel(), dtype=torch.float32, pin_memory=True).view(audio.shape)

    def h2d():
        pin.copy_(audio)
        x = pin.to("cuda", non_blocking=True)
        torch.cuda.synchronize()
        return x

    print("pin copy + H2D (10 MB): %.2f ms" % t(h2d))
    y = torch.randn(32, 1, 80001, device="cuda").clamp(-1, 1)
    host = torch.empty(y.numel(), dtype=torch.float32, pin_memory=True).view(y.shape)

    def d2h():
        host.copy_(y, non_blocking=True)
        torch.cuda.synchronize()

    print("D2H (10 MB): %.2f ms" % t(d2h))

    def write():
        for i in range(32):
            pl.save_pcm16(os.path.join(tmp, f"o{i}.wav"), host[i][:, :80000], 16000)

    print("PCM16 encode + write of 32 files: %.2f ms" % t(write))
