"""end-to-end rate of the `anonymize` data plane on one GPU: wav files in, anonymized PCM16 wav files out
(file reads, host->device, convert with F0 on path, device->host, crop, PCM16 encode, file writes).
  python tools/bench_pipeline.py [n_utts] [jobs] [batch]"""
import os
import sys
import tempfile
import time
import types

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.path.insert(0, "tests")
import numpy as np
import torch
import satools_amd
from satools_amd import pipeline as pl, synthetic
from pipeline_toy import write_wav

n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 512
jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
ragged = len(sys.argv) > 4 and sys.argv[4] == "ragged"     # utterance lengths 3.0 .. 5.0 s, all different
TAG = "hifigan_bn_tdnnf_600h_vq_48_v1"
with tempfile.TemporaryDirectory() as tmp:
    data = os.path.join(tmp, "data", "bench")
    os.makedirs(os.path.join(data, "clear"))
    scp, u2s = [], []
    base = [synthetic.harm_batch([i], 80000)[0].numpy().astype(np.float64) for i in range(32)]
    for i in range(n_utts):
        path = os.path.join(data, "clear", f"utt{i:05d}.wav")
        x = base[i % 32]
        write_wav(path, x[:80000 - (i * 997) % 32000] if ragged else x)
        scp.append(f"utt{i:05d} {path}\n")
        u2s.append(f"utt{i:05d} src{i % 40}\n")
    open(os.path.join(data, "wav.scp"), "w").writelines(scp)
    open(os.path.join(data, "utt2spk"), "w").writelines(u2s)
    model = satools_amd.load_model("synthetic:" + TAG)
    model.to("cuda")
    model.eval()
    settings = types.SimpleNamespace(model="-", f0_modification="", target_constant_spkid="?", results_dir="wav", batch_size=batch,
                                     data_loader_nj=8, new_datadir_suffix="_anon", device="cuda")
    wavscp = pl.read_wav_scp(os.path.join(data, "wav.scp"))
    warm = dict(list(wavscp.items())[:batch * jobs])
    pl.process_data(data, "random_per_spk", pl.split_dict(warm, jobs), settings, model=model)       # weights, tables
    # every process_data call sets up its streams, workspaces and page-locked staging buffers (~0.5 s per job):
    # a cost of the job, amortised over its thousands of utterances, so use enough of them here
    def timed(scp):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = pl.process_data(data, "random_per_spk", pl.split_dict(scp, jobs), settings, model=model)
        torch.cuda.synchronize()
        return n, time.perf_counter() - t0

    # a quarter of the utterances first: the difference of the two runs is the steady state (a job's set-up — streams, their workspaces,
    # page-locked buffers, the drain of the last batches — is paid once per process_data call)
    n_q, dt_q = timed(dict(list(wavscp.items())[:max(batch * jobs, n_utts // 4 // batch * batch)]))
    if os.environ.get("PROFILE") == "1":        # where the launching thread of the job spends its time
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        n, dt = timed(wavscp)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)
    else:
        n, dt = timed(wavscp)
    secs = sum((80000 - (i * 997) % 32000) if ragged else 80000 for i in range(n_utts)) / 16000.0
    if ragged:
        print(f"ragged lengths (3-5 s): {secs:.0f} s of audio -> {secs / dt:.0f} x real-time")
    print(f"{n} utterances x 5 s, batch {batch}, {jobs} jobs on one GPU: {dt:.2f} s wall = {n * 5.0 / dt:.0f} x real-time "
          f"(files in -> PCM16 files out; {dt / (n / batch) * 1e3:.1f} ms per batch)")
    if n > n_q:
        per_batch = (dt - dt_q) / ((n - n_q) / batch)
        print(f"  {n_q} utterances: {dt_q:.2f} s -> {per_batch * 1e3:.2f} ms per further batch = {batch * 5.0 / per_batch:.0f} x real-time in the steady state, "
              f"{dt - per_batch * n / batch:.2f} s of set-up and drain per job call")
