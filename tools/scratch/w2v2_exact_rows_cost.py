"""what the near-tie guard's second decision costs on the wav2vec2 tag: the exact-f32 run of 1 / 2 / 4 utterances alone on the device"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import satools_amd
from satools_amd import synthetic
model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_wav2vec2_vq_48_v1")
model.to("cuda"); model.eval()
ext = model.bn_extractor
wav = synthetic.harm_batch(list(range(32))).to("cuda")
ext.vq_tie_sigmas = 0.0
with torch.no_grad():
    feats = ext.features(wav)
    for n in (1, 2, 4, 8):
        rows = list(range(n))
        for _ in range(2):
            ext._exact_rows(rows, feats, wav)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ext._exact_rows(rows, feats, wav)
        torch.cuda.synchronize()
        print(f"exact-f32 run of {n} utterance(s): {1e3 * (time.perf_counter() - t0) / 3:.1f} ms")
    t0 = time.perf_counter()
    for _ in range(3):
        ext._extract_bn_private(wav)
    torch.cuda.synchronize()
    print(f"split-f16 get_bn of 32: {1e3 * (time.perf_counter() - t0) / 3:.1f} ms")
