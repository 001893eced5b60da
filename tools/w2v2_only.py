"""time the wav2vec2-tag bottleneck extractor alone (batch 32 x 5 s); with `ab`: the 1x1 GEMM kernels interleaved in
one process (k1_gemm = 1: 128 x 128 register-staged kernel, 2: LDS-DMA ring kernel, 3: the ring on the 16x16x32 MFMA shape), five rounds of five forwards each"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satools_amd
from satools_amd import _lib, synthetic

model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_wav2vec2_vq_48_v1")
model.to("cuda")
model.bn_extractor.vq_tie_sigmas = 0.0      # the extractor ALONE as configured (the near-tie guard would add its calibration and exact-f32 re-runs to the trace)
wav = synthetic.harm_batch(list(range(32))).to("cuda")


def run(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        model.get_bn(wav)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


run(2)
if len(sys.argv) > 1 and sys.argv[1] == "ab":
    res = {1: [], 2: [], 3: []}
    for rnd in range(5):
        for opt in (1, 2, 3):
            _lib.check(_lib.lib().sat_conv_set_option(b"k1_gemm", opt), "set_option")
            res[opt].append(run(5))
    for opt in (1, 2, 3):
        print(f"k1_gemm={opt}: get_bn ms per batch, five rounds: " + " ".join(f"{v:.2f}" for v in res[opt]) + f"   median {sorted(res[opt])[2]:.2f}")
    _lib.check(_lib.lib().sat_conv_set_option(b"k1_gemm", 3), "set_option")
elif len(sys.argv) > 1 and sys.argv[1] == "pitch":
    # interleaved A/B of the residual-stream pitch (wav2vec2.py: residual_pitch)
    res, outs = {0: [], 1: []}, {}
    ext = model.bn_extractor
    for rnd in range(5):
        for v in (0, 1):
            ext.residual_pitch = v
            res[v].append(run(5))
            outs[v] = model.get_bn(wav).clone()
    for v in (0, 1):
        print(f"residual_pitch={v}: get_bn ms per batch, five rounds: " + " ".join(f"{t:.2f}" for t in res[v]) + f"   median {sorted(res[v])[2]:.2f}")
    print("outputs bit-identical:", bool(torch.equal(outs[0], outs[1])))
elif len(sys.argv) > 2 and sys.argv[1] == "opt":
    # interleaved A/B of a sat_conv_set_option switch: python tools/w2v2_only.py opt <name> <value_a> <value_b>
    name, va, vb = sys.argv[2].encode(), int(sys.argv[3]), int(sys.argv[4])
    res, outs = {va: [], vb: []}, {}
    for rnd in range(5):
        for v in (va, vb):
            _lib.check(_lib.lib().sat_conv_set_option(name, v), "set_option")
            res[v].append(run(5))
            outs[v] = model.get_bn(wav).clone()
    for v in (va, vb):
        print(f"{name.decode()}={v}: get_bn ms per batch, five rounds: " + " ".join(f"{t:.2f}" for t in res[v]) + f"   median {sorted(res[v])[2]:.2f}")
    print("outputs bit-identical:", bool(torch.equal(outs[va], outs[vb])))
else:
    print("wav2vec2-tag get_bn %.2f ms" % run(5))
