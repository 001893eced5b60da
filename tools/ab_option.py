"""A/B of a sat_conv_set_option switch (or, with the prefix `gen:`, a sat_hifigan_set_option switch; with `attr:`, an attribute of the
Python generator object, which repacks the weights where they depend on it) on the generator forward
(batch 32 x 250 frames), interleaved rounds in one process: python tools/ab_option.py <option> <value_a> <value_b>"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satools_amd
from satools_amd import _lib

name, va, vb = sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3])
GEN, ATTR = name.startswith(b"gen:"), name.startswith(b"attr:")
name = name[4:] if GEN else name[5:] if ATTR else name
model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1")
model.to("cuda")
g = model.hifigan
x = torch.randn(32, g.imput_dim, 250, device="cuda")


def setopt(v):
    if ATTR:
        setattr(g, name.decode(), v)
    elif GEN:
        g(x[:1])                             # (the handle exists after the first forward)
        _lib.check(_lib.lib().sat_hifigan_set_option(g._handle, name, v), "set_option")
    else:
        _lib.check(_lib.lib().sat_conv_set_option(name, v), "set_option")


def run(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = g(x)[0]
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, y


res, outs = {va: [], vb: []}, {}
for v in (va, vb):
    setopt(v)
    run(3)
for rnd in range(5):
    for v in (va, vb):
        setopt(v)
        t, outs[v] = run(8)
        res[v].append(t)
for v in (va, vb):
    print(f"{name.decode()}={v}: generator forward ms, five rounds: " + " ".join(f"{t:.2f}" for t in res[v]) + f"   median {sorted(res[v])[2]:.2f}")
print("outputs bit-identical:", bool(torch.equal(outs[va], outs[vb])), " max abs diff %.2e" % float((outs[va] - outs[vb]).abs().max()))
