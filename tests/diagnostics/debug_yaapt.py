"""diagnostic: compare the HIP YAAPT's intermediates with the oracle's, stage by stage"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import satools_amd
from satools_amd import f0 as f0_hip, synthetic
from oracle import yaapt as oy

OPTS = {"frame_length": 35.0, "frame_space": 20.0, "nccf_thresh1": 0.25, "tda_frame_length": 25.0}
torch.set_num_threads(1)
cases = sys.argv[1:] or ["harm0_16384", "rand0_16384", "rand0_80000"]
for name in cases:
    kind, n = name.split("_"); n = int(n)
    wav = synthetic.harm_batch([int(kind[4:])], n) if kind.startswith("harm") else synthetic.rand_batch(int(kind[4:]), 1, n)
    aux = {}
    ref = oy.yaapt_one(wav[0], OPTS, aux=aux)
    got, g = f0_hip.yaapt(wav.cuda(), OPTS, return_aux=True)
    got = got.cpu()[0]
    g = {k: v.cpu()[0] for k, v in g.items()}
    L = aux["filt"].numel()
    print("==", name, "final mismatches at", (got != ref).nonzero().flatten().tolist())
    def cmp(label, a, b):
        a, b = a.float(), b.float()
        d = (a - b).abs()
        nd = int((d > 0).sum())
        print(f"  {label:12s} maxabs {float(d.max()):.3e} (ref max {float(b.abs().max()):.3e}) n_diff {nd}" + (f" first at {np.argwhere((d>0).numpy())[:6].tolist()}" if 0 < nd <= 12 else ""))
    cmp("filt", g["filt"][0, :L], aux["filt"]); cmp("filt2", g["filt"][1, :L], aux["filt2"])
    cmp("energy", g["energy"], aux["energy"]); cmp("vuv", g["vuv"], aux["vuv"])
    cmp("cand_pitch", g["cand"][:4], aux["cand_pitch"]); cmp("cand_merit", g["cand"][4:], aux["cand_merit"])
    cmp("spec_pitch", g["spec_pitch"], aux["spec_pitch"]); print("   pitch_std", float(g["scal"][0]), float(aux["pitch_std"]))
    cmp("tp1", g["tp"][0], aux["tp1"][0]); cmp("tm1", g["tm"][0], aux["tm1"][0])
    cmp("tp2", g["tp"][1], aux["tp2"][0]); cmp("tm2", g["tm"][1], aux["tm2"][0])
