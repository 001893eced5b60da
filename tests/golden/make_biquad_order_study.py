#!/usr/bin/env python3
"""Rounding-order study of the YAAPT band-limiting biquads (torchaudio is third-party and
absent, so its `lfilter` arithmetic cannot be pinned by running it).

Two FIR summation orders of oracle/biquad.py are run through the whole YAAPT restatement (oracle/yaapt.py):
  "torchaudio"         b' = b / a0 first, then torch's CPU conv1d order (FMA chain in tap order) — shipped;
  "raw_b_then_divide"  round 1's ((b2 x[t-2] + b1 x[t-1]) + b0 x[t]) / a0.
over every input of tests/golden/fx_f0.npz plus 100 extra seeded utterances (50 `harm`, 50 `rand`, 5 s), and the
number of F0 frames that differ is written to tests/golden/fx_biquad_order.json together with the number of
band-limited samples that differ (the filters DO differ in the last ulp; the question is whether any YAAPT
decision notices).  tests/test_oracle_yaapt.py asserts the committed counts on a subset it recomputes.

Run from the repo root:   python tests/golden/make_biquad_order_study.py      (about 2 minutes on 8 cores)
No reference import is needed: both variants are the repo's own restatement."""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path[:0] = [ROOT]
OPTS = {"frame_length": 35.0, "frame_space": 20.0, "nccf_thresh1": 0.25, "tda_frame_length": 25.0}


def inputs():
    names = sorted(np.load(os.path.join(GOLD, "fx_f0.npz")).files)
    names = [n for n in names if not n.endswith("_batch")]
    extra = [f"harm{s}_80000" for s in range(100, 150)] + [f"rand{s}_80000" for s in range(100, 150)]
    return names, extra


def wav_of(name):
    from satools_amd import synthetic
    kind, n = name.split("_")[0], int(name.split("_")[1])
    if kind.startswith("harm"):
        return synthetic.harm_batch([int(kind[4:])], n)
    return synthetic.rand_batch(int(kind[4:]), 1, n)


def one(name):
    import torch
    torch.set_num_threads(1)          # the reference's YAAPT setting (yaapt.py:27)
    from oracle import biquad
    from oracle import yaapt as oy
    w = wav_of(name)
    res = {"name": name}
    tracks = {}
    for order in biquad.ORDERS:
        aux = {}
        try:
            tracks[order] = oy.yaapt_one(w[0], OPTS, aux=aux, biquad_order=order).numpy()
            res.setdefault("filt", {})[order] = (aux["filt"].numpy(), aux["filt2"].numpy())
        except RuntimeError as e:     # no voiced frame: the reference fails the same way
            tracks[order] = None
    a, b = (tracks[o] for o in biquad.ORDERS)
    if a is None or b is None:
        return {"name": name, "frames": 0, "f0_frames_differ": 0 if (a is None) == (b is None) else -1,
                "filtered_samples_differ": None, "voiced": 0, "raised": True}
    fa, fb = res["filt"][biquad.ORDERS[0]], res["filt"][biquad.ORDERS[1]]
    return {"name": name, "frames": int(a.size), "f0_frames_differ": int((a != b).sum()),
            "filtered_samples_differ": int((fa[0] != fb[0]).sum() + (fa[1] != fb[1]).sum()),
            "max_abs_filter_diff": float(max(np.abs(fa[0] - fb[0]).max(), np.abs(fa[1] - fb[1]).max())),
            "voiced": int((a > 0).sum()), "raised": False}


def main():
    names, extra = inputs()
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        rows = pool.map(one, names + extra)
    tot = lambda rs, k: int(sum(r[k] or 0 for r in rs))
    fx_rows, ex_rows = rows[:len(names)], rows[len(names):]
    out = {"orders": ["torchaudio", "raw_b_then_divide"], "shipped": "torchaudio",
           "fx_f0_inputs": {"utterances": len(fx_rows), "frames": tot(fx_rows, "frames"),
                            "f0_frames_differ": tot(fx_rows, "f0_frames_differ"),
                            "filtered_samples_differ": tot(fx_rows, "filtered_samples_differ")},
           "extra_100": {"utterances": len(ex_rows), "frames": tot(ex_rows, "frames"), "voiced_frames": tot(ex_rows, "voiced"),
                         "f0_frames_differ": tot(ex_rows, "f0_frames_differ"),
                         "filtered_samples_differ": tot(ex_rows, "filtered_samples_differ"),
                         "raised_in_both": int(sum(r["raised"] for r in ex_rows))},
           "per_input": rows}
    json.dump(out, open(os.path.join(GOLD, "fx_biquad_order.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "per_input"}, indent=1))


if __name__ == "__main__":
    main()
