#!/usr/bin/env python3
"""Cross-check of the RECURSION half of the YAAPT band-limiting biquads against a second, independent
implementation that IS installed here, and a study of what such a check can and cannot tell apart.

torchaudio (third party, absent, version unpinned by the reference: satools/satools/hifigan/yaapt.py:46-47 ->
torchaudio.functional.lowpass_biquad / highpass_biquad -> lfilter) cannot be run.  oracle/biquad.py restates its published
algorithm; the FIR half is pinned bit for bit against torch's own conv1d (tests/test_oracle_yaapt.py).  For the IIR half:

 1. STRUCTURE.  scipy.signal.lfilter (1.15, an unrelated code base: direct form II transposed in C) on the same RBJ coefficients,
    in float64 ("truth": the same recurrence without rounding) and in float32, against the oracle's float32 direct-form-I loop,
    over the band-limiting of x and x^2 (what YAAPT filters) of 20 utterances.  A transcription error of the recurrence — a1 / a2
    swapped, a sign, coefficients not normalised by a0, the high-pass numerator — is O(1e-2 .. 1) of the signal; rounding is
    O(1e-6).  So the check PINS the structure and the coefficients.
 2. ROUNDING ORDER.  Four evaluation orders of the f32 recursion (oracle/biquad.py IIR_ORDERS: the published C++ loop's
    multiply-subtract with the a2 term first = shipped; a1 term first; the same loop with FMA contraction; products summed
    first): how many samples differ between them, how far each lies from the float64 truth, and — through the whole YAAPT
    restatement — how many F0 frames move.  All four lie in the same envelope around the truth (and so does scipy's float32 result,
    which is yet another order): an independent implementation CANNOT tell them apart, only torchaudio's own binary could.  What
    the study adds is the price of being wrong: the F0 frames that change.

Writes tests/golden/fx_biquad_iir.json; tests/test_oracle_yaapt.py recomputes a subset.   python tests/golden/make_biquad_iir_crosscheck.py
(about 4 minutes on 8 cores).  No reference import is needed."""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path[:0] = [ROOT]
OPTS = {"frame_length": 35.0, "frame_space": 20.0, "nccf_thresh1": 0.25, "tda_frame_length": 25.0}
FILTERS = (("lp", 50.0), ("hp", 1500.0))


def wav_of(name):
    from satools_amd import synthetic
    kind, n = name.split("_")[0], int(name.split("_")[1])
    if kind.startswith("harm"):
        return synthetic.harm_batch([int(kind[4:])], n)
    return synthetic.rand_batch(int(kind[4:]), 1, n)


def scipy_lfilter(x, kind, cutoff, dtype):
    from scipy.signal import lfilter
    from oracle import biquad
    b, a = biquad.coeffs(kind, 16000, cutoff)
    return lfilter((b / a[0]).astype(dtype), (a / a[0]).astype(dtype), x.astype(dtype))


def structure_one(name):
    """max |oracle f32 - scipy| per filter on x and x^2 of one utterance, unclamped; and the same for deliberately wrong recurrences"""
    from oracle import biquad
    x = wav_of(name)[0].numpy()
    out = {"name": name}
    for sig_name, sig in (("x", x), ("x2", (x * x).astype(np.float32))):
        for kind, cutoff in FILTERS:
            ours = biquad.biquad(sig, kind, 16000, cutoff, clamp=False).astype(np.float64)
            t64 = scipy_lfilter(sig, kind, cutoff, np.float64)
            s32 = scipy_lfilter(sig, kind, cutoff, np.float32).astype(np.float64)
            scale = float(np.abs(t64).max())
            key = f"{sig_name}_{kind}"
            out[key] = {"scale": scale, "ours_vs_f64": float(np.abs(ours - t64).max()), "scipy32_vs_f64": float(np.abs(s32 - t64).max()),
                        "ours_vs_scipy32": float(np.abs(ours - s32).max())}
            # wrong transcriptions, in float64 (their distance to the truth is what a structure error looks like)
            from scipy.signal import lfilter
            b, a = biquad.coeffs(kind, 16000, cutoff)
            bn, an = (b / a[0]).astype(np.float64), (a / a[0]).astype(np.float64)
            wrong = {"a1_a2_swapped": lfilter(bn, np.array([1.0, an[2], an[1]]), sig.astype(np.float64)),
                     "a_not_normalised": lfilter(bn, np.array([1.0, float(a[1]), float(a[2])]), sig.astype(np.float64)),
                     "other_numerator": lfilter(((biquad.coeffs("hp" if kind == "lp" else "lp", 16000, cutoff)[0]) / a[0]).astype(np.float64), an, sig.astype(np.float64))}
            out[key]["wrong"] = {k: float(min(np.abs(np.nan_to_num(v, nan=1e30, posinf=1e30, neginf=-1e30) - t64).max(), 1e30)) for k, v in wrong.items()}
    return out


def order_one(name):
    """the four recursion orders on the band-limited x and x^2 of one utterance (YAAPT's own two signals, with its zero padding), and
    the F0 tracks they lead to"""
    import torch
    torch.set_num_threads(1)
    from oracle import biquad
    from oracle import yaapt as oy
    w = wav_of(name)
    res = {"name": name, "orders": {}}
    tracks, filt = {}, {}
    for o in biquad.IIR_ORDERS:
        aux = {}
        try:
            tracks[o] = oy.yaapt_one(w[0], OPTS, aux=aux, biquad_iir=o).numpy()
            filt[o] = np.concatenate([aux["filt"].numpy(), aux["filt2"].numpy()])
        except RuntimeError:
            tracks[o] = None
    ref_t, ref_f = tracks["torchaudio"], filt.get("torchaudio")
    for o in biquad.IIR_ORDERS:
        if tracks[o] is None or ref_t is None:
            res["orders"][o] = {"raised": True}
            continue
        res["orders"][o] = {"f0_frames_differ": int((tracks[o] != ref_t).sum()), "samples_differ": int((filt[o] != ref_f).sum()),
                            "max_abs_diff": float(np.abs(filt[o].astype(np.float64) - ref_f.astype(np.float64)).max())}
    res["frames"] = 0 if ref_t is None else int(ref_t.size)
    res["samples"] = 0 if ref_f is None else int(ref_f.size)
    return res


def main():
    fx = sorted(np.load(os.path.join(GOLD, "fx_f0.npz")).files)
    fx = [n for n in fx if not n.endswith("_batch")]
    struct_names = [f"harm{s}_80000" for s in range(200, 210)] + [f"rand{s}_80000" for s in range(200, 210)]
    order_names = fx + [f"harm{s}_80000" for s in range(100, 120)] + [f"rand{s}_80000" for s in range(100, 120)]
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        srows = pool.map(structure_one, struct_names)
        orows = pool.map(order_one, order_names)
    from oracle import biquad
    keys = [f"{s}_{k}" for s in ("x", "x2") for k, _ in FILTERS]
    structure = {}
    for key in keys:
        rel = lambda f: max(r[key][f] / max(r[key]["scale"], 1e-30) for r in srows)
        structure[key] = {"ours_vs_f64_rel": rel("ours_vs_f64"), "scipy32_vs_f64_rel": rel("scipy32_vs_f64"), "ours_vs_scipy32_rel": rel("ours_vs_scipy32"),
                          "wrong_transcriptions_rel_min": {w: min(r[key]["wrong"][w] / max(r[key]["scale"], 1e-30) for r in srows) for w in srows[0][key]["wrong"]}}
    orders = {}
    for o in biquad.IIR_ORDERS:
        rows = [r["orders"][o] for r in orows if not r["orders"][o].get("raised")]
        orders[o] = {"f0_frames_differ": int(sum(r["f0_frames_differ"] for r in rows)), "samples_differ": int(sum(r["samples_differ"] for r in rows)),
                     "max_abs_diff": float(max(r["max_abs_diff"] for r in rows)),
                     "utterances_with_moved_frames": [r["name"] for r in orows if not r["orders"][o].get("raised") and r["orders"][o]["f0_frames_differ"]]}
    out = {"scipy": __import__("scipy").__version__, "shipped": "torchaudio", "iir_orders": list(biquad.IIR_ORDERS),
           "structure": {"utterances": len(srows), "per_filter": structure,
                         "reading": "relative to the filtered signal's peak; rounding differences are <= 1e-4 of it, every wrong recurrence >= 1e-2"},
           "rounding_order": {"utterances": len(orows), "frames": int(sum(r["frames"] for r in orows)), "samples": int(sum(r["samples"] for r in orows)),
                              "vs_shipped": orders,
                              "reading": "an independent implementation agrees with every order to the same rounding envelope: it pins the "
                                         "structure, not the order; only torchaudio's own binary could pin the order"},
           "per_input": [{"name": r["name"], **{o: r["orders"][o].get("f0_frames_differ") for o in biquad.IIR_ORDERS}} for r in orows]}
    with open(os.path.join(GOLD, "fx_biquad_iir.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("structure", "rounding_order")}, indent=1)[:6000])


if __name__ == "__main__":
    main()
