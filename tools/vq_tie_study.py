"""How wide must the near-tie window of the VQ guard be?  (asrbn._TdnnfBase.vq_tie_sigmas)

For the 536 utterances per tag of tests/test_hip_robust.py::test_vq_flip_rate_of_the_default_arithmetic: the extractor as configured
(split-f16) and on its exact-f32 kernels; per frame the gap of the default arithmetic's two best distances in units of
2 sigma_c |e_a - e_a'| (sigma_c = sigma_rel |z_t| / sqrt(D), sigma_rel calibrated as _tie_guard does).  Prints the normalised gap of
every flipped frame and, for K = 2 .. 12, the share of utterances a window of K sigma flags.

    python tools/vq_tie_study.py [fbank|w2v2|both]
"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

TAGS = {"fbank": "hifigan_bn_tdnnf_600h_vq_48_v1", "w2v2": "hifigan_bn_tdnnf_wav2vec2_vq_48_v1"}


def long_batch(seeds, n):
    from satools_amd import synthetic
    rows = []
    for sd in seeds:
        parts, k = [], 0
        while sum(p.shape[0] for p in parts) < n:
            parts.append(synthetic.harm_batch([sd * 100 + k], 80000)[0])
            k += 1
        rows.append(torch.cat(parts)[:n])
    return torch.stack(rows)


def main():
    import satools_amd
    from satools_amd import synthetic
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    dev = "cuda:0"
    for name, tag in TAGS.items():
        if which not in ("both", name):
            continue
        model = satools_amd.load_model("synthetic:" + tag)
        model.to(dev)
        model.eval()
        ext = model.bn_extractor
        ext.vq_tie_sigmas = 1.0
        pair, scale = ext._tie_guard(torch.device(dev))
        sigma_rel = ext._tie[2]
        D = ext._tie[3]
        print(f"{tag}: sigma_rel = {sigma_rel:.3e}, D = {D}")
        sets = [("5 s", [synthetic.harm_batch(list(range(3000 + 32 * i, 3032 + 32 * i)), 80000) for i in range(16)]),
                ("20 s", [long_batch(list(range(4000 + 4 * i, 4004 + 4 * i)), 20 * 16000) for i in range(4)]),
                ("35 s", [long_batch(list(range(5000 + 2 * i, 5002 + 2 * i)), 35 * 16000) for i in range(4)])]
        Ks = [2, 3, 4, 5, 6, 8, 10, 12]
        for sname, batches in sets:
            n_utt, flagged = 0, {k: 0 for k in Ks}
            flips_norm = []
            for wav in batches:
                wav = wav.to(dev)
                _, (z, idx, dist) = ext.extract_bn(wav.clone(), want_aux=True)
                with ext._exact(ext):
                    _, (z32, idx32, _) = ext.extract_bn(wav.clone(), want_aux=True)
                B, T, n = dist.shape
                d2, order = torch.sort(dist.double(), dim=2)
                gap = d2[:, :, 1] - d2[:, :, 0]
                a, b = order[:, :, 0], order[:, :, 1]
                pd = pair.double()[a, b]
                zn = z.double().norm(dim=1)                                       # [B, T]
                unit = 2.0 * sigma_rel * zn / math.sqrt(D) * pd
                norm = gap / unit.clamp_min(1e-300)
                flip = idx.long() != idx32.long()
                flips_norm += norm[flip].tolist()
                n_utt += B
                for k in Ks:
                    flagged[k] += int((norm <= k).any(dim=1).sum())
            print(f"  {sname}: {n_utt} utterances; flips at normalised gaps {[round(v, 2) for v in sorted(flips_norm)]}")
            print("     flagged share by K: " + ", ".join(f"K={k}: {100.0 * flagged[k] / n_utt:.1f} %" for k in Ks))


if __name__ == "__main__":
    main()
