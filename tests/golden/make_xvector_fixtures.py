#!/usr/bin/env python3
"""Golden fixtures for the x-vector extractor (SURVEY row aX / §8 f3): builds the REFERENCE's ECAPA-TDNN Net through
its own model-config `build(args)` (egs/asv/voxceleb/local/tuning/ecapa_tdnn.py), loads the seeded synthetic state
dict of satools_amd.synthetic.xvector_state strictly (pins key names and shapes), and stores its outputs on seeded
synthetic utterances.  torchaudio's MelSpectrogram is the stand-in of tests/golden/refstub (oracle/melspec.py:
third party, parity unpinned).     python tests/golden/make_xvector_fixtures.py"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf   # noqa: E402


def main():
    ref = mf.setup_reference()
    import satools_amd   # noqa: F401
    from satools_amd import synthetic
    m = mf.exec_config(os.path.join(ref, "egs/asv/voxceleb/local/tuning/ecapa_tdnn.py"))
    net = m.build(types.SimpleNamespace(fine_tune="false"))(num_speakers=10)
    net.eval()
    sd = synthetic.xvector_state(0, 10)
    missing = net.load_state_dict(sd, strict=True)
    print("strict load ok:", missing)
    json.dump({k: list(v.shape) for k, v in net.state_dict().items()}, open(os.path.join(HERE, "state_dict_keys_xvector.json"), "w"), indent=0)
    out = {}
    hooks = {}
    sn = net.sequence_network
    for name, mod in (("feats", net.preprocessor), ("layer1", sn.layer1), ("seq", sn), ("pooled", net.stat_pooling)):
        mod.register_forward_hook(lambda _m, _i, o, name=name: hooks.__setitem__(name, o.detach()))
    for tag, seeds, n in (("harm0_16000", [0], 16000), ("harm3_48000", [3], 48000), ("harm7_24123", [7], 24123)):
        wav = synthetic.harm_batch(seeds, n)[0]
        with torch.no_grad():
            (_loss, _s), xv = net(wav)
        out[tag + "/xvector"] = xv.numpy()
        out[tag + "/feats"] = hooks["feats"].numpy()
        out[tag + "/layer1_sub8"] = hooks["layer1"][:, ::8, ::4].numpy()
        out[tag + "/seq_sub16"] = hooks["seq"][:, ::16, ::4].numpy()
        out[tag + "/pooled"] = hooks["pooled"].numpy()
    np.savez_compressed(os.path.join(HERE, "fx_xvector.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
