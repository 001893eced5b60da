"""x-vector extractor (SURVEY row aX / §8 f3): the CPU oracle against outputs of the reference's own ECAPA-TDNN Net
(tests/golden/make_xvector_fixtures.py), and the host parameter tree against the reference's state-dict keys."""
import json
import os

import numpy as np
import torch

import satools_amd   # noqa: F401
from satools_amd import synthetic, xvector
from oracle import xvector as ox

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_state_dict_keys_are_the_references():
    want = json.load(open(os.path.join(GOLD, "state_dict_keys_xvector.json")))
    net = xvector.build()(num_speakers=10)
    have = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert have == want
    net.load_state_dict(synthetic.xvector_state(0, 10), strict=True)


def test_oracle_matches_reference_outputs():
    fx = np.load(os.path.join(GOLD, "fx_xvector.npz"))
    sd = synthetic.xvector_state(0, 10)
    for tag, seed, n in (("harm0_16000", 0, 16000), ("harm3_48000", 3, 48000), ("harm7_24123", 7, 24123)):
        got = {}
        wav = synthetic.harm_batch([seed], n)
        xv = ox.xvector(sd, wav, hook=lambda k, v: got.__setitem__(k, v))
        assert np.abs(got["feats"].numpy() - fx[tag + "/feats"]).max() < 2e-4
        assert np.abs(got["seq"][:, ::16, ::4].numpy() - fx[tag + "/seq_sub16"]).max() < 1e-4
        assert np.abs(got["pooled"].numpy() - fx[tag + "/pooled"]).max() < 1e-4
        assert np.abs(xv.numpy() - fx[tag + "/xvector"]).max() < 1e-5
        assert abs(float(xv.norm()) - 1.0) < 1e-6
    # batch of equal-length utterances = the one-utterance calls (the reference extracts with batch size 1)
    wav = synthetic.harm_batch([1, 2], 16000)
    both = ox.xvector(sd, wav)
    assert torch.allclose(both[0], ox.xvector(sd, wav[0])[0], atol=1e-6)
