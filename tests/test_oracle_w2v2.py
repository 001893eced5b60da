"""wav2vec2-tag plumbing of the oracle against the reference's own `tdnnf_wav2vec2_vq.Net` run with the
torchaudio stand-in (fixtures from tests/golden/make_fixtures.py).  The wav2vec2 arithmetic itself is
third-party and unpinned; what is pinned here is everything around it.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import convert as oconv
from oracle import tdnnf as otd
from oracle import wav2vec2 as ow
from satools_amd import synthetic


@pytest.fixture(scope="module")
def w2v2_state():
    return synthetic.checkpoint("hifigan_bn_tdnnf_wav2vec2_vq_48_v1")


def test_state_dict_keys_match_the_reference(gold, w2v2_state):
    state, net = w2v2_state
    ref = gold.json("state_dict_keys_w2v2.json")
    assert [[k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()] == ref
    assert sum(v.numel() for k, v in net.state_dict().items() if k.startswith("bn_extractor.")) == 327220873


def test_extract_bn_matches_reference(gold, w2v2_state):
    state, _ = w2v2_state
    asr, _ = oconv.split_state_dict(state["base_model_state_dict"])
    fx = gold.npz("fx_w2v2.npz")
    aux, acts = {}, {}
    bn = otd.extract_bn_w2v2(asr, synthetic.harm_batch([0, 1], 16000), aux=aux, hook=lambda n, t: acts.__setitem__(n, t))
    assert bn.shape == (2, 50, 256)
    assert np.abs(acts["w2v2"][:, :, ::16].numpy() - fx["harm01_16000/w2v2_last_sub"]).max() < 1e-4
    agree = aux["idx"].numpy() == fx["harm01_16000/idx"]
    assert agree[fx["harm01_16000/margin"] > 5e-3].all() and agree.mean() > 0.97
    assert np.abs(bn.permute(0, 2, 1).numpy() - fx["harm01_16000/bn"])[:, :, agree[0] & agree[1]].max() < 2e-4


def test_frame_counts(gold):
    shapes = gold.json("fx_shapes_w2v2.json")
    for n, (bn_shape, f0_shape) in shapes.items():
        assert ow.frames_out(int(n)) + 1 == bn_shape[2]     # 249 wav2vec2 frames -> replicate-padded to 250
    assert ow.frames_out(80000) == 249
