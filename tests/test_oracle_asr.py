"""CPU: the oracle's restatement of the ASR half of the fbank-tag net (Net.forward, tdnnf_vq.py:259-284;
SURVEY §8 f4) against fixtures produced by the reference itself (tests/golden/make_fixtures.py, section "asr")."""
import os

import numpy as np
import pytest
import torch

import satools_amd  # noqa: F401
from satools_amd import synthetic
from conftest import GOLD, rms
from oracle import convert as oconv
from oracle import tdnnf as otd


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLD, "fx_asr.npz"))


@pytest.mark.parametrize("name,ids,n", [("harm0_16000", [0], 16000), ("harm01_32000", [0, 1], 32000)])
def test_forward_matches_reference(fx, fbank_tag_state, name, ids, n):
    sd, _ = oconv.split_state_dict(fbank_tag_state[0]["base_model_state_dict"])
    wav = synthetic.harm_batch(ids, n)
    acts = {}
    chain, xent = otd.forward_fbank(sd, wav, hook=lambda k, v: acts.__setitem__(k, v))
    assert chain.shape[2] == 3280 and chain.shape == xent.shape
    for key, sub in (("vq_layer", 16), ("after0", 16), ("after6", 16)):
        want = torch.from_numpy(fx[f"{name}/{key}_sub"])
        assert acts[key][..., ::sub].shape == want.shape
        assert rms(acts[key][..., ::sub] - want) <= 1e-5 * max(1.0, rms(want)), key
    for got, key in ((chain, "chain_sub"), (xent, "xent_sub")):
        want = torch.from_numpy(fx[f"{name}/{key}"])
        assert rms(got[..., ::8] - want) <= 1e-5 * max(1.0, rms(want)), key
    assert torch.allclose(torch.logsumexp(xent, dim=2), torch.from_numpy(fx[f"{name}/xent_lse"]), atol=1e-4)


def test_sub15_layer_shapes(fbank_tag_state):
    """chain/nn.py:267-304: T frames -> (2(T-1))//3 + 1 windows; bypass zero-padded past int(T/1.5)"""
    sd, _ = oconv.split_state_dict(fbank_tag_state[0]["base_model_state_dict"])
    for t in (4, 5, 6, 7, 58):
        x = torch.randn(1, t, 1024, generator=torch.Generator().manual_seed(t))
        y = otd.tdnnf_layer(sd, "tdnnfs_after.0.", x, 1, 1.5, bypass=True)
        assert y.shape == (1, (2 * (t - 1)) // 3 + 1, 1024)
