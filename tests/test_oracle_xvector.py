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


def test_melspectrogram_is_cross_checked_against_an_independent_implementation():
    """SURVEY §8 row aX / f3, the front end: torchaudio.transforms.MelSpectrogram (satools/satools/sidekit/preprocessor.py:205-213: n_fft 1024,
    win 400, hop 160, hann, power 2, 80 mel over 90-7600 Hz) is third party and absent; oracle/melspec.py restates it.  Round 5 pins the
    restatement against an independent implementation that IS installed: HF transformers' `audio_utils` (its `mel_filter_bank(norm=None,
    mel_scale="htk")` and `spectrogram(center=True, pad_mode="reflect")` are written to reproduce torchaudio's `melscale_fbanks` and
    `Spectrogram`) — filter bank, power mel spectrogram and the log the reference takes of it; the other mel scale differs by O(1)."""
    from transformers import audio_utils as au
    from oracle import melspec
    fb_o = melspec.melscale_fbanks().numpy()
    fb_t = au.mel_filter_bank(513, 80, 90.0, 7600.0, 16000, norm=None, mel_scale="htk")
    assert fb_o.shape == fb_t.shape == (513, 80) and np.abs(fb_o - fb_t).max() < 2e-5        # (f32 linspace against float64)
    assert np.abs(fb_o - au.mel_filter_bank(513, 80, 90.0, 7600.0, 16000, norm=None, mel_scale="slaney")).max() > 0.5
    win = au.window_function(400, "hann", periodic=True, frame_length=1024, center=True)       # the 400-sample window centred in the 1024-point frame, as torch.stft does
    assert np.abs(win[312:712] - torch.hann_window(400, periodic=True).numpy()).max() < 1e-6 and not win[:312].any() and not win[712:].any()
    for seed, n in ((0, 16000), (1, 48000), (2, 8123)):
        x = synthetic.harm_batch([seed], n)[0] if seed else torch.randn(n, generator=torch.Generator().manual_seed(0)) * 0.1
        ref = au.spectrogram(x.numpy().astype(np.float64), win, frame_length=1024, hop_length=160, fft_length=1024, power=2.0, center=True,
                             pad_mode="reflect", onesided=True, mel_filters=fb_t, mel_floor=0.0, dtype=np.float64)
        got = melspec.melspectrogram(x).numpy()
        assert got.shape == ref.shape == (80, 1 + n // 160)
        assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max()
        assert np.abs(np.log(got + 1e-6) - np.log(ref + 1e-6)).max() < 2e-4                     # (`torch.log(MelSpec(x) + 1e-6)`, preprocessor.py:229-230)
