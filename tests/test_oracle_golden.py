"""The CPU oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_fixtures.py).  CPU only."""
import numpy as np
import torch

from conftest import rms
from oracle import convert as oconv
from oracle import f0 as of0
from oracle import fbank as ofb
from oracle import hifigan as ohg
from oracle import tdnnf as otd
from satools_amd import synthetic


def _wav(name):
    kind, n = name.split("_")[0], int(name.split("_")[1])
    if kind.startswith("harm"):
        return synthetic.harm_batch([int(c) for c in kind[4:]], n)
    return synthetic.rand_batch(int(kind[4:]), 1, n)


def test_fbank_matches_reference(gold):
    fx = gold.npz("fx_fbank.npz")
    for name in fx.files:
        got = ofb.fbank(_wav(name) * 32768, 80).numpy()
        assert got.shape == fx[name].shape
        # log-mel values are O(10); f32 FFT re-association noise only
        assert np.abs(got - fx[name]).max() < 2e-4, name


def test_tdnnf_layers_and_vq_match_reference(gold, fbank_tag_state):
    state, _ = fbank_tag_state
    asr, _ = oconv.split_state_dict(state["base_model_state_dict"])
    fx = gold.npz("fx_tdnnf.npz")
    for name in ("harm0_8000", "rand0_8000"):
        acts, aux = {}, {}
        bn = otd.extract_bn_fbank(asr, _wav(name), aux=aux, hook=lambda n, t: acts.__setitem__(n, t))
        for lay in ["tdnn1"] + [f"tdnnfs.{i}" for i in range(0, 20, 2)]:
            ref = fx[f"{name}/{lay}"]
            assert np.abs(acts[lay][..., ::16].numpy() - ref).max() < 5e-5, (name, lay)
        assert np.abs(aux["z"].numpy() - fx[f"{name}/z"]).max() < 5e-5
        assert np.array_equal(aux["idx"].reshape(-1).numpy(), fx[f"{name}/idx"])
        assert np.abs(aux["dist"].reshape(-1, 48).numpy() - fx[f"{name}/dist"]).max() < 2e-3
        assert np.abs(bn.permute(0, 2, 1).numpy() - fx[f"{name}/bn"]).max() < 5e-5


def test_vq_indices_full_length(gold, fbank_tag_state):
    state, _ = fbank_tag_state
    asr, _ = oconv.split_state_dict(state["base_model_state_dict"])
    fx = gold.npz("fx_tdnnf.npz")
    aux = {}
    bn = otd.extract_bn_fbank(asr, synthetic.harm_batch([0, 1], 80000), aux=aux)
    idx, margin = fx["harm01_80000/idx"], fx["harm01_80000/margin"]
    agree = aux["idx"].numpy() == idx
    # frames whose reference margin exceeds the f32 distance noise must agree; report the rest
    assert agree[margin > 2e-3].all()
    assert agree.mean() > 0.995
    assert np.abs(bn.permute(0, 2, 1)[:, ::8].numpy() - fx["harm01_80000/bn_sub"])[..., agree[0] & agree[1]].max() < 5e-5


def test_f0_norm_quant_awgn_match_reference(gold):
    fx = gold.npz("fx_f0norm.npz")
    for a, b in (("in_1xT", "out_1xT"), ("in_2xT", "out_2xT"), ("in_zero_row", "out_zero_row")):
        x = torch.from_numpy(fx[a].copy())
        y = of0.norm_keep_zeros_(x)
        assert y.data_ptr() == x.data_ptr()  # in place, like the reference
        assert np.allclose(y.numpy(), fx[b], atol=1e-6)
    x = torch.from_numpy(fx["in_2xT"].copy()).unsqueeze(0)
    assert np.allclose(of0.norm_keep_zeros_(x).numpy(), fx["out_1x2xT"], atol=1e-6)
    nf = torch.from_numpy(fx["out_2xT"].copy()).unsqueeze(0).permute(1, 0, 2)
    q = of0.quantize(nf, 16)
    assert np.array_equal(q.numpy(), fx["quant16"])
    torch.manual_seed(1234)
    from satools_amd import f0_transforms
    noise = f0_transforms.draw_awgn(q.shape, f0_transforms.parse_awgn_db("quant_16_awgn_2"))
    assert np.array_equal(of0.awgn(q, noise).numpy(), fx["quant16_awgn2_seed1234"])
    assert f0_transforms.parse_quant_bins("quant_16_awgn_2") == 16


def test_generator_matches_reference(gold, fbank_tag_state):
    state, net = fbank_tag_state
    _, gen = oconv.split_state_dict(state["base_model_state_dict"])
    fx = gold.npz("fx_gen.npz")
    acts = {}
    spk = torch.nn.functional.one_hot(torch.from_numpy(fx["spk_argmax"]), len(net.spk))
    f0 = torch.from_numpy(fx["f0_raw"].copy())
    y = oconv.forward(gen, f0, torch.from_numpy(fx["bn"]), spk, hook=lambda n, t: acts.__setitem__(n, t))
    assert np.allclose(f0.numpy(), fx["f0_after"], atol=1e-6)  # normalised in place
    for k in ("conv_pre", "ups.0", "resblocks.0", "resblocks.1", "resblocks.2"):
        assert np.abs(acts[k].numpy() - fx[k]).max() < 2e-6, k
    for st, step in zip(range(1, 5), (4, 16, 32, 64)):
        assert np.abs(acts[f"ups.{st}"][..., ::step].numpy() - fx[f"ups.{st}_sub{step}"]).max() < 2e-6
        for j in range(3):
            k = f"resblocks.{3 * st + j}"
            assert np.abs(acts[k][..., ::step].numpy() - fx[f"{k}_sub{step}"]).max() < 2e-6, k
    assert rms(y.numpy() - fx["y"]) < 1e-6


def test_convert_matches_reference_with_reference_f0(gold, fbank_tag_state):
    """end to end (fbank tag) with the F0 track taken from the reference run; YAAPT itself is
    checked in test_oracle_yaapt.py"""
    state, net = fbank_tag_state
    sd = state["base_model_state_dict"]
    fx, f0fx = gold.npz("fx_e2e.npz"), gold.npz("fx_f0.npz")
    y = oconv.convert_fbank(sd, net.spk, synthetic.harm_batch([0], 80000), net.spk[3], torch.from_numpy(f0fx["harm0_80000"]))
    assert y.shape == (1, 80001) and fx["harm0_80000_str"].shape == (1, 80001)
    assert rms(y.numpy() - fx["harm0_80000_str"]) < 1e-5
    y = oconv.convert_fbank(sd, net.spk, synthetic.harm_batch([0, 1], 80000), [net.spk[3], net.spk[10]],
                            torch.from_numpy(f0fx["harm01_80000_batch"]))
    assert y.shape == (2, 1, 80001)
    assert rms(y.numpy() - fx["harm01_80000_list"]) < 1e-5
    # batch-coupled F0 normalisation: the same utterance differs alone vs in a batch of 2
    assert np.abs(fx["harm01_80000_list"][0, 0] - fx["harm0_80000_str"][0]).max() > 1e-6


def test_mean_reversion_matches_reference(gold):
    """hifigan/nn.py:64-90 `mean_reverv_f0` (option f0-transformation=mean-reverv_<alpha>:<n>)"""
    from oracle import f0 as of0
    fx = gold.npz("fx_meanrev.npz")
    assert of0.parse_mean_reverv("quant_16_mean-reverv_0.5:32") == (0.5, 32)
    for T in ("T52", "T250"):
        x = torch.from_numpy(fx["in_" + T])
        for spec in ("mean-reverv_0.5:32", "mean-reverv_0.3:7", "mean-reverv_1:4"):
            got = of0.mean_reversion(x.clone(), *of0.parse_mean_reverv(spec))
            assert got.shape == x.shape and np.array_equal(got.numpy(), fx[f"{T}/{spec}"]), (T, spec)
    assert gold.json("fx_meanrev.json")["batch_of_2_raises"] == "RuntimeError"
    import pytest
    with pytest.raises(RuntimeError):
        of0.mean_reversion(torch.zeros(2, 1, 50), 0.5, 32)
