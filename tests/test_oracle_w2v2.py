"""wav2vec2-tag plumbing of the oracle against the reference's own `tdnnf_wav2vec2_vq.Net` run with the
torchaudio stand-in (fixtures from tests/golden/make_fixtures.py), and the wav2vec2 model itself against an
independent implementation, HF transformers' stable-layer-norm Wav2Vec2Model (fixtures from
tests/golden/make_w2v2_crosscheck.py; the architecture torchaudio's own `import_huggingface_model` maps
one-to-one onto the reference's configuration).  CPU only."""
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
    assert agree.all()
    assert np.abs(bn.permute(0, 2, 1).numpy() - fx["harm01_16000/bn"])[:, :, agree[0] & agree[1]].max() < 2e-4


def test_frame_counts(gold):
    shapes = gold.json("fx_shapes_w2v2.json")
    assert shapes.pop("forward_2x32000") == [2, 66, 3280]      # the reference's own validate_model assertion (66 frames)
    for n, (bn_shape, f0_shape) in shapes.items():
        assert ow.frames_out(int(n)) + 1 == bn_shape[2]     # 249 wav2vec2 frames -> replicate-padded to 250
    assert ow.frames_out(80000) == 249


def test_w2v2_model_matches_hf_transformers(gold, w2v2_state):
    """row a16: the restated wav2vec2-large reproduces HF's Wav2Vec2Model layer by layer: conv feature extractor,
    feature projection, positional conv, RAW outputs of encoder layers 0 / 11 / 23 (no encoder-level LayerNorm in
    `extract_features`), and the LayerNorm-after-the-stack of `forward()`"""
    state, _ = w2v2_state
    pre = {k[len("bn_extractor.preprocessor."):]: v for k, v in state["base_model_state_dict"].items()
           if k.startswith("bn_extractor.preprocessor.")}
    fx = gold.npz("fx_w2v2_hf.npz")
    m = ow.Wav2Vec2Restated(24)
    m.load_state_dict(pre, strict=True)
    m.eval()
    wav = synthetic.harm_batch([0, 1], 16000)
    with torch.no_grad():
        fe = m.feature_extractor(wav)
        proj = m.encoder.feature_projection(fe)
        l0_in = proj + m.encoder.transformer.pos_conv_embed(proj)
        outs = m.extract_features(wav)[0]
        fwd = m.forward(wav)[0]
    assert len(outs) == 24
    cmp = lambda a, b: float(np.abs(a.numpy() - b).max())
    assert cmp(fe[:, :, ::8], fx["fe_sub"]) < 1e-5
    assert cmp(proj[:, :, ::16], fx["proj_sub"]) < 1e-5
    assert cmp(l0_in[:, :, ::16], fx["layer0_in_sub"]) < 1e-5
    for li in (0, 11, 23):
        assert cmp(outs[li][:, :, ::16], fx[f"layer{li}_sub"]) < 2e-5, li
    assert cmp(outs[23][0], fx["layer23"]) < 2e-5
    assert cmp(fwd[:, :, ::16], fx["after_final_ln_sub"]) < 1e-5
    # the refuted placement (encoder-level LayerNorm before the stack) is far from HF
    with torch.no_grad():
        wrong = m.extract_features(wav, _ln_placement="before_stack")[0][-1]
    assert cmp(wrong[:, :, ::16], fx["layer23_sub"]) > 0.5
    ver = gold.json("fx_w2v2_hf.json")["max_abs_diff_vs_hf"]
    assert ver["none_in_extract_features"]["layer23"] < 1e-4 < ver["before_stack"]["layer23"]


def test_asr_forward_matches_reference(gold, w2v2_state):
    """f4 for the wav2vec2-tag net: `Net.forward` (tdnnf_wav2vec2_vq.py:316-345) up to the chain / xent outputs"""
    state, _ = w2v2_state
    asr, _ = oconv.split_state_dict(state["base_model_state_dict"])
    fx = gold.npz("fx_w2v2.npz")
    acts = {}
    chain, xent = otd.forward_w2v2(asr, synthetic.harm_batch([0, 1], 16000), hook=lambda k, v: acts.__setitem__(k, v))
    assert chain.shape == xent.shape and chain.shape[2] == 3280
    r = lambda a: float(np.sqrt(np.mean(np.square(a))))
    for key, sub in (("vq_layer", 16), ("after0", 16)):
        want = fx[f"harm01_16000/{key}_sub"]
        assert r(acts[key][..., ::sub].numpy() - want) <= 2e-5 * max(1.0, r(want)), key
    for got, key in ((chain, "chain_sub"), (xent, "xent_sub")):
        want = fx[f"harm01_16000/{key}"]
        assert got[..., ::8].shape == want.shape and r(got[..., ::8].numpy() - want) <= 2e-5 * max(1.0, r(want)), key
    assert np.abs(torch.logsumexp(xent, dim=2).numpy() - fx["harm01_16000/xent_lse"]).max() < 1e-4
