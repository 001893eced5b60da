"""YAAPT on the HIP device against the reference's own F0 tracks (golden fixtures) and the oracle.
Needs a real MI355X: run with `-m gpu`."""
import numpy as np
import pytest
import torch

from conftest import rms

pytestmark = pytest.mark.gpu
DEV = "cuda"
OPTS = {"frame_length": 35.0, "frame_space": 20.0, "nccf_thresh1": 0.25, "tda_frame_length": 25.0}


def _wav(name):
    from satools_amd import synthetic
    kind, n = name.split("_")[0], int(name.split("_")[1])
    if kind.startswith("harm"):
        return synthetic.harm_batch([int(c) for c in kind[4:]], n)
    return synthetic.rand_batch(int(kind[4:]), 1, n)


def test_f0_matches_reference_tracks(gold):
    """frame-exact agreement with the reference's tracks.  Voiced values are 16000/(lag+1) or
    k*1.953125 Hz, so a frame either matches to the last bit or jumps; the only tolerated jumps are
    at frame 0, whose squared-signal NCCF is flat at ~1.0 (zero padding) and is decided by rounding
    noise in the reference itself (tests/test_oracle_yaapt.py)."""
    from satools_amd import f0 as f0_hip
    fx = gold.npz("fx_f0.npz")
    total = agree = 0
    bad = []
    for name in fx.files:
        got = f0_hip.yaapt(_wav(name).to(DEV), OPTS).cpu().numpy()
        ref = fx[name]
        assert got.shape == ref.shape, name
        eq = got == ref
        total += eq.size
        agree += int(eq.sum())
        if not eq[:, 1:].all():
            bad.append((name, np.argwhere(~eq).tolist()))
    print(f"F0 frames identical to the reference: {agree}/{total}")
    assert not bad, bad
    assert agree / total > 0.995


def test_f0_batch_equals_single_and_oracle():
    from oracle import yaapt as oy
    from satools_amd import f0 as f0_hip
    from satools_amd import synthetic
    torch.set_num_threads(1)
    wav = synthetic.harm_batch([5, 6, 7, 8], 48000)
    ref = oy.yaapt(wav, OPTS).numpy()
    got = f0_hip.yaapt(wav.to(DEV), OPTS).cpu().numpy()
    single = np.concatenate([f0_hip.yaapt(wav[i:i + 1].to(DEV), OPTS).cpu().numpy() for i in range(4)])
    assert np.array_equal(got, single)              # utterances are independent on the GPU too
    assert (got[:, 1:] == ref[:, 1:]).all()
    assert (got == ref).mean() > 0.995


def test_f0_silent_input_raises_like_the_reference():
    from satools_amd import f0 as f0_hip
    with pytest.raises(RuntimeError):
        f0_hip.yaapt(torch.zeros(1, 8000, device=DEV), OPTS)


def test_convert_with_on_path_f0_matches_golden(gold):
    """full convert(): fbank -> TDNNF-VQ, YAAPT, one-hot, generator, nothing handed over"""
    import satools_amd
    from satools_amd import synthetic
    fx = gold.npz("fx_e2e.npz")
    model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1")
    model.to(DEV)
    model.eval()
    y = model.convert(synthetic.harm_batch([0], 80000).to(DEV), target=model.spk[3])
    e1 = rms(y.cpu().numpy() - fx["harm0_80000_str"])
    y = model.convert(synthetic.harm_batch([0, 1], 80000).to(DEV), target=[model.spk[3], model.spk[10]])
    e2 = rms(y.cpu().numpy() - fx["harm01_80000_list"])
    y = model.convert(synthetic.rand_batch(0, 1, 16000).to(DEV), target=model.spk[7])
    e3 = rms(y.cpu().numpy() - fx["rand0_16000_str"])
    print("convert (F0 on path) RMS error vs reference:", e1, e2, e3)
    assert max(e1, e2, e3) < 1e-4
    # get_f0 returns on the input's device, like the reference
    f0 = model.get_f0(synthetic.harm_batch([0], 8000))
    assert f0.device.type == "cpu" and f0.shape == (1, 25)


def test_convert_quant_awgn_option_matches_golden(gold):
    import satools_amd
    from satools_amd import synthetic
    fx = gold.npz("fx_e2e.npz")
    model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1",
                                   option_args={"f0_transformation": "quant_16_awgn_2"})
    model.to(DEV)
    model.eval()
    torch.manual_seed(1234)
    y = model.convert(synthetic.harm_batch([0, 1], 16000).to(DEV), target=[model.spk[3], model.spk[10]])
    err = rms(y.cpu().numpy() - fx["harm01_16000_quant16_awgn2_seed1234"])
    print("convert quant_16_awgn_2 RMS error vs reference:", err)
    assert err < 1e-4


def test_ragged_batch_equals_per_utterance_tracks():
    """sat_yaapt_ragged_f32: zero-padded utterances of different lengths tracked in one launch sequence give, bit
    for bit, the tracks of the one-utterance calls (what the reference's data loader computes), zero-padded"""
    import satools_amd  # noqa: F401
    from satools_amd import f0 as f0_hip, synthetic
    opts = {"frame_length": 35.0, "frame_space": 20.0, "nccf_thresh1": 0.25, "tda_frame_length": 25.0}
    lens = [80000, 16000, 47999, 32123, 80000, 9600]
    wav = torch.zeros(len(lens), max(lens))
    for i, n in enumerate(lens):
        wav[i, :n] = synthetic.harm_batch([i], n)[0]
    wav = wav.to("cuda")
    got = f0_hip.yaapt_ragged(wav, lens, opts).cpu()
    assert got.shape == (len(lens), 250)
    for i, n in enumerate(lens):
        one = f0_hip.yaapt(wav[i:i + 1, :n].contiguous(), opts).cpu()[0]
        assert torch.equal(got[i, :one.shape[0]], one), f"utterance {i} (n = {n})"
        assert (got[i, one.shape[0]:] == 0).all()
    # uniform lengths through the ragged entry point = the batch entry point
    u = synthetic.harm_batch([0, 1, 2], 32000).to("cuda")
    assert torch.equal(f0_hip.yaapt_ragged(u, [32000] * 3, opts), f0_hip.yaapt(u, opts))


@pytest.mark.parametrize("B,n", [(1, 8000), (3, 16123), (2, 31999), (33, 4800), (32, 16000)])
def test_prefilter_bit_identical_to_the_oracle_biquads(B, n):
    """row a18: the band-limited signals (x and x^2 through low-pass -> clamp -> high-pass -> clamp) of the lane-per-chain
    pipeline kernel equal oracle/biquad.py (torchaudio's order: b / a0 first, FIR as conv1d's FMA chain, recursion as
    multiply-subtract) on every sample; aligned (16-byte) and unaligned row paths, one and two blocks of 32 utterances;
    the zero extension past the padded length is zero"""
    from oracle import biquad
    from satools_amd import f0 as f0_hip
    from satools_amd import synthetic
    seeds = list(range(40, 40 + B))
    wav = synthetic.harm_batch(seeds, n)
    if B > 2:
        wav[1] = (wav[1] * 40).clamp(-1, 1)                          # a clipped utterance: exercises the clamps
    _, aux = f0_hip.yaapt(wav.to(DEV), OPTS, return_aux=True)
    filt = aux["filt"].cpu().numpy()                                 # [B, 2, Lz]
    pad = (filt.shape[2] - n) // 2 if False else 280
    L = n + 2 * pad
    for b in sorted(set([0, 1 % B, B // 2, B - 1])):
        x = np.concatenate([np.zeros(pad, np.float32), wav[b].numpy(), np.zeros(pad, np.float32)])
        for sig, v in ((0, x), (1, x * x)):
            ref = biquad.band_limit(v)
            assert np.array_equal(filt[b, sig, :L], ref), (b, sig, int((filt[b, sig, :L] != ref).sum()))
            assert not filt[b, sig, L:].any()


def test_mean_reversion_option_matches_golden(gold):
    """option f0-transformation=mean-reverv_<alpha>:<n> (hifigan/nn.py:64-90, hifigan.py:79-80): the kernel against the
    reference's own outputs — bit for bit for the short windows (an FMA chain in tap order, as torch's conv1d sums
    them), within one ulp of O(1) values for the 32-tap window (the reference's conv1d takes a oneDNN path whose
    summation order is internal to it); `convert` with the option against the reference run; a batch of 2 raises
    RuntimeError as in the reference (its conv1d reads the squeezed [B, T] tensor as B channels)"""
    import satools_amd
    from satools_amd import synthetic
    fx = gold.npz("fx_meanrev.npz")
    model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1",
                                   option_args={"f0_transformation": "mean-reverv_0.5:32"})
    model.to(DEV)
    model.eval()
    for T in ("T52", "T250"):
        x = torch.from_numpy(fx["in_" + T]).to(DEV)
        for spec in ("mean-reverv_0.5:32", "mean-reverv_0.3:7", "mean-reverv_1:4"):
            m = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1", option_args={"f0_transformation": spec}) \
                if spec != "mean-reverv_0.5:32" else model
            if m is not model:
                m.to(DEV)
            got = m.f0_transformation(x.clone()).cpu().numpy()
            ref = fx[f"{T}/{spec}"]
            assert got.shape == ref.shape
            if spec.endswith(":32"):
                assert np.abs(got - ref).max() <= 2.4e-7, (T, spec, np.abs(got - ref).max())
            else:
                assert np.array_equal(got, ref), (T, spec)
    y = model.convert(synthetic.harm_batch([0], 16000).to(DEV), target=model.spk[3])
    err = rms(y.cpu().numpy() - fx["harm0_16000_meanrev_0.5_32"])
    print("convert mean-reverv_0.5:32 RMS error vs reference:", err)
    assert y.shape == (1, 16001) and err < 1e-4
    assert gold.json("fx_meanrev.json")["batch_of_2_raises"] == "RuntimeError"
    with pytest.raises(RuntimeError):
        model.convert(synthetic.harm_batch([0, 1], 16000).to(DEV), target=[model.spk[3], model.spk[10]])
