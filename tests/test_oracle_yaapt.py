"""The YAAPT restatement (oracle/yaapt.py) against F0 tracks produced by the reference itself
(scripted `satools.hifigan.yaapt.yaapt`, tests/golden/make_fixtures.py).  CPU only.

The reference runs with torch.set_num_threads(1) (satools/satools/hifigan/yaapt.py:27) and so do
these tests: at frame 0 the NCCF of the squared-signal track is computed on a frame that is mostly
zero padding, is flat at ~1.0 to the last ulp, and its "first local maximum" is decided by rounding
noise — torch itself returns a different frame-0 value with 8 threads than with 1 for 3 of these 16
inputs.  Everything else is bit-exact at any thread count."""
import numpy as np
import pytest
import torch

from oracle import biquad
from oracle import yaapt as oy
from satools_amd import synthetic

OPTS = {"frame_length": 35.0, "frame_space": 20.0, "nccf_thresh1": 0.25, "tda_frame_length": 25.0}


@pytest.fixture(autouse=True)
def one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def _wav(name):
    kind, n = name.split("_")[0], int(name.split("_")[1])
    if kind.startswith("harm"):
        return synthetic.harm_batch([int(c) for c in kind[4:]], n)
    return synthetic.rand_batch(int(kind[4:]), 1, n)


def test_yaapt_bit_exact_vs_reference(gold):
    fx = gold.npz("fx_f0.npz")
    for name in fx.files:
        got = oy.yaapt(_wav(name), OPTS).numpy()
        assert got.shape == fx[name].shape, name
        assert np.array_equal(got, fx[name]), name


def test_frame_counts_follow_reference_shape_table(gold):
    shapes = gold.json("fx_shapes.json")
    for n, (bn_shape, f0_shape, y_shape) in shapes.items():
        plan = oy.Plan(int(n), OPTS)
        assert plan.nframes == f0_shape[1]
        assert y_shape[-1] == 320 * bn_shape[2] + 1


def test_plan_constants():
    p = oy.Plan(80000, OPTS)
    assert (p.pad, p.L, p.nframes, p.frame_size, p.frame_jump) == (280, 80560, 250, 560, 320)
    assert (p.nl_lo, p.nl_hi) == (60, 205)
    assert (p.nframe_size, p.wl, p.half_wl, p.min_shc, p.max_shc) == (1120, 21, 10, 31, 256)
    assert (p.pk_width, p.pk_center, p.pk_min_lag, p.pk_max_lag) == (25, 13, 17, 217)
    assert (p.tda_len, p.tda_nframes, p.nccf_center) == (400, 250, 2)


def test_biquad_is_a_clamped_cascade():
    x = np.linspace(-1, 1, 4000, dtype=np.float32) * np.sin(np.arange(4000) * 0.3).astype(np.float32)
    y = biquad.band_limit(x)
    assert y.dtype == np.float32 and y.shape == x.shape and np.abs(y).max() <= 1.0
    # zero input -> zero output (leading padding of the YAAPT signal stays exactly 0)
    assert not biquad.band_limit(np.zeros(100, np.float32)).any()


def test_biquad_fir_is_torch_conv1d_bit_for_bit():
    """row a18: the shipped FIR order (b / a0 first, FMA chain in tap order) IS what torch's CPU conv1d evaluates for
    torchaudio's DifferentiableFIR call; the oracle's exact FMA emulation reproduces it on every sample"""
    import torch.nn.functional as F
    for kind, fc in (("lp", 50.0), ("hp", 1500.0)):
        k = biquad.kernel_constants(kind, 16000, fc)
        w = torch.tensor([k[2], k[1], k[0]]).view(1, 1, 3)
        for x in (synthetic.harm_utterance(3, 40000), synthetic.rand_batch(1, 1, 40000)[0] * 2 - 1):
            ref = F.conv1d(F.pad(x.view(1, 1, -1), (2, 0)), w).view(-1).numpy()
            assert np.array_equal(biquad.fir(x.numpy(), kind, 16000, fc), ref)
            assert not np.array_equal(biquad.fir(x.numpy(), kind, 16000, fc, "raw_b_then_divide"), ref)


def test_biquad_rounding_order_is_a_tested_decision(gold):
    """the committed study (tests/golden/make_biquad_order_study.py): the two FIR orders give different last bits on
    most band-limited samples; on the 15 fixture inputs no F0 frame notices, on 100 extra utterances 255 of 25 000
    frames do (7 utterances) — so the order is decision-relevant and the torchaudio one is shipped.  A subset is
    recomputed here."""
    st = gold.json("fx_biquad_order.json")
    assert st["shipped"] == "torchaudio" and st["orders"] == list(biquad.ORDERS)
    assert st["fx_f0_inputs"]["f0_frames_differ"] == 0 and st["fx_f0_inputs"]["frames"] == 1635
    assert (st["extra_100"]["f0_frames_differ"], st["extra_100"]["frames"]) == (255, 25000)
    assert st["extra_100"]["filtered_samples_differ"] > 10_000_000
    rows = {r["name"]: r for r in st["per_input"]}
    for name in ("harm135_80000", "rand139_80000", "harm100_80000"):
        seed = int(name[4:name.index("_")])
        w = synthetic.harm_batch([seed], 80000) if name.startswith("harm") else synthetic.rand_batch(seed, 1, 80000)
        a = oy.yaapt(w, OPTS).numpy()
        b = oy.yaapt(w, OPTS, biquad_order="raw_b_then_divide").numpy()
        assert int((a != b).sum()) == rows[name]["f0_frames_differ"], name
    assert rows["harm135_80000"]["f0_frames_differ"] == 3 and rows["harm100_80000"]["f0_frames_differ"] == 0


def test_all_unvoiced_input_raises_like_the_reference():
    # the reference's spec_track applies medfilt to an empty tensor when no frame is voiced and
    # fails inside unfold (yaapt.py:54-69 via :257); the restatement fails the same way
    with pytest.raises(RuntimeError):
        oy.yaapt(torch.zeros(1, 8000), OPTS)


def test_biquad_recursion_is_cross_checked_against_scipy(gold):
    """SURVEY §8 row a18, the IIR half (tests/golden/make_biquad_iir_crosscheck.py -> fx_biquad_iir.json): an independent implementation
    (scipy.signal.lfilter, direct form II transposed) PINS the structure and the coefficients of oracle/biquad.py — rounding is <= 1e-4
    of the filtered signal's peak, every wrong recurrence >= 1e-2 — and CANNOT pin the rounding order of the f32 recursion: the shipped
    order and three alternatives sit in the same envelope around the float64 truth, as does scipy's own float32 result.  What the
    order costs is committed too (about 1 % of the F0 frames of the study move, on 2-3 of its 55 utterances).  Recomputed here: the
    structure check on two utterances and the order study on one utterance of each kind."""
    from scipy.signal import lfilter
    fx = gold.json("fx_biquad_iir.json")
    assert fx["shipped"] == "torchaudio" and fx["iir_orders"] == list(biquad.IIR_ORDERS)
    for key, v in fx["structure"]["per_filter"].items():
        assert v["ours_vs_f64_rel"] < 1.5e-4 and v["scipy32_vs_f64_rel"] < 1.5e-4 and v["ours_vs_scipy32_rel"] < 3e-4, (key, v)
        assert all(w > 1e-2 for w in v["wrong_transcriptions_rel_min"].values()), (key, v)
    ro = fx["rounding_order"]["vs_shipped"]
    assert ro["torchaudio"]["f0_frames_differ"] == 0
    for o in ("c1_first", "fma", "sum_first"):
        assert 0 < ro[o]["f0_frames_differ"] <= 0.02 * fx["rounding_order"]["frames"] and ro[o]["max_abs_diff"] < 1e-6, (o, ro[o])
    # structure, recomputed
    for name in ("harm200", "rand201"):
        x = (synthetic.harm_batch([200], 80000) if name.startswith("harm") else synthetic.rand_batch(201, 1, 80000))[0].numpy()
        for kind, cutoff in (("lp", 50.0), ("hp", 1500.0)):
            b, a = biquad.coeffs(kind, 16000, cutoff)
            t64 = lfilter((b / a[0]).astype(np.float64), (a / a[0]).astype(np.float64), x.astype(np.float64))
            s32 = lfilter((b / a[0]).astype(np.float32), (a / a[0]).astype(np.float32), x).astype(np.float64)
            scale = np.abs(t64).max()
            for o in ("torchaudio", "c1_first", "sum_first"):
                ours = biquad.biquad(x, kind, 16000, cutoff, iir=o, clamp=False).astype(np.float64)
                assert np.abs(ours - t64).max() < 1.5e-4 * scale and np.abs(ours - s32).max() < 3e-4 * scale, (name, kind, o)
            swapped = lfilter((b / a[0]).astype(np.float64), np.array([1.0, a[2] / a[0], a[1] / a[0]], dtype=np.float64), x.astype(np.float64))
            assert not np.abs(np.nan_to_num(swapped, nan=1e30) - t64).max() < 1e-2 * scale
    # order study, recomputed on one utterance whose frames move and one whose frames do not
    torch.set_num_threads(1)
    per = {r["name"]: r for r in fx["per_input"]}
    for name, w in (("harm106_80000", synthetic.harm_batch([106], 80000)), ("harm100_80000", synthetic.harm_batch([100], 80000))):
        ref = oy.yaapt_one(w[0], OPTS).numpy()
        alt = oy.yaapt_one(w[0], OPTS, biquad_iir="c1_first").numpy()
        assert int((alt != ref).sum()) == per[name]["c1_first"], name
