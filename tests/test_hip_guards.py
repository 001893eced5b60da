"""GPU tests of the round-6 guards: the precision guard wired into `load_model` (reference: satools/satools/infer_helper.py:10-59),
`CoreHifiGan.last_arithmetic`, the range probe, the near-tie guard of the VQ decision inside `convert()` (chain/nn.py:424-459), and the
buffer checks of the public SAT_CONV_F16F8R entry."""
import logging
import math
import os

import numpy as np
import pytest
import torch

from conftest import rms

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FBANK_TAG = "hifigan_bn_tdnnf_600h_vq_48_v1"
W2V2_TAG = "hifigan_bn_tdnnf_wav2vec2_vq_48_v1"


def _oracle_f0(wav):
    from oracle import yaapt as oyaapt
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        return oyaapt.yaapt(wav, {"frame_length": 35.0, "frame_space": 20.0, "nccf_thresh1": 0.25, "tda_frame_length": 25.0})
    finally:
        torch.set_num_threads(nt)


def _trained_like_state(sigma_rows=1.0, seed=11, big_pair=None):
    """the synthetic fbank-tag checkpoint with statistics a trained weight-normed HiFi-GAN can have and seeded-random weights do not:
    `weight_g` of conv1 of every ResBlock step of the two thick stages (C = 256 / 128: the SAT_CONV_F16F8R layers) drawn log-normal
    over its output channels (sigma_rows nats), conv1's bias scaled with its row, and the matching INPUT channel of conv2 divided by
    the same gain — leaky_relu is positively homogeneous, so the generator's function is unchanged in exact arithmetic while the rows
    of a layer, and with them the inner activations' channels, differ by an order of magnitude and more.
    big_pair = (resblock, step, log2 gain): that ONE layer pair scaled as a whole on top (conv1 up, conv2 down)"""
    from satools_amd import synthetic
    state, _ = synthetic.checkpoint(FBANK_TAG)
    sd = {k: v.clone() for k, v in state["base_model_state_dict"].items()}
    g = torch.Generator().manual_seed(seed)
    n_k = 3                                                    # resblocks per stage (kernel sizes 3 / 7 / 11)
    for rb in range(2 * n_k):                                  # stages 1 and 2
        for step in range(3):
            k1, k2 = f"hifigan.resblocks.{rb}.convs1.{step}", f"hifigan.resblocks.{rb}.convs2.{step}"
            c = sd[k1 + ".weight_g"].shape[0]
            gain = torch.exp2(torch.round(torch.randn(c, generator=g) * sigma_rows / math.log(2.0)))     # powers of two: exact rescaling
            if big_pair is not None and (rb, step) == tuple(big_pair[:2]):
                gain = gain * 2.0 ** big_pair[2]
            sd[k1 + ".weight_g"] = sd[k1 + ".weight_g"] * gain.view(-1, 1, 1)
            sd[k1 + ".bias"] = sd[k1 + ".bias"] * gain
            # conv2's weight = g2 * v2 / |v2| per OUTPUT channel: dividing input channel c of the folded weight by gain[c] needs the
            # folded form — replace (g2, v2) by the plain folded weight rescaled, with g2 = its row norms (weight_norm's identity)
            v2, g2 = sd[k2 + ".weight_v"], sd[k2 + ".weight_g"]
            w2 = torch._weight_norm(v2, g2, 0) / gain.view(1, -1, 1)
            sd[k2 + ".weight_v"] = w2
            sd[k2 + ".weight_g"] = w2.flatten(1).norm(dim=1).view(-1, 1, 1)
    state = dict(state)
    state["base_model_state_dict"] = sd
    return state, sd


def _load_through_the_guard(tmp_path, name, state, caplog):
    import satools_amd
    path = tmp_path / name / "final.pt"
    path.parent.mkdir()
    torch.save(state, str(path))
    with caplog.at_level(logging.INFO, logger="satools_amd"):
        model = satools_amd.load_model(str(path))
        assert model.__dict__.get("_precision_check_pending") is True
        model.to(DEV)
    model.eval()
    rep = model.__dict__.get("precision_report")
    assert rep is not None and model.__dict__["_precision_check_pending"] is False
    assert any("precision guard" in r.message for r in caplog.records)
    return model, rep


def test_load_model_runs_the_precision_guard_on_a_trained_like_checkpoint(tmp_path, caplog):
    """`load_model(<checkpoint file>)` + `.to("cuda")` = the precision guard, once, with its report logged and kept.
    (a) log-normal row gains (one nat: the rows of a layer spread over ~2^6): every arithmetic passes — the 8-bit cross-term operands
        carry the layer scale's range (e4m3 weights lose nothing down to 2^-14 of the layer's largest, as the f16 lo halves; counted:
        `generator_f8_weights`), the planes stay far inside the f16 / e5m2 range (`generator_range`) — nothing falls back, and the
        generator forced onto the 8-bit kernels meets the 1e-5 bar against the CPU oracle;
    (b) the same checkpoint with ONE layer pair carrying a 2^12 gain (conv1 up, conv2 down: the same function): its inner activation
        reaches 2e4, still inside the f16 / e5m2 range (the sidecar is e5m2 BECAUSE of such layers) — nothing falls back and
        the 8-bit kernels still meet the bar;
    (c) a 2^15 gain: the inner activation passes 65 504 (counted by the range probe at that stage).  The 8-bit sidecar saturates and
        "f16f8r" fails the comparison with the exact-f32 kernels: the guard takes the generator to "f16x3", whose planes still carry
        such values (hi, rounded toward zero, stops at 65 504 and lo takes the rest up to twice that) — convert() meets the path's bar;
    (d) a 2^18 gain: both split arithmetics fail, the generator falls back to the exact-f32 kernels and convert() is at 1e-6 of the CPU
        oracle again instead of saturating silently."""
    from oracle import convert as oconv
    from oracle import hifigan as ohg
    from satools_amd import synthetic
    wav = synthetic.harm_batch([2], 16000)
    f0 = _oracle_f0(wav)
    # (a)
    state, sd = _trained_like_state(sigma_rows=1.0)
    model, rep = _load_through_the_guard(tmp_path, "trained_like", state, caplog)
    print("precision guard, log-normal row gains:", rep)
    st = rep["generator_f8_weights"]
    assert st["values"] > 0 and st["clipped"] == 0 and st["flushed"] < 0.01 * st["values"], st      # (measured 0.33 %: lo halves of the smallest rows)
    assert rep["fallback"] == [] and model.hifigan.precision == "f16f8r", rep
    assert sum(rep["generator_range"]["past_e5m2_max"]) == 0 and rep["generator_arithmetic_f16f8r_ran"].startswith("f16f8r")
    ref = oconv.convert_fbank(sd, model.spk, wav, model.spk[1], f0)
    model.hifigan.set_force_f8(1)                      # one utterance is too small a batch for the 8-bit kernels' default dispatch
    try:
        y = model.convert(wav.to(DEV), target=model.spk[1])
        assert model.hifigan.last_arithmetic.startswith("f16f8r")
    finally:
        model.hifigan.set_force_f8(0)
    err = rms(y.cpu().numpy() - ref.numpy())
    print("convert() on the 8-bit kernels, log-normal row gains: RMS error against the CPU oracle", err)
    assert err < 1e-5, err
    model.to(DEV)                                      # the guard runs once
    assert model.__dict__["precision_report"] is rep
    # (b)
    caplog.clear()
    state_b, sd_b = _trained_like_state(sigma_rows=1.0, big_pair=(1, 1, 12))
    model_b, rep_b = _load_through_the_guard(tmp_path, "one_pair_2p12", state_b, caplog)
    print("precision guard, one layer pair with a 2^12 gain:", rep_b)
    assert rep_b["fallback"] == [] and model_b.hifigan.precision == "f16f8r", rep_b
    assert 4096.0 < rep_b["generator_range"]["max_abs"][0] < 57344.0 and sum(rep_b["generator_range"]["past_e5m2_max"]) == 0
    ref_b = oconv.convert_fbank(sd_b, model_b.spk, wav, model_b.spk[1], f0)
    model_b.hifigan.set_force_f8(1)
    try:
        err_b = rms(model_b.convert(wav.to(DEV), target=model_b.spk[1]).cpu().numpy() - ref_b.numpy())
    finally:
        model_b.hifigan.set_force_f8(0)
    print("convert() on the 8-bit kernels, one layer pair with a 2^12 gain: RMS error against the CPU oracle", err_b)
    assert err_b < 1e-5, err_b
    # (c)
    caplog.clear()
    state_c, sd_c = _trained_like_state(sigma_rows=1.0, big_pair=(1, 1, 15))
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        model_c, rep_c = _load_through_the_guard(tmp_path, "one_pair_2p15", state_c, caplog)
    print("precision guard, one layer pair with a 2^15 gain:", rep_c)
    assert rep_c["fallback"] == ["generator: f16f8r -> f16x3"] and model_c.hifigan.precision == "f16x3" and w, rep_c
    assert rep_c["generator_range"]["past_e5m2_max"][0] > 0, rep_c["generator_range"]        # (what the saturated stage hands on may count further down too)
    ref_c = oconv.convert_fbank(sd_c, model_c.spk, wav, model_c.spk[1], f0)
    err_c = rms(model_c.convert(wav.to(DEV), target=model_c.spk[1]).cpu().numpy() - ref_c.numpy())
    print("convert() after the guard's fall-back to f16x3: RMS error against the CPU oracle", err_c)
    assert err_c < 1e-4, err_c                       # the bar of the path; hi (round toward zero) stops at 65 504 and lo carries the rest up to 2 x that
    # (d)
    caplog.clear()
    state_d, sd_d = _trained_like_state(sigma_rows=1.0, big_pair=(1, 1, 18))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        model_d, rep_d = _load_through_the_guard(tmp_path, "one_pair_2p18", state_d, caplog)
    print("precision guard, one layer pair with a 2^18 gain:", rep_d)
    assert "generator" in rep_d["fallback"] and model_d.hifigan.precision == "f32" and w, rep_d
    ref_d = oconv.convert_fbank(sd_d, model_d.spk, wav, model_d.spk[1], f0)
    err_d = rms(model_d.convert(wav.to(DEV), target=model_d.spk[1]).cpu().numpy() - ref_d.numpy())
    print("convert() after the guard's fall-back to the exact-f32 kernels: RMS error against the CPU oracle", err_d)
    assert err_d < 1e-5, err_d


def test_load_model_guard_can_be_skipped_and_leaves_synthetic_loads_alone(tmp_path, monkeypatch):
    import satools_amd
    from satools_amd import synthetic
    state, _ = synthetic.checkpoint(FBANK_TAG)
    path = tmp_path / "plain" / "final.pt"
    path.parent.mkdir()
    torch.save(state, str(path))
    monkeypatch.setenv("SATOOLS_AMD_CHECK_PRECISION", "0")
    model = satools_amd.load_model(str(path))
    model.to(DEV)
    assert "precision_report" not in model.__dict__ and model.hifigan.precision == type(model.hifigan).precision
    monkeypatch.delenv("SATOOLS_AMD_CHECK_PRECISION")
    model = satools_amd.load_model(str(path))          # the same (well-conditioned) weights through the guard: nothing falls back
    model.to(DEV)
    rep = model.__dict__["precision_report"]
    assert rep["fallback"] == [] and model.hifigan.precision == type(model.hifigan).precision, rep
    syn = satools_amd.load_model("synthetic:" + FBANK_TAG)
    syn.to(DEV)
    assert "precision_report" not in syn.__dict__


def test_last_arithmetic_names_what_a_batch_ran():
    """the "f16f8r" generator serves a batch too small for the ring kernel on the f16x3 tiles: `last_arithmetic` says which ran"""
    import satools_amd
    model = satools_amd.load_model("synthetic:" + FBANK_TAG)
    model.to(DEV)
    g = model.hifigan
    if g.precision != "f16f8r":
        pytest.skip("default generator precision is not f16f8r")
    assert g.last_arithmetic is None
    x1 = torch.randn(1, g.imput_dim, 50, device=DEV)
    g(x1)
    assert g.last_arithmetic == "f16x3"
    x32 = torch.randn(32, g.imput_dim, 250, device=DEV)
    g(x32)
    assert g.last_arithmetic == "f16f8r(stages 1,2)", g.last_arithmetic
    g.set_force_f8(1)
    try:
        y8 = g(x1)[0]
        assert g.last_arithmetic == "f16f8r(stages 1,2)"
    finally:
        g.set_force_f8(0)
    y3 = g(x1)[0]
    assert g.last_arithmetic == "f16x3"
    d = rms((y8 - y3).cpu().numpy())
    assert 0 < d < 1e-5, d                                  # two arithmetics: different bits, both far inside the bar
    g.precision = "f32"
    g(x1)
    assert g.last_arithmetic == "f32"


def test_range_probe_counts_values_past_the_8bit_range():
    import satools_amd
    model = satools_amd.load_model("synthetic:" + FBANK_TAG)
    model.to(DEV)
    g = model.hifigan
    x = torch.randn(2, g.imput_dim, 40, device=DEV)
    rep = g.range_probe(x)
    assert len(rep["max_abs"]) == 5 and sum(rep["past_e5m2_max"]) == 0 and all(0 < v < 1e4 for v in rep["max_abs"]), rep
    rep2 = g.range_probe(x * 3e4)                           # the input projection's output grows with it: past 57 344 somewhere
    assert sum(rep2["past_e5m2_max"]) > 0 and max(rep2["max_abs"]) > 57344, rep2
    y = g(x)[0]                                             # the probe is off again
    assert torch.isfinite(y).all()


@pytest.mark.parametrize("tag", [FBANK_TAG, W2V2_TAG])
def test_convert_patches_near_tie_utterances(tag):
    """convert() with the near-tie guard forced to flag EVERY utterance (a window of 1e6 sigma): the flagged rows are decided again on
    the exact kernels and generated again, so the result is convert() of a model whose extractor runs exact f32 — to the generator's
    batch-size-dependent arithmetic (~1e-6) — and the deferred form (`defer_status=True`) gives the bits of the plain call"""
    import satools_amd
    from satools_amd import synthetic
    model = satools_amd.load_model("synthetic:" + tag)
    model.to(DEV)
    model.eval()
    ext = model.bn_extractor
    wav = synthetic.harm_batch([31, 32, 33], 32000).to(DEV)
    tg = [model.spk[1], model.spk[0], model.spk[2]]
    f0 = model.get_f0(wav)
    keep = ext.vq_tie_sigmas
    ext.vq_tie_sigmas = 0.0
    y_plain = model.convert(wav, target=tg).clone()
    ext.vq_tie_sigmas = 1e6
    ext.vq_tie_force_patch = True           # (an utterance is generated again only when its indices change: here, always)
    ext.__dict__.pop("tie_stats", None)
    y_all = model.convert(wav, target=tg).clone()
    assert ext.tie_stats["rerun"] == 3 and ext.tie_stats["changed"] == 3 and ext.tie_stats["utterances"] == 3, ext.tie_stats
    y_def, st = model.convert(wav, target=tg, defer_status=True)
    st.check()
    torch.cuda.synchronize()
    assert st.rows == [0, 1, 2] and torch.equal(y_def, y_all)
    model.set_f0(f0.clone().unsqueeze(0))                       # F0 handed in (set_f0): the same deferred path
    y_set = model.convert(wav, target=tg).clone()
    assert torch.equal(y_set, y_all)
    ext.vq_tie_force_patch = False
    ext.__dict__.pop("tie_stats", None)
    y_cmp = model.convert(wav, target=tg).clone()       # every utterance decided again, none CHANGED (no flip among them): nothing rewritten
    assert ext.tie_stats["rerun"] == 3 and (ext.tie_stats["changed"] > 0 or torch.equal(y_cmp, y_plain)), ext.tie_stats
    ext.vq_tie_sigmas = 0.0
    with ext._exact(ext):
        y_exact = model.convert(wav, target=tg).clone()
    ext.vq_tie_sigmas = keep
    e_all, e_plain = rms((y_all - y_exact).cpu().numpy()), rms((y_plain - y_exact).cpu().numpy())
    print(f"{tag}: convert() with every utterance decided again vs the exact-f32 extractor {e_all:.2e}; without the guard {e_plain:.2e}")
    assert e_all < 3e-6, e_all


def test_windowed_exact_decision_gives_the_frames_of_the_full_run():
    """fbank tag: the second decision of a near-tie utterance computes only the window of frames around its near-ties (TDNNF layers are
    'valid' windows over frames, and the exact-f32 kernels give a frame the same bits whatever tile it falls in): the window's
    quantised frames and indices ARE those of the whole utterance on the exact kernels — windows at the start, in the middle, at the
    very end (moved left) and a span too wide to window"""
    import satools_amd
    from satools_amd import synthetic
    model = satools_amd.load_model("synthetic:" + FBANK_TAG)
    model.to(DEV)
    model.eval()
    ext = model.bn_extractor
    wav = synthetic.harm_batch([41, 42, 43, 44], 80000).to(DEV)
    with torch.no_grad():
        feats = ext._features_of(wav)
        rows = [0, 1, 2, 3]
        zq_f, idx_f, t0_f = ext._exact_rows(rows, feats, wav)
        Tq = idx_f.shape[1]
        assert t0_f == [0, 0, 0, 0] and zq_f.shape[2] == Tq
        S, W = ext._stack_receptive_field()
        assert (feats.shape[2] - W) // S + 1 == Tq
        for spans in ([(0, 0), (5, 9), (Tq - 1, Tq - 1), (Tq - 40, Tq - 3)], [(120, 121), (7, 7), (200, 230), (Tq - 2, Tq - 1)]):
            zq_w, idx_w, t0 = ext._exact_rows(rows, feats, wav, spans=spans)
            Lq = idx_w.shape[1]
            assert Lq < Tq // 2
            for i, (lo, hi) in enumerate(spans):
                assert t0[i] <= lo and hi < t0[i] + Lq <= Tq, (spans[i], t0[i], Lq)
                assert torch.equal(idx_w[i], idx_f[i, t0[i]:t0[i] + Lq]) and torch.equal(zq_w[i], zq_f[i, :, t0[i]:t0[i] + Lq]), (spans[i], t0[i])
        zq_a, idx_a, t0_a = ext._exact_rows(rows, feats, wav, spans=[(3, Tq - 5)] * 4)       # too wide: the full run
        assert t0_a == [0, 0, 0, 0] and torch.equal(idx_a, idx_f) and torch.equal(zq_a, zq_f)


def test_f8r_entry_refuses_foreign_packings_and_short_sidecars():
    """ops.conv1d(mode=CONV_F16F8R) moves its operands by LDS-DMA at offsets derived from the shapes: an f16x3 packing (half the bytes)
    or a short / stale sidecar must be refused, not read past its end (round-5 advisor item; sat_conv1d_desc.x_split8 / y_split8)"""
    from satools_amd import _lib, ops, packing
    B, C, T, k = 2, 128, 700, 7
    x = torch.randn(B, C, T, device=DEV)
    w = torch.randn(C, C, k, device=DEV) * 0.02
    b = torch.zeros(C, device=DEV)
    xs = ops.act_split(x, 0.1)
    xs8 = ops.planes_f8_sidecar(xs)
    w8, w3 = packing.pack_conv_weight_f16f8r(w), packing.pack_conv_weight_f16x3(w)
    kw = dict(bias=b, pad_left=3, mode=_lib.CONV_F16F8R, x_split=xs, y_split=ops.split_like(B, C, T, DEV), y_split_slope=0.1, no_y=True)
    ops.conv1d(x, w8, C, k, x_split8=xs8, **kw)                                     # the well-formed call runs
    with pytest.raises(_lib.SatError, match="pack_conv_weight_f16f8r"):
        ops.conv1d(x, w3, C, k, x_split8=xs8, **kw)
    with pytest.raises(_lib.SatError, match="x_split8"):
        ops.conv1d(x, w8, C, k, x_split8=xs8[:1], **kw)
    with pytest.raises(_lib.SatError, match="x_split8"):
        ops.conv1d(x, w8, C, k, **kw)
    with pytest.raises(_lib.SatError, match="y_split8"):
        ops.conv1d(x, w8, C, k, x_split8=xs8, y_split8=ops.sidecar_like(1, C, T, DEV), **kw)
    # a STALE sidecar (of other planes) is not detectable by size: it is a pure function of the planes, so a caller can check it
    other = ops.planes_f8_sidecar(ops.act_split(x * 2, 0.1))
    assert not torch.equal(other, xs8) and torch.equal(ops.planes_f8_sidecar(xs), xs8)
    torch.cuda.synchronize()


def test_split_f16_kernels_refuse_leaky_relu_slopes_outside_0_1():
    """the split-f16 epilogues apply a leaky-relu as max(v, slope v) and undo one as min(r, r / slope) (csrc/common.h lrelu_max,
    lrelu_undo_min: the bits of the select for 0 <= slope <= 1 only): other slopes are refused where the reference's own activations
    (0.1, 0.01, none) never go; the exact-f32 kernels keep the select and take any slope"""
    import torch.nn.functional as F
    from satools_amd import _lib, ops, packing
    B, C, T, k = 2, 32, 300, 3
    x = torch.randn(B, C, T, device=DEV)
    w = torch.randn(C, C, k, device=DEV) * 0.1
    b = torch.randn(C, device=DEV) * 0.1
    w3, w0 = packing.pack_conv_weight_f16x3(w), packing.pack_conv_weight(w)
    for slope in (0.0, 0.1, 1.0):
        y = ops.conv1d(x, w3, C, k, bias=b, pad_left=1, pad_right=1, in_lrelu=slope, mode=1)
        ref = F.conv1d(F.leaky_relu(x, slope), w, b, padding=1)
        assert (y - ref).abs().max() < 1e-4
    for slope in (1.5, -0.1):
        with pytest.raises(_lib.SatError, match=r"\[0, 1\]"):
            ops.conv1d(x, w3, C, k, bias=b, pad_left=1, pad_right=1, in_lrelu=slope, mode=1)
        with pytest.raises(_lib.SatError, match=r"\[0, 1\]"):
            ops.act_split(x, slope)
        y = ops.conv1d(x, w0, C, k, bias=b, pad_left=1, pad_right=1, in_lrelu=slope, mode=0)      # exact f32: any slope
        assert (y - F.conv1d(F.leaky_relu(x, slope), w, b, padding=1)).abs().max() < 1e-4
    with pytest.raises(_lib.SatError, match=r"\[0, 1\]"):
        ops.conv1d(x, w3, C, k, bias=b, pad_left=1, pad_right=1, mode=1, y_split=ops.split_like(B, C, T, DEV), y_split_slope=2.0)
    torch.cuda.synchronize()
