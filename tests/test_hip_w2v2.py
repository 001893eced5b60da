"""wav2vec2 tag on the HIP path: support kernels against torch, extract_bn / convert against the
reference-generated fixtures.  Needs a real MI355X: run with `-m gpu`."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rms

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def test_conv0_and_layernorm_phase_split():
    from satools_amd import ops
    x = _rand(2, 4005, seed=1, scale=0.3)
    w, b = _rand(512, 1, 10, seed=2, scale=0.3), _rand(512, seed=3, scale=0.1)
    g, beta = 1 + _rand(512, seed=4, scale=0.1), _rand(512, seed=5, scale=0.1)
    ref = F.conv1d(x.unsqueeze(1), w, b, stride=5)
    y = ops.w2v2_conv0(x.to(DEV), w.reshape(512, 10).contiguous().to(DEV), b.to(DEV))
    assert y.shape == ref.shape and (y.cpu() - ref).abs().max() < 1e-5
    ln = F.gelu(F.layer_norm(ref.transpose(1, 2), (512,), g, beta)).transpose(1, 2)
    z = ops.layernorm_ch(y, g.to(DEV), beta.to(DEV), gelu=True)
    assert (z.cpu() - ln).abs().max() < 2e-5
    zs = ops.layernorm_ch(y, g.to(DEV), beta.to(DEV), gelu=True, split_phases=True).cpu()
    T = ln.shape[2]
    assert zs.shape == (2, 1024, (T + 1) // 2)
    assert (zs[:, :512, :] - ln[:, :, 0::2]).abs().max() < 2e-5
    assert (zs[:, 512:, :T // 2] - ln[:, :, 1::2]).abs().max() < 2e-5
    if T % 2:
        assert not zs[:, 512:, -1].any()


@pytest.mark.parametrize("k", [3, 2])
def test_stride2_conv_as_polyphase(k):
    from satools_amd import ops, packing
    from satools_amd.wav2vec2 import _polyphase_stride2_weight
    for T in (801, 800):
        x, w, b = _rand(2, 64, T, seed=1), _rand(96, 64, k, seed=2, scale=0.1), _rand(96, seed=3)
        ref = F.conv1d(x, w, b, stride=2)
        g, beta = torch.ones(64), torch.zeros(64)
        xs = torch.zeros(2, 128, (T + 1) // 2)
        xs[:, :64] = x[:, :, 0::2]
        xs[:, 64:, :T // 2] = x[:, :, 1::2]
        wc, kp = _polyphase_stride2_weight(w)
        y = ops.conv1d(xs.to(DEV), packing.pack_conv_weight(wc.to(DEV)), 96, kp, bias=b.to(DEV), pad_left=0, pad_right=0,
                       t_out=ref.shape[2])
        assert y.shape == ref.shape and (y.cpu() - ref).abs().max() < 3e-5


def test_attention_as_grouped_convs():
    from satools_amd import ops
    B, H, D, T, P = 2, 4, 64, 249, 256
    q, k, v = (_rand(B, H * D, T, seed=s) for s in (1, 2, 3))
    pad = lambda t: F.pad(t, (0, P - T)).contiguous().to(DEV)
    sh = lambda t: t.view(B, H, D, T)
    s = torch.einsum("bhcq,bhcj->bhqj", sh(q), sh(k)) * D ** -0.5
    ref = torch.einsum("bhqj,bhcj->bhcq", torch.softmax(s, -1), sh(v)).reshape(B, H * D, T)
    st = torch.empty(B * H * T, P, device=DEV)
    ops.attention_scores(pad(q), pad(k), st, B, H, D, T)
    ops.softmax_cols(st, B * H, T, scale=D ** -0.5)
    vt = ops.transpose_heads(pad(v), B, H, D, T)
    o = ops.attention_apply(st, vt, B, H, D, T)
    assert o.shape == ref.shape and (o.cpu() - ref).abs().max() < 2e-5


def test_gelu_and_post_residual_epilogue():
    from satools_amd import ops, packing
    x, w, b = _rand(2, 64, 100, seed=1), _rand(64, 4, 128, seed=2, scale=0.05), _rand(64, seed=3)
    ref = x + F.gelu(F.conv1d(x, w, b, padding=64, groups=16)[..., :-1])
    y = ops.conv1d(x.to(DEV), packing.pack_conv_weight(w.to(DEV), groups=16), 64, 128, bias=b.to(DEV), pad_left=64,
                   pad_right=63, groups=16, gelu=True, post_res=x.to(DEV))
    assert y.shape == ref.shape and (y.cpu() - ref).abs().max() < 3e-5


@pytest.fixture(scope="module")
def model():
    import satools_amd
    m = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_wav2vec2_vq_48_v1")
    m.to(DEV)
    m.eval()
    return m


def test_extract_bn_matches_reference_fixture(model, gold):
    from satools_amd import synthetic
    fx = gold.npz("fx_w2v2.npz")
    wav = synthetic.harm_batch([0, 1], 16000)
    feats = model.bn_extractor
    feats._fe_len = []
    bn, (z, idx, dist) = model.bn_extractor.extract_bn(wav.clone().to(DEV), want_aux=True)
    assert bn.shape == (2, 50, 256)
    margin = torch.from_numpy(fx["harm01_16000/margin"])
    agree = idx.cpu().long() == torch.from_numpy(fx["harm01_16000/idx"]).long()
    print("wav2vec2-tag VQ index agreement with the reference run:", agree.float().mean().item())
    assert agree.all(), f"{(~agree).sum().item()} of {agree.numel()} VQ indices differ (smallest margin {margin.min().item():.2e})"
    ref = torch.from_numpy(fx["harm01_16000/bn"]).permute(0, 2, 1)
    assert (bn.cpu() - ref)[agree].abs().max() < 5e-4
    out = model.get_bn(wav.to(DEV))
    assert out.shape == (2, 256, 50)


def test_w2v2_last_layer_matches_oracle(model, gold):
    from satools_amd import synthetic
    fx = gold.npz("fx_w2v2.npz")
    wav = synthetic.harm_batch([0, 1], 16000).to(DEV)
    ext = model.bn_extractor
    lens, t = [], 16000
    from satools_amd.wav2vec2 import CONV_LAYERS
    for _, k, s in CONV_LAYERS:
        t = (t - k) // s + 1
        lens.append(t)
    ext._fe_len = lens
    y = ext.w2v2_features(wav).cpu()                      # [2, 1024, 49]
    ref = torch.from_numpy(fx["harm01_16000/w2v2_last_sub"])   # [2, 49, 64] (every 16th channel)
    err = (y.permute(0, 2, 1)[:, :, ::16] - ref).abs().max().item()
    print("wav2vec2 last-layer max abs error vs the restated CPU model:", err, "(values up to", ref.abs().max().item(), ")")
    assert err < 2e-3


def test_w2v2_last_layer_matches_hf_transformers(model, gold):
    """row a16: the HIP wav2vec2 path against HF transformers' stable-layer-norm Wav2Vec2Model (the independent
    cross-check of tests/golden/make_w2v2_crosscheck.py): RAW output of encoder layer 23 = what torchaudio's
    `extract_features(x)[0][-1]` returns (no encoder-level LayerNorm), whole tensor of utterance 0 + every 16th
    channel of both"""
    from satools_amd import synthetic
    from satools_amd.wav2vec2 import CONV_LAYERS
    fx = gold.npz("fx_w2v2_hf.npz")
    wav = synthetic.harm_batch([0, 1], 16000).to(DEV)
    ext = model.bn_extractor
    lens, t = [], 16000
    for _, k, s in CONV_LAYERS:
        t = (t - k) // s + 1
        lens.append(t)
    ext._fe_len = lens
    y = ext.w2v2_features(wav).cpu().permute(0, 2, 1)     # [2, 49, 1024]
    ref_sub, ref0 = torch.from_numpy(fx["layer23_sub"]), torch.from_numpy(fx["layer23"])
    err = max((y[:, :, ::16] - ref_sub).abs().max().item(), (y[0] - ref0).abs().max().item())
    e_rms = (y[0] - ref0).pow(2).mean().sqrt().item() / ref0.pow(2).mean().sqrt().item()
    print(f"wav2vec2 layer-23 output vs HF transformers: max abs {err:.2e} (values up to {ref0.abs().max().item():.1f}), relative RMS {e_rms:.2e}")
    assert err < 2e-3 and e_rms < 2e-5
    # and it is NOT the LayerNorm-ed tensor HF returns as last_hidden_state
    assert (y[:, :, ::16] - torch.from_numpy(fx["after_final_ln_sub"])).abs().max() > 1.0


def test_convert_w2v2_tag_matches_fixture(model, gold):
    from satools_amd import synthetic
    fx = gold.npz("fx_w2v2.npz")
    y = model.convert(synthetic.harm_batch([0], 16000).to(DEV), target=model.spk[3])
    assert y.shape == fx["harm0_16000/convert"].shape == (1, 16001)
    err = rms(y.cpu().numpy() - fx["harm0_16000/convert"])
    print("wav2vec2-tag convert RMS error vs reference run:", err)
    assert err < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("C,T,split", [(512, 999, True), (512, 1000, True), (1024, 249, False), (512, 249, False)])
def test_layernorm_planes_equal_split_of_f32_output(C, T, split):
    """sat_layernorm_channels_planes_f32: the f32 output equals the plain entry point's, and the planes are the
    split (hi | lo f16, act_split with slope 1) of exactly those values, odd-length zero slot included"""
    from satools_amd import ops
    g = torch.Generator().manual_seed(C + T)
    x = (torch.randn(2, C, T, generator=g) * 3 + 0.5).cuda()
    gamma, beta = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    y0 = ops.layernorm_ch(x, gamma, beta, gelu=split, split_phases=split)
    y1, ys = ops.layernorm_ch(x, gamma, beta, gelu=split, split_phases=split, planes=True)
    assert torch.equal(y0, y1)
    assert torch.equal(ys, ops.act_split(y0, 1.0))
    _, ys2 = ops.layernorm_ch(x, gamma, beta, gelu=split, split_phases=split, planes=True, want_f32=False)
    assert torch.equal(ys, ys2)
    ref = torch.nn.functional.layer_norm(x.cpu().permute(0, 2, 1), (C,), gamma.cpu(), beta.cpu(), 1e-5).permute(0, 2, 1)
    if split:
        ref = torch.nn.functional.gelu(ref)
        ref = torch.cat([ref[:, :, 0::2], torch.nn.functional.pad(ref[:, :, 1::2], (0, ref[:, :, 0::2].shape[2] - ref[:, :, 1::2].shape[2]))], 1)
    assert (y0.cpu() - ref).abs().max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("T", [249, 256, 100, 31, 257, 312, 700])
def test_fused_attention_matches_float64(T):
    """sat_attention_f16x3 (split-f16 products, scores in registers) against softmax(scale q^T k) v in float64; the
    pad columns of v hold NaN (they are uninitialised memory on the path) and must not leak"""
    from satools_amd import ops
    B, heads, hd = 2, 16, 64
    g = torch.Generator().manual_seed(T)
    q, k, v = (torch.randn(B, heads * hd, T, generator=g) * s for s in (1.5, 1.5, 1.0))
    vp = torch.full((B, heads * hd, ((T + 63) // 64) * 64 if T > 256 else 256), float("nan"))
    vp[:, :, :T] = v
    o, os_ = ops.attention_fused(ops.act_split(q.cuda(), 1.0), ops.act_split(k.cuda(), 1.0), vp.cuda(), B, heads, hd, T,
                                 hd ** -0.5, want_f32=True)
    qd, kd, vd = (t.double().reshape(B, heads, hd, T) for t in (q, k, v))
    p = torch.softmax(torch.einsum("bhcq,bhcj->bhqj", qd, kd) * hd ** -0.5, dim=-1)
    ref = torch.einsum("bhqj,bhdj->bhdq", p, vd).reshape(B, heads * hd, T)
    err = (o.cpu().double() - ref).abs().max().item()
    e_rms = (o.cpu().double() - ref).pow(2).mean().sqrt().item()
    print("fused attention max abs err", err, "rms", e_rms)
    # split-f16 drops the lo*lo term (2^-22 per product): ~5e-7 on a score of std 2, the same on a weight
    assert err < 1e-5 and e_rms < 1e-6
    assert torch.equal(os_, ops.act_split(o, 1.0))


def test_long_utterance_attention(model):
    """more than 256 frames (5.1 s): the fused attention kernel keeps a running softmax over blocks of 256 keys; the
    bottleneck features of the split-f16 path agree with the all-f32 setting (scores GEMM + softmax + apply GEMM)"""
    from satools_amd import synthetic
    bx = model.bn_extractor
    old = bx.w2v2_precision
    try:
        out = {}
        for n in (80000, 100000):                      # 249 and 312 frames
            wav = synthetic.harm_batch([3, 4], n).to(DEV)
            for prec in ("f16x3", "f32"):
                bx.w2v2_precision = prec
                bn, (z, idx, dist) = bx.extract_bn(wav.clone(), want_aux=True)
                out[(n, prec)] = (z.clone(), idx.clone())
            za, ia = out[(n, "f16x3")]
            zb, ib = out[(n, "f32")]
            assert za.shape == zb.shape
            err = (za - zb).abs().max().item()
            print(f"n={n}: pre-VQ bottleneck, split-f16 vs f32: max abs diff {err:.2e}; indices equal: {(ia == ib).float().mean().item():.4f}")
            assert err < 2e-3 and (ia == ib).float().mean() > 0.99
    finally:
        bx.w2v2_precision = old


@pytest.mark.parametrize("n", [4005, 16000, 4010])
def test_conv0_fused_with_layernorm_gives_the_same_bits(n):
    """sat_w2v2_conv0_layernorm_f32 = sat_layernorm_channels(_planes)_f32 of sat_w2v2_conv0_f32, bit for bit"""
    from satools_amd import ops
    x = _rand(2, n, seed=1, scale=0.3).to(DEV)
    w, b = _rand(512, 10, seed=2, scale=0.3).to(DEV), _rand(512, seed=3, scale=0.1).to(DEV)
    g, beta = (1 + _rand(512, seed=4, scale=0.1)).to(DEV), _rand(512, seed=5, scale=0.1).to(DEV)
    y0, ys0 = ops.layernorm_ch(ops.w2v2_conv0(x, w, b), g, beta, gelu=True, split_phases=True, planes=True)
    y1, ys1 = ops.w2v2_conv0_ln(x, w, b, g, beta, want_f32=True)
    assert y0.shape == y1.shape and torch.equal(y0, y1) and torch.equal(ys0, ys1)
    _, ys2 = ops.w2v2_conv0_ln(x, w, b, g, beta)
    assert torch.equal(ys0, ys2)


def test_asr_forward_matches_reference(model, gold):
    """f4 for the wav2vec2-tag net: `TdnnfWav2vec2VqNet.forward` (tdnnf_wav2vec2_vq.py:316-345) against the reference's
    own `Net.forward` run — chain output and xent log-softmax over 3280 pdfs, 1.5x subsampling included; the frame
    count of the reference's validate_model check (2 x 32000 samples -> 66 frames)"""
    from satools_amd import synthetic
    fx = gold.npz("fx_w2v2.npz")
    chain, xent = model.bn_extractor(synthetic.harm_batch([0, 1], 16000).to(DEV))
    assert chain.shape == xent.shape and chain.shape[2] == 3280
    for got, key in ((chain, "chain_sub"), (xent, "xent_sub")):
        want = torch.from_numpy(fx[f"harm01_16000/{key}"])
        g = got.cpu()[..., ::8]
        assert g.shape == want.shape
        err = (g - want).pow(2).mean().sqrt().item() / want.pow(2).mean().sqrt().item()
        print(f"wav2vec2-tag ASR forward {key}: relative RMS error {err:.2e}")
        assert err < 1e-4
    assert (torch.logsumexp(xent.cpu(), dim=2) - torch.from_numpy(fx["harm01_16000/xent_lse"])).abs().max() < 1e-3
    x = (torch.arange(2 * 32000, dtype=torch.float32).reshape(2, 32000) / 64000.0).to(DEV)
    assert list(model.bn_extractor(x)[0].shape) == gold.json("fx_shapes_w2v2.json")["forward_2x32000"] == [2, 66, 3280]


def test_fast_gelu_accuracy():
    """the GELU of the epilogues (branch-free erf: Abramowitz-Stegun 7.1.26 above |x| = 0.6, Taylor below) against
    torch's exact-erf GELU in float64, isolated from every matrix product: LayerNorm+GELU kernel vs GELU of the same
    kernel's plain LayerNorm output"""
    from satools_amd import ops
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(2, 512, 400, generator=g) * 2.5).to(DEV)
    gamma, beta = (1 + 0.5 * torch.randn(512, generator=g)).to(DEV), (0.5 * torch.randn(512, generator=g)).to(DEV)
    ln = ops.layernorm_ch(x, gamma, beta)
    got = ops.layernorm_ch(x, gamma, beta, gelu=True).cpu().double()
    ref = F.gelu(ln.cpu().double())
    err = (got - ref).abs().max().item()
    print("fast GELU max abs error vs float64 exact GELU:", err, "(inputs up to", ln.abs().max().item(), ")")
    assert err < 6e-7


def test_full_size_batch_properties_w2v2_tag(model):
    """BASELINE configs[2] at full size (32 x 5 s; grids, the 31-bit fast-epilogue addressing switch and the 1 GB conv-0
    tensor differ from the B <= 2 cases): (1) the bottleneck extractor treats utterances independently up to the
    reference's own `pad_input` batch mixing — the wav2vec2 features of any slice equal those rows of the full batch,
    bit for bit; (2) `convert` is deterministic, finite, in range, of the right shape; (3) VQ indices of three
    utterances equal those of one-utterance calls wherever the padding frames cannot reach (frames 3 .. T - 4)."""
    from satools_amd import synthetic
    from satools_amd.wav2vec2 import CONV_LAYERS
    seeds = list(range(32))
    wav = synthetic.harm_batch(seeds).to(DEV)
    targets = synthetic.targets(model.spk, seeds)
    y = model.convert(wav, target=targets)
    assert y.shape == (32, 1, 80001) and bool(torch.isfinite(y).all()) and float(y.abs().max()) <= 1.0
    assert torch.equal(y, model.convert(wav, target=targets))
    ext = model.bn_extractor
    lens, t = [], 80000
    for _, k, s in CONV_LAYERS:
        t = (t - k) // s + 1
        lens.append(t)
    ext._fe_len = lens
    full = ext.w2v2_features(wav).clone()                  # [32, 1024, 249]
    assert full.shape == (32, 1024, 249)
    for sl in (slice(0, 1), slice(5, 9), slice(29, 32)):
        assert torch.equal(ext.w2v2_features(wav[sl].contiguous()), full[sl])
    _, (z, idx, dist) = ext.extract_bn(wav.clone(), want_aux=True)
    assert idx.shape == (32, 250)
    for i in (0, 17, 31):
        _, (_, idx1, _) = ext.extract_bn(wav[i:i + 1].clone(), want_aux=True)
        assert torch.equal(idx1[0, 3:-3], idx[i, 3:-3]), i


def test_converts_in_flight_on_separate_streams_equal_serial(model):
    """the benchmark's mode on the wav2vec2 tag (bench.py run_steps, the reference's jobs_per_compute_device): eight convert() batches issued
    round-robin onto four HIP streams from one host thread, different inputs per batch, against the same calls made
    one after the other on the default stream — same bits (per-stream workspaces, the F0 side streams and the
    weight caches are shared state that concurrent jobs must not trample)"""
    from satools_amd import synthetic
    batches = [synthetic.harm_batch(list(range(4 * j, 4 * j + 4)), 16000 + 1600 * (j % 3)).to(DEV) for j in range(8)]
    tg = [synthetic.targets(model.spk, list(range(4 * j, 4 * j + 4))) for j in range(8)]
    serial = [model.convert(w, target=t).clone() for w, t in zip(batches, tg)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=DEV) for _ in range(4)]
    for rnd in range(2):                       # the first round also allocates the per-stream workspaces
        outs = []
        for j, (w, t) in enumerate(zip(batches, tg)):
            with torch.cuda.stream(streams[j % 4]):
                outs.append(model.convert(w, target=t))
        torch.cuda.synchronize()
        for j in range(8):
            assert torch.equal(outs[j], serial[j]), (rnd, j)


def test_full_size_converts_in_flight_are_deterministic(model):
    """BASELINE's batch (32 x 5 s) in the benchmark's mode: twenty convert() calls round-robin on four HIP streams, every
    output bit-identical to the serial result (a race between jobs on shared state would show up as a differing batch)"""
    from satools_amd import synthetic
    seeds = list(range(32))
    wav = synthetic.harm_batch(seeds).to(DEV)
    targets = synthetic.targets(model.spk, seeds)
    ref = model.convert(wav, target=targets).clone()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=DEV) for _ in range(4)]
    outs = []
    for j in range(20):
        with torch.cuda.stream(streams[j % 4]):
            outs.append(model.convert(wav, target=targets))
    torch.cuda.synchronize()
    bad = [j for j, o in enumerate(outs) if not torch.equal(o, ref)]
    assert not bad, bad


def test_convert_w2v2_tag_quant_awgn_matches_fixture(gold):
    """BASELINE configs[3]: wav2vec2 tag + f0-transformation=quant_16_awgn_2, F0 computed on the path, against the
    reference's own `convert` run under torch.manual_seed(1234)"""
    import satools_amd
    from satools_amd import synthetic
    fx = gold.npz("fx_w2v2.npz")
    m = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_wav2vec2_vq_48_v1", option_args={"f0_transformation": "quant_16_awgn_2"})
    m.to(DEV)
    m.eval()
    torch.manual_seed(1234)
    y = m.convert(synthetic.harm_batch([0, 1], 16000).to(DEV), target=[m.spk[3], m.spk[10]])
    ref = fx["harm01_16000/convert_quant16_awgn2_seed1234"]
    assert y.shape == ref.shape == (2, 1, 16001)
    err = rms(y.cpu().numpy() - ref)
    print("wav2vec2-tag convert quant_16_awgn_2 RMS error vs reference run:", err)
    assert err < 1e-4
