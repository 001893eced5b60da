"""x-vector extractor (ECAPA-TDNN, SURVEY row aX / §8 f3) on the HIP device against outputs of the reference's own
Net (tests/golden/fx_xvector.npz) and the CPU oracle.  Needs a real MI355X: run with `-m gpu`."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def net():
    import satools_amd  # noqa: F401
    from satools_amd import synthetic, xvector
    m = xvector.build()(num_speakers=10)
    m.load_state_dict(synthetic.xvector_state(0, 10), strict=True)
    return m.to(DEV)


def test_conv_relu_then_batchnorm_epilogue():
    """relu_first: conv -> ReLU -> folded BatchNorm (sidekit/nn.py:106-118), the order ECAPA-TDNN uses"""
    import satools_amd  # noqa: F401
    from satools_amd import ops, packing
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 77, generator=g)
    w = torch.randn(48, 64, 3, generator=g) * 0.1
    sc, sh = 0.5 + torch.rand(48, generator=g), torch.randn(48, generator=g) * 0.2
    ref = F.relu(F.conv1d(x, w, None, padding=2, dilation=2)) * sc[None, :, None] + sh[None, :, None]
    y = ops.conv1d(x.to(DEV), packing.pack_conv_weight(w.to(DEV)), 48, 3, pad_left=2, dilation=2, ch_scale=sc.to(DEV),
                   ch_shift=sh.to(DEV), relu=True, relu_first=True)
    assert (y.cpu() - ref).abs().max() < 2e-5


def test_front_end_matches_reference(net):
    from satools_amd import synthetic
    fx = np.load(os.path.join(GOLD, "fx_xvector.npz"))
    for tag, seed, n in (("harm0_16000", 0, 16000), ("harm7_24123", 7, 24123)):
        feats = net.features(synthetic.harm_batch([seed], n).to(DEV)).cpu().numpy()
        ref = fx[tag + "/feats"]
        assert feats.shape == ref.shape
        err = np.abs(feats - ref).max()
        print(tag, "log-mel + InstanceNorm max abs error vs reference:", err)
        assert err < 5e-5          # torch.stft (pocketfft) vs the in-LDS radix-2 FFT, through log and the normalisation (measured 3e-6)


def test_xvector_matches_reference_and_oracle(net):
    from satools_amd import synthetic
    from oracle import xvector as ox
    fx = np.load(os.path.join(GOLD, "fx_xvector.npz"))
    sd = synthetic.xvector_state(0, 10)
    for tag, seed, n in (("harm0_16000", 0, 16000), ("harm3_48000", 3, 48000), ("harm7_24123", 7, 24123)):
        wav = synthetic.harm_batch([seed], n)
        (loss, logits), xv = net(wav[0].to(DEV))
        assert xv.shape == (1, 192) and torch.isnan(loss) and logits is None
        got = xv.cpu().numpy()
        ref = fx[tag + "/xvector"]
        cos = float((got * ref).sum())
        err = np.abs(got - ref).max()
        print(f"{tag}: max abs error vs reference {err:.2e}, cosine {cos:.7f}; vs oracle {np.abs(got - ox.xvector(sd, wav).numpy()).max():.2e}")
        assert err < 5e-6 and cos > 0.999999          # measured 3-4e-7
        assert abs(float(np.linalg.norm(got)) - 1.0) < 1e-5
    # a batch of equal-length utterances = the one-utterance calls (the reference extracts with batch size 1)
    wav = synthetic.harm_batch([1, 2], 16000).to(DEV)
    both = net(wav)[1]
    one = net(wav[1])[1]
    assert torch.allclose(both[1], one[0], atol=1e-6)


def test_cpu_input_is_refused(net):
    from satools_amd._lib import SatError
    with pytest.raises(SatError):
        net(torch.zeros(16000))


def test_reference_checkpoint_format_loads(tmp_path, net):
    """a checkpoint in the reference's dict format (base_model_path = the ECAPA model config) goes through load_model"""
    import satools_amd
    from satools_amd import synthetic
    ck = {"task_path": "/egs/asv/voxceleb", "base_model_path": "local/tuning/ecapa_tdnn.py", "base_model_params": {"num_speakers": 10},
          "base_model_args": {"fine_tune": "false"}, "base_model_state_dict": synthetic.xvector_state(0, 10)}
    torch.save(ck, tmp_path / "final.pt")
    m = satools_amd.load_model(str(tmp_path / "final.pt")).to(DEV)
    wav = synthetic.harm_batch([0], 16000)[0].to(DEV)
    assert torch.equal(m(wav)[1], net(wav)[1])


def test_linear_rows_is_the_linear_layer_on_pooled_vectors():
    """sat_linear_rows_f32 (SE_Connect's two Linear layers, the embedding layer with its eval BatchNorm as an affine) against float64:
    vector and scalar forms, batch groups with a ragged tail, every epilogue option"""
    from satools_amd import ops
    from satools_amd._lib import SatError
    g = torch.Generator().manual_seed(3)
    for B, cin, cout in ((1, 512, 256), (32, 256, 512), (37, 3072, 192), (9, 130, 7), (8, 4, 1)):
        x = torch.randn(B, cin, generator=g)
        w = torch.randn(cout, cin, generator=g) * cin ** -0.5
        b, sc, sh = torch.randn(cout, generator=g), torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
        ref = x.double() @ w.double().t()
        for kw, exp in (({}, ref), ({"bias": b}, ref + b.double()), ({"bias": b, "relu": True}, torch.relu(ref + b.double())),
                        ({"ch_scale": sc, "ch_shift": sh}, ref * sc.double() + sh.double()),
                        ({"bias": b, "relu": True, "ch_scale": sc}, torch.relu(ref + b.double()) * sc.double())):
            dev = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
            got = ops.linear_rows(x.cuda(), w.cuda(), **dev)
            assert got.shape == (B, cout) and float((got.cpu().double() - exp).abs().max()) < 2e-6 * max(1.0, float(exp.abs().max())), (B, cin, cout, list(kw))
        col = ops.linear_rows(x.cuda().unsqueeze(2), w.cuda(), bias=b.cuda())
        assert col.shape == (B, cout, 1) and torch.equal(col[:, :, 0], ops.linear_rows(x.cuda(), w.cuda(), bias=b.cuda()))
    with pytest.raises(SatError):
        ops.linear_rows(torch.zeros(2, 8, device="cuda"), torch.zeros(4, 9, device="cuda"))
    with pytest.raises(SatError):
        ops.linear_rows(torch.zeros(2, 8, device="cuda"), torch.zeros(4, 8, device="cuda"), ch_shift=torch.zeros(4, device="cuda"))


def test_res2_chain_is_the_chain_of_convs():
    """sat_res2_chain_f32 (Res2Conv1dReluBn of 64-channel pieces in one launch) against the module's arithmetic in float64 — lengths around
    the window edges (64- and 128-column centres), utterances shorter than a window, every dilation the staged halo allows — and the net's
    embedding with and without it"""
    from satools_amd import ops
    from satools_amd._lib import SatError
    g = torch.Generator().manual_seed(5)
    for B, T, nums, dil in ((2, 500, 7, 2), (3, 64, 7, 4), (1, 65, 7, 3), (2, 129, 7, 4), (1, 7, 7, 4), (33, 260, 3, 1), (64, 1030, 7, 4)):
        C = (nums + 1) * 64
        y = torch.randn(B, C, T, generator=g)
        w = torch.randn(nums, 64, 64, 3, generator=g) * (64 * 3) ** -0.5          # Conv1d.weight layout per piece: [co][ci][tap]
        sc, sh = torch.rand(nums, 64, generator=g) + 0.5, torch.randn(nums, 64, generator=g) * 0.1
        ref, sp = [], None
        for i in range(nums):
            xin = y[:, 64 * i:64 * (i + 1)].double() + (sp if sp is not None else 0)
            sp = F.conv1d(xin, w[i].double(), padding=dil, dilation=dil)
            sp = torch.relu(sp) * sc[i].double().view(1, -1, 1) + sh[i].double().view(1, -1, 1)
            ref.append(sp)
        ref.append(y[:, 64 * nums:].double())
        ref = torch.cat(ref, 1)
        got = ops.res2_chain(y.cuda(), w.permute(0, 3, 2, 1).contiguous().cuda(), sc.cuda(), sh.cuda(), dil)
        err = float((got.cpu().double() - ref).abs().max())
        assert got.shape == y.shape and err < 5e-6 * max(1.0, float(ref.abs().max())), (B, T, nums, dil, err)
        assert torch.equal(got[:, 64 * nums:].cpu(), y[:, 64 * nums:])
    with pytest.raises(SatError):
        ops.res2_chain(torch.zeros(1, 512, 40, device="cuda"), torch.zeros(7, 3, 64, 64, device="cuda"), torch.zeros(7, 64, device="cuda"),
                       torch.zeros(7, 64, device="cuda"), 5)


def test_embedding_with_and_without_the_fused_chain(net):
    from satools_amd import synthetic
    wav = synthetic.harm_batch([0, 1, 2], 48000).to("cuda")
    try:
        net.res2_chain = True
        a = net(wav)[1]
        net.res2_chain = False
        b = net(wav)[1]
    finally:
        net.res2_chain = True
    assert float((a - b).abs().max()) < 2e-6
