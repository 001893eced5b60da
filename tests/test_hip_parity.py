"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed golden
vectors.  Needs a real MI355X: run with `-m gpu`."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rms

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from satools_amd import ops, packing
    return ops, packing


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _lib_err():
    from satools_amd import _lib
    return _lib.SatError


class conv_option:
    """sat_conv_set_option(name, value) for the duration of a with-block (restored to `default` afterwards)"""

    def __init__(self, name, value, default):
        self.name, self.value, self.default = name.encode(), value, default

    def __enter__(self):
        from satools_amd import _lib
        _lib.check(_lib.lib().sat_conv_set_option(self.name, self.value), "sat_conv_set_option")

    def __exit__(self, *a):
        from satools_amd import _lib
        _lib.check(_lib.lib().sat_conv_set_option(self.name, self.default), "sat_conv_set_option")



# ---------------------------------------------------------------------------------------------
# fused conv1d kernel vs torch (f32 CPU) on the shapes the path uses
# tolerance: f32 re-association only: |err| <= 2e-5 * (|x| . |w|) scale
# ---------------------------------------------------------------------------------------------
CONV_CASES = [
    # (C_in, C_out, K, dilation, stride, pad_left, pad_right, T, B)   generator shapes
    (16, 16, 3, 1, 1, 1, 1, 700, 2), (16, 16, 7, 3, 1, 9, 9, 1000, 2), (16, 16, 11, 5, 1, 25, 25, 1100, 1),
    (32, 32, 3, 5, 1, 5, 5, 600, 2), (32, 32, 11, 1, 1, 5, 5, 513, 2),
    (64, 64, 7, 1, 1, 3, 3, 300, 2), (64, 64, 11, 3, 1, 15, 15, 257, 1),
    (128, 128, 3, 3, 1, 3, 3, 260, 2), (128, 128, 11, 5, 1, 25, 25, 129, 2),
    (256, 256, 7, 5, 1, 15, 15, 125, 2), (256, 256, 11, 1, 1, 5, 5, 250, 1),
    (504, 512, 7, 1, 1, 3, 3, 50, 2),
    # TDNNF shapes: valid conv over frames, stride 2 layer, 1x1
    (80, 128, 3, 1, 1, 0, 0, 88, 2), (1024, 128, 3, 1, 1, 0, 0, 70, 2), (1024, 128, 1, 1, 2, 0, 0, 75, 2),
    (128, 1024, 1, 1, 1, 0, 0, 66, 2), (1024, 256, 3, 1, 1, 0, 0, 40, 1),
    # odd tap counts through the runtime-tap kernel, strided
    (512, 512, 2, 1, 2, 0, 0, 99, 2), (64, 96, 10, 1, 5, 0, 0, 400, 1), (48, 40, 5, 2, 1, 4, 4, 150, 2),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv1d_matches_torch(case):
    ops, packing = _ops()
    cin, cout, k, d, s, pl, pr, T, B = case
    x = _rand(B, cin, T, seed=1)
    w = _rand(cout, cin, k, seed=2, scale=1.0 / np.sqrt(cin * k))
    b = _rand(cout, seed=3)
    ref = F.conv1d(F.pad(x, (pl, pr)), w, b, stride=s, dilation=d)
    wp = packing.pack_conv_weight(w.to(DEV))
    y = ops.conv1d(x.to(DEV), wp, cout, k, bias=b.to(DEV), dilation=d, stride=s, pad_left=pl, pad_right=pr)
    assert y.shape == ref.shape
    assert (y.cpu() - ref).abs().max() < 3e-5


F16X3_CASES = [(16, 16, 3, 1, 700), (16, 16, 11, 5, 1100), (32, 32, 7, 3, 600), (64, 64, 11, 1, 257), (128, 128, 3, 5, 260),
               (256, 256, 7, 5, 125), (256, 256, 11, 3, 250), (504, 512, 7, 1, 50)]


@pytest.mark.parametrize("case", F16X3_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv1d_split_f16_matches_torch(case):
    """SAT_CONV_F16X3: operands carried as hi+lo f16 (22 bits), products hi*hi + hi*lo + lo*hi with f32
    accumulation.  Bound: 2^-21 per product on top of f32 re-association -> same 3e-5 absolute bar as
    the exact-f32 kernel on these O(1) outputs, and an RMS error within 2x of the f32 kernel's"""
    ops, packing = _ops()
    cin, cout, k, d, T = case
    x = _rand(2, cin, T, seed=1)
    w = _rand(cout, cin, k, seed=2, scale=1.0 / np.sqrt(cin * k))
    b = _rand(cout, seed=3)
    res = _rand(2, cout, T, seed=4)
    pl = (k * d - d) // 2
    ref = (F.conv1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), dilation=d, padding=pl) + res.double())
    y16 = ops.conv1d(x.to(DEV), packing.pack_conv_weight_f16x3(w.to(DEV)), cout, k, bias=b.to(DEV), dilation=d,
                     pad_left=pl, in_lrelu=0.1, res=res.to(DEV), mode=1)
    y32 = ops.conv1d(x.to(DEV), packing.pack_conv_weight(w.to(DEV)), cout, k, bias=b.to(DEV), dilation=d,
                     pad_left=pl, in_lrelu=0.1, res=res.to(DEV))
    e16, e32 = rms(y16.cpu().double() - ref), rms(y32.cpu().double() - ref)
    print(f"rms error vs f64: split-f16 {e16:.2e}, exact f32 {e32:.2e}")
    assert (y16.cpu().double() - ref).abs().max() < 3e-5
    assert e16 < 2.5 * e32 + 1e-8


@pytest.mark.parametrize("case", [(16, 3, 1, 1000), (16, 7, 3, 700), (16, 11, 5, 2500), (32, 3, 5, 449), (32, 7, 1, 224), (32, 11, 3, 1500)],
                         ids=lambda c: "x".join(map(str, c)))
def test_fused_resblock_pair_matches_torch(case):
    """out = conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 + x in one kernel (thin generator stages)"""
    ops, packing = _ops()
    C, k, d, T = case
    x = _rand(2, C, T, seed=1)
    w1, w2 = _rand(C, C, k, seed=2, scale=0.6 / np.sqrt(C * k)), _rand(C, C, k, seed=3, scale=0.6 / np.sqrt(C * k))
    b1, b2 = _rand(C, seed=4, scale=0.1), _rand(C, seed=5, scale=0.1)
    xd = x.double()
    t1 = F.conv1d(F.leaky_relu(xd, 0.1), w1.double(), b1.double(), dilation=d, padding=(k * d - d) // 2)
    ref = F.conv1d(F.leaky_relu(t1, 0.1), w2.double(), b2.double(), padding=(k - 1) // 2) + xd
    pk = packing.pack_conv_weight_f16x3
    y = ops.resblock_pair(x.to(DEV), pk(w1.to(DEV)), b1.to(DEV), pk(w2.to(DEV)), b2.to(DEV), k, d)
    err = (y.cpu().double() - ref).abs().max().item()
    print("max abs err vs f64:", err)
    assert err < 2e-5
    acc0 = _rand(2, C, T, seed=6).to(DEV)
    out = acc0.clone()
    ops.resblock_pair(x.to(DEV), pk(w1.to(DEV)), b1.to(DEV), pk(w2.to(DEV)), b2.to(DEV), k, d, out=out, accum=True, accum_div=3.0)
    assert (out.cpu().double() - (acc0.cpu().double() + ref) / 3).abs().max() < 2e-5


def test_conv1d_fused_prologue_epilogue():
    ops, packing = _ops()
    B, C, T, k, d = 2, 64, 333, 7, 3
    x, w, b = _rand(B, C, T, seed=1), _rand(C, C, k, seed=2, scale=0.05), _rand(C, seed=3)
    res = _rand(B, C, T, seed=4)
    acc0 = _rand(B, C, T, seed=5)
    ref = F.conv1d(F.leaky_relu(x, 0.1), w, b, dilation=d, padding=(k * d - d) // 2) + res
    wp = packing.pack_conv_weight(w.to(DEV))
    y = ops.conv1d(x.to(DEV), wp, C, k, bias=b.to(DEV), dilation=d, pad_left=(k * d - d) // 2, in_lrelu=0.1, res=res.to(DEV))
    assert (y.cpu() - ref).abs().max() < 3e-5
    # MRF accumulation: dst = (dst + v) / 3
    out = acc0.to(DEV).clone()
    ops.conv1d(x.to(DEV), wp, C, k, bias=b.to(DEV), dilation=d, pad_left=(k * d - d) // 2, in_lrelu=0.1, res=res.to(DEV),
               out=out, accum=True, accum_div=3.0)
    assert (out.cpu() - (acc0 + ref) / 3).abs().max() < 3e-5
    # TDNNF epilogue: bypass with frame offset + stride, folded BatchNorm, ReLU
    xin = _rand(B, 128, 90, seed=6)
    w1 = _rand(128, 32, 1, seed=7, scale=0.2)
    z = _rand(B, 32, 44, seed=8)
    sc, sh = _rand(128, seed=9).abs() + 0.5, _rand(128, seed=10)
    ref = F.relu((F.conv1d(z, w1, None) + 0.66 * xin[:, :, 1:89:2]) * sc[None, :, None] + sh[None, :, None])
    y = ops.conv1d(z.to(DEV), packing.pack_conv_weight(w1.to(DEV)), 128, 1, res=xin.to(DEV), res_scale=0.66, res_toff=1,
                   res_tstride=2, ch_scale=sc.to(DEV), ch_shift=sh.to(DEV), relu=True)
    assert (y.cpu() - ref).abs().max() < 3e-5


def test_conv1d_groups():
    ops, packing = _ops()
    B, C, T, k, g = 2, 128, 120, 16, 4
    x, w, b = _rand(B, C, T, seed=1), _rand(C, C // g, k, seed=2, scale=0.05), _rand(C, seed=3)
    ref = F.conv1d(x, w, b, padding=k // 2, groups=g)
    y = ops.conv1d(x.to(DEV), packing.pack_conv_weight(w.to(DEV), groups=g), C, k, bias=b.to(DEV), pad_left=k // 2,
                   pad_right=k // 2, groups=g)
    assert y.shape == ref.shape
    assert (y.cpu() - ref).abs().max() < 3e-5


@pytest.mark.parametrize("cfg", [(64, 32, 11, 5, 77), (128, 64, 8, 4, 50), (32, 16, 4, 2, 301), (512, 256, 11, 5, 25)])
def test_convtranspose_as_polyphase_conv(cfg):
    ops, packing = _ops()
    cin, cout, k, u, T = cfg
    x = _rand(2, cin, T, seed=1)
    w = _rand(cin, cout, k, seed=2, scale=1.0 / np.sqrt(cin * k / u))
    b = _rand(cout, seed=3)
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=u, padding=(k - u) // 2)
    wc, kp, pl = packing.convtranspose_as_phase_conv(w.to(DEV), u, (k - u) // 2)
    y = ops.conv1d(x.to(DEV), packing.pack_conv_weight(wc, up=u), cout, kp, bias=b.to(DEV), pad_left=pl, up=u, in_lrelu=0.1)
    assert y.shape == ref.shape == (2, cout, T * u)
    assert (y.cpu() - ref).abs().max() < 3e-5


@pytest.mark.parametrize("cfg", [(128, 64, 8, 4, 50), (256, 128, 8, 4, 333), (64, 32, 4, 2, 700), (32, 16, 4, 2, 1301)])
def test_polyphase_conv_writes_split_planes(cfg):
    """transposed conv (rates 2 and 4) from split planes straight to split planes through the LDS-transposed
    epilogue: bit-identical to the f32 output of the same kernel put through the split pass"""
    ops, packing = _ops()
    cin, cout, k, u, T = cfg
    x = _rand(2, cin, T, seed=1).to(DEV)
    w = _rand(cin, cout, k, seed=2, scale=1.0 / np.sqrt(cin * k / u)).to(DEV)
    b = _rand(cout, seed=3).to(DEV)
    wc, kp, pl = packing.convtranspose_as_phase_conv(w, u, (k - u) // 2)
    wp = packing.pack_conv_weight_f16x3(wc, up=u)
    xs = ops.act_split(x, 0.1)
    y = ops.conv1d(x, wp, cout, kp, bias=b, pad_left=pl, up=u, mode=1, x_split=xs)
    ref = F.conv_transpose1d(F.leaky_relu(x.cpu(), 0.1), w.cpu(), b.cpu(), stride=u, padding=(k - u) // 2)
    assert (y.cpu() - ref).abs().max() < 3e-5
    ys = torch.zeros(2, cout // 16, 2, 2, T * u, 8, dtype=torch.float16, device=DEV)
    guard = torch.full_like(y, 7.0)
    ops.conv1d(x, wp, cout, kp, bias=b, pad_left=pl, up=u, mode=1, x_split=xs, y_split=ys, y_split_slope=0.1, no_y=True, out=guard)
    assert (guard == 7.0).all()
    assert torch.equal(ys, ops.act_split(y, 0.1))


@pytest.mark.parametrize("cfg", [(256, 128, 133, 2), (128, 64, 700, 3), (256, 128, 161, 1), (128, 64, 5000, 2)])
def test_stride4_upsampler_on_the_ring_with_rows_grouped_by_phase(cfg):
    """ConvTranspose1d(k 8, stride 4, padding 2) planes to planes on the LDS-DMA ring (conv_ring16.hip, sat_conv1d_desc.up_grouped):
    rows ordered (16-channel group, phase, channel), the plane units transposed inside the quads of lanes before the stores, the
    all-zero (tap slot, phase) products left out.  Against float64, against the LDS-transposed epilogue of the 64 x 256 tile
    (another accumulation order: f32 rounding), and the instantiation without the zero products against the one that multiplies
    everything (the same bits)."""
    ops, packing = _ops()
    from satools_amd import _lib
    cin, cout, T, B = cfg
    k, u = 8, 4
    pad = (k - u) // 2
    assert packing.upsample_grouped_supported(cin, cout, k, u, pad) and not packing.upsample_grouped_supported(cin, cout, 4, 2, 1)
    assert not packing.upsample_grouped_supported(64, 32, k, u, pad) and not packing.upsample_grouped_supported(96, 64, k, u, pad)
    zt = packing.convtranspose_zero_taps(k, u, pad)
    assert zt == 0x30c            # phases 0, 1: slots 0, 1; phases 2, 3: slots 1, 2
    x = _rand(B, cin, T, seed=1).to(DEV)
    w = _rand(cin, cout, k, seed=2, scale=1.0 / np.sqrt(cin * k / u)).to(DEV)
    b = _rand(cout, seed=3).to(DEV)
    xs = ops.act_split(x, 0.1)
    wc, kp, pl = packing.convtranspose_as_phase_conv(w, u, pad)
    wg, kp2, pl2 = packing.convtranspose_as_phase_conv(w, u, pad, grouped=True)
    assert (kp, pl) == (kp2, pl2) == (3, 1)
    # the mask names exactly the zero (slot, phase) pairs of the polyphase weight
    for slot in range(kp):
        for r in range(u):
            assert bool((wc[r::u, :, slot] == 0).all()) == bool(zt >> (slot * 4 + r) & 1)
    y_tile = ops.split_like(B, cout, T * u, DEV)
    ops.conv1d(x, packing.pack_conv_weight_f16x3(wc, up=u), cout, kp, bias=b, pad_left=pl, up=u, mode=1, x_split=xs, y_split=y_tile,
               y_split_slope=0.1, no_y=True)
    wpg = packing.pack_conv_weight_f16x3(wg, up=u)
    outs = []
    for mask in (zt, 0, 0x300):   # (a subset of the true zeros is not the pattern with an instantiation of its own: everything is multiplied)
        ys = torch.full((B, cout // 16, 2, 2, T * u, 8), 7.0, dtype=torch.float16, device=DEV)
        guard = torch.full((B, cout, T * u), 7.0, device=DEV)
        ops.conv1d(x, wpg, cout, kp, bias=b, pad_left=pl, up=u, mode=1, x_split=xs, y_split=ys, y_split_slope=0.1, no_y=True, out=guard,
                   up_grouped=True, up_zero_taps=mask)
        assert "UMASK = %d" % (zt if mask == zt else 0) in _lib.lib().sat_last_dispatch_name().decode()
        assert (guard == 7.0).all()
        outs.append(ys)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # a mask that claims a (slot, phase) pair the packed weights do NOT have all-zero is refused (the packer recorded the true zeros)
    with pytest.raises(_lib.SatError):
        ops.conv1d(x, wpg, cout, kp, bias=b, pad_left=pl, up=u, mode=1, x_split=xs, y_split=ys, y_split_slope=0.1, no_y=True,
                   up_grouped=True, up_zero_taps=0x1)
    ref = F.leaky_relu(F.conv_transpose1d(F.leaky_relu(x.double().cpu(), 0.1), w.double().cpu(), b.double().cpu(), stride=u, padding=pad), 0.1)
    got = ops.unsplit(outs[0]).double().cpu()
    assert (got - ref).abs().max() < 4e-6
    assert (got - ops.unsplit(y_tile).double().cpu()).abs().max() < 4e-6
    # what the layout needs is said, not guessed
    with pytest.raises(_lib.SatError):
        ops.conv1d(x, wpg, cout, kp, bias=b, pad_left=pl, up=u, mode=1, x_split=xs, up_grouped=True)                    # f32 output
    with pytest.raises(_lib.SatError):
        ops.conv1d(x, wpg, cout, kp, bias=b, pad_left=pl, up=u, mode=0, up_grouped=True)                               # exact-f32 mode


def test_convpost_matches_oracle():
    ops, _ = _ops()
    x, w, b = _rand(2, 16, 2500, seed=1), _rand(1, 16, 7, seed=2, scale=0.1), _rand(1, seed=3, scale=0.1)
    ref = torch.tanh(F.conv1d(F.pad(F.leaky_relu(x), (1, 0), mode="reflect"), w, b, padding=3))
    y = ops.convpost(x.to(DEV), w.reshape(16, 7).contiguous().to(DEV), b.to(DEV))
    assert y.shape == ref.shape == (2, 1, 2501)
    assert (y.cpu() - ref).abs().max() < 2e-6


@pytest.mark.parametrize("T", [2, 5, 1021, 1024, 1027, 4099])
def test_convpost_four_samples_per_lane_gives_the_bits_of_the_one_sample_form(T):
    """the output stage's two forms (sat_conv_set_option("convpost_quad")): T + 1 outputs around the 1024-sample tile edges, a tile with
    fewer than four outputs, the reflected first sample — bit for bit, and against torch"""
    ops, _ = _ops()
    x, w, b = _rand(3, 16, T, seed=T), _rand(1, 16, 7, seed=2, scale=0.1), _rand(1, seed=3, scale=0.1)
    ref = torch.tanh(F.conv1d(F.pad(F.leaky_relu(x), (1, 0), mode="reflect"), w, b, padding=3))
    ys = []
    for quad in (0, 1):
        with conv_option("convpost_quad", quad, 1):
            ys.append(ops.convpost(x.to(DEV), w.reshape(16, 7).contiguous().to(DEV), b.to(DEV)).cpu())
    assert ys[0].shape == ref.shape == (3, 1, T + 1)
    assert torch.equal(ys[0], ys[1])
    assert (ys[1] - ref).abs().max() < 2e-6


def test_conv_pre_on_half_width_tiles_gives_the_same_bits():
    """the 7-tap f32-input conv of the generator (conv_pre) on 64 x 128 tiles (sat_conv_set_option("half_tile7"), taken when the 64 x 256
    tiles are at most one per CU) against the 64 x 256 form: a ragged last tile, frames that end inside the first half tile"""
    ops, packing = _ops()
    for B, T in ((2, 250), (3, 129), (1, 300)):
        x = _rand(B, 80, T, seed=T).to(DEV)
        w = packing.pack_conv_weight_f16x3((_rand(512, 80, 7, seed=5, scale=0.05)).to(DEV))
        b = _rand(512, seed=6, scale=0.1).to(DEV)
        ys = []
        for v in (0, 1):
            with conv_option("half_tile7", v, 1):
                ys.append(ops.conv1d(x, w, 512, 7, bias=b, pad_left=3, pad_right=3, mode=1).cpu())
        assert torch.equal(ys[0], ys[1]), (B, T)
        ref = F.conv1d(x.cpu().double(), _rand(512, 80, 7, seed=5, scale=0.05).double(), b.cpu().double(), padding=3)
        assert rms(ys[1].double().numpy() - ref.numpy()) < 1e-5 * rms(ref.numpy())


@pytest.mark.parametrize("ctx,feat,bott,out,T,bypass", [(3, 1024, 128, 1024, 252, 0.66), (3, 80, 128, 1024, 61, 0.0), (1, 256, 64, 256, 40, 0.66),
                                                       (2, 64, 40, 64, 33, 0.66)])
def test_tdnnf_layer_call_equals_its_two_launches(ctx, feat, bott, out, T, bypass):
    """sat_tdnnf_layer_f32 (one C-ABI call per TDNNF layer, chain/nn.py:267-347) against the two sat_conv1d_f32 calls it composes — bit for
    bit in exact-f32 mode and on split planes — and against torch in float64: linearB over `ctx` frames, linearA, bypass from
    identity_lidx frames in (chain/nn.py:233-247), BatchNorm (eval, no affine), ReLU"""
    ops, packing = _ops()
    B = 3
    x = _rand(B, feat, T, seed=T).relu()
    wB, wA = _rand(bott, feat, ctx, seed=1, scale=1.0 / np.sqrt(feat * ctx)), _rand(out, bott, 1, seed=2, scale=1.0 / np.sqrt(bott))
    bB, bA = _rand(bott, seed=3, scale=0.1), _rand(out, seed=4, scale=0.1)
    mean, var = _rand(out, seed=5, scale=0.1), _rand(out, seed=6).abs() + 0.5
    scale, shift = 1.0 / torch.sqrt(var + 1e-5), -mean / torch.sqrt(var + 1e-5)
    lidx = 1 if ctx == 2 else ctx // 2
    t_q = T - (ctx - 1)
    xd = x.double()
    ref = F.conv1d(F.conv1d(xd, wB.double(), bB.double()), wA.double(), bA.double())
    if bypass:
        ref = ref + bypass * xd[:, :, lidx:lidx + t_q]
    ref = F.relu(ref * scale.double()[None, :, None] + shift.double()[None, :, None])
    dev = lambda t: t.to(DEV)
    for mode, pack in ((0, packing.pack_conv_weight), (1, packing.pack_conv_weight_f16x3)):
        pB, pA = pack(dev(wB)), pack(dev(wA))
        planes = mode == 1 and bott % 16 == 0
        xs = ops.act_split(dev(x), 1.0) if (mode == 1 and feat % 16 == 0) else None
        kw = dict(res=dev(x), res_scale=bypass, res_toff=lidx) if bypass else {}
        # the two launches
        zs = ops.split_like(B, bott, t_q, DEV) if planes else None
        z = ops.conv1d(dev(x), pB, bott, ctx, bias=dev(bB), pad_left=0, pad_right=0, mode=mode, x_split=xs,
                       **(dict(y_split=zs, y_split_slope=1.0, no_y=True) if planes else {}))
        ys2 = ops.split_like(B, out, t_q, DEV) if mode == 1 and out % 16 == 0 else None
        y2 = ops.conv1d(z, pA, out, 1, bias=dev(bA), ch_scale=dev(scale), ch_shift=dev(shift), relu=True, mode=mode, x_split=zs, y_split=ys2,
                        y_split_slope=1.0, **kw)
        # the one call
        ys1 = ops.split_like(B, out, t_q, DEV) if ys2 is not None else None
        y1 = ops.tdnnf_layer(dev(x), pB, dev(bB), pA, dev(bA), bott, out, ctx, bn_scale=dev(scale), bn_shift=dev(shift), bypass_scale=bypass, mode=mode,
                             x_split=xs, y_split=ys1, z_split=ops.split_like(B, bott, t_q, DEV) if planes else None)
        assert y1.shape == (B, out, t_q) and torch.equal(y1, y2), (mode, ctx)
        if ys1 is not None:
            assert torch.equal(ys1, ys2)
        err = rms(y1.double().cpu().numpy() - ref.numpy()) / rms(ref.numpy())
        assert err < (2e-6 if mode == 0 else 5e-6), (mode, err)
        if planes and xs is not None and ys1 is not None:
            # planes only: no f32 output, the bypass rebuilt from the input planes (22 significand bits of x); `x` is a shape carrier
            ys3 = ops.split_like(B, out, t_q, DEV)
            carrier = torch.full((B, feat, T), float("nan"), device=DEV)
            y3 = ops.tdnnf_layer(carrier, pB, dev(bB), pA, dev(bA), bott, out, ctx, bn_scale=dev(scale), bn_shift=dev(shift), bypass_scale=bypass,
                                 mode=mode, x_split=xs, y_split=ys3, z_split=ops.split_like(B, bott, t_q, DEV), no_y=True, bypass_from_planes=True)
            assert y3.shape == (B, out, t_q)
            got = ops.unsplit(ys3).double().cpu().numpy()
            assert np.isfinite(got).all() and rms(got - ref.numpy()) / rms(ref.numpy()) < 5e-6
            if not bypass:
                assert torch.equal(ys3, ys1)
    if feat % 16 == 0:
        with pytest.raises(_lib_err()):
            ops.tdnnf_layer(dev(x), packing.pack_conv_weight(dev(wB)), dev(bB), packing.pack_conv_weight(dev(wA)), dev(bA), bott, out, ctx,
                            bypass_scale=bypass, mode=0, bypass_from_planes=True)


# ---------------------------------------------------------------------------------------------
# front end and bottleneck
# ---------------------------------------------------------------------------------------------
def _model(tag="hifigan_bn_tdnnf_600h_vq_48_v1", **opt):
    from satools_amd import load_model
    m = load_model("synthetic:" + tag, option_args=opt or None)
    m.to(DEV)
    m.eval()
    return m


@pytest.fixture(scope="module")
def model():
    return _model()


@pytest.fixture
def model_f16x3(model):
    """the module's model with the generator pinned to "f16x3" (the tests of that arithmetic's own bit equalities; the product default
    is "f16f8r" since round 5, whose kernels tests/test_hip_f8r.py covers)"""
    g = model.hifigan
    old = g.precision
    g.precision = "f16x3"
    g.invalidate()
    try:
        yield model
    finally:
        g.precision = old
        g.invalidate()


def test_fbank_cmvn_pad_matches_oracle(model):
    from oracle import fbank as ofb
    from oracle import tdnnf as otd
    from satools_amd import synthetic
    for wav in (synthetic.harm_batch([0], 8000), synthetic.harm_batch([0, 1, 2], 16384), synthetic.rand_batch(0, 2, 80000)):
        ref = ofb.fbank(wav * 32768, 80)
        ref = otd.pad_input(ref - ref.mean(dim=1).unsqueeze(1), 19).permute(0, 2, 1)
        got = model.bn_extractor.features((wav * 32768).to(DEV))
        assert got.shape == ref.shape
        assert (got.cpu() - ref).abs().max() < 3e-4


def test_vq_matches_oracle():
    ops, _ = _ops()
    from oracle import tdnnf as otd
    z = _rand(2, 300, 256, seed=1)
    cb = z.reshape(-1, 256)[torch.randperm(600, generator=torch.Generator().manual_seed(0))[:48]] + _rand(48, 256, seed=2, scale=0.3)
    qr, ir, dr = otd.vq(z, cb)
    q, idx, dist = ops.vq(z.permute(0, 2, 1).contiguous().to(DEV), cb.to(DEV), want_dist=True)
    dr = dr.view(2, 300, 48)
    assert (dist.cpu() - dr).abs().max() < 1e-3          # expanded-form f32 distances, |z|^2 ~ 256
    srt = dr.sort(2)[0]
    margin = srt[..., 1] - srt[..., 0]
    agree = idx.cpu().long() == ir.view(2, 300)
    assert agree[margin > 2e-3].all()
    # the margin rule, tight: an index may differ from the oracle's only where the oracle's two best distances are closer than twice
    # the largest difference between the kernel's and the oracle's distances (both are f32 evaluations of the same expanded form)
    derr = float((dist.cpu() - dr).abs().max())
    print(f"vq: {int((~agree).sum())} of {agree.numel()} indices differ; distance error {derr:.1e}; their margins {margin[~agree].tolist()}")
    assert (margin[~agree] <= 2 * derr).all()
    assert (q.cpu().permute(0, 2, 1) - qr)[agree].abs().max() < 1e-6


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_extract_bn_matches_oracle_and_golden(model, gold, fbank_tag_state, precision):
    """both TDNNF arithmetic settings: the VQ indices must be the reference's on every fixture frame"""
    model.bn_extractor.precision = precision
    try:
        _check_extract_bn(model, gold, fbank_tag_state)
    finally:
        model.bn_extractor.precision = type(model.bn_extractor).precision


def test_bottleneck_stack_on_planes_only_against_the_f32_hand_over(model):
    """inside the bottleneck stack a plain layer hands on split planes only (no f32 store, the bypass rebuilt from the input planes:
    sat_tdnnf_layer_f32 with x = y = NULL, `tdnnf_planes_only`) — against the stack whose layers also write and read the f32 tensors:
    features within the split-f16 error (1e-5 relative), the same VQ indices, and the stack of the exact-f32 setting untouched"""
    from satools_amd import synthetic
    ext = model.bn_extractor
    wav = synthetic.harm_batch([3, 4, 5], 80000).to(DEV)
    keep, sig = ext.tdnnf_planes_only, ext.vq_tie_sigmas
    ext.vq_tie_sigmas = 0.0
    try:
        res = {}
        for v in (False, True):
            ext.tdnnf_planes_only = v
            bn, (z, idx, _) = ext.extract_bn(wav.clone(), want_aux=True)
            res[v] = (bn.clone(), z.clone(), idx.clone())
        ext.precision = "f32"
        e32 = {}
        for v in (False, True):
            ext.tdnnf_planes_only = v
            e32[v] = ext.extract_bn(wav.clone()).clone()
    finally:
        ext.tdnnf_planes_only, ext.vq_tie_sigmas = keep, sig
        ext.precision = type(ext).precision
    assert torch.isfinite(res[True][1]).all()
    err = rms((res[True][1] - res[False][1]).cpu().numpy()) / rms(res[False][1].cpu().numpy())
    print(f"pre-quantiser features, planes only vs f32 hand-over: {err:.2e} relative RMS; indices equal: {bool(torch.equal(res[True][2], res[False][2]))}")
    assert 0 < err < 1e-5
    assert torch.equal(res[True][2], res[False][2])
    assert (res[True][0] - res[False][0]).abs().max() < 1e-5          # x + (q - x): the same codes, the last bit of the sum follows x
    assert torch.equal(e32[True], e32[False])


def _check_extract_bn(model, gold, fbank_tag_state):
    from oracle import convert as oconv
    from oracle import tdnnf as otd
    from satools_amd import synthetic
    state, _ = fbank_tag_state
    asr, _ = oconv.split_state_dict(state["base_model_state_dict"])
    fx = gold.npz("fx_tdnnf.npz")
    wav = synthetic.harm_batch([0, 1], 80000)
    aux = {}
    ref = otd.extract_bn_fbank(asr, wav, aux=aux)
    bn, (z, idx, dist) = model.bn_extractor.extract_bn(wav.clone().to(DEV), want_aux=True)
    assert bn.shape == ref.shape == (2, 250, 256)
    assert (z.cpu().permute(0, 2, 1) - aux["z"]).abs().max() < 2e-4   # 11 layers of f32 re-association
    margin = torch.from_numpy(fx["harm01_80000/margin"])
    agree = idx.cpu().long() == torch.from_numpy(fx["harm01_80000/idx"]).long()
    assert agree[margin > 5e-3].all()
    print("VQ index agreement with the reference:", agree.float().mean().item())
    assert agree.all()
    assert (bn.cpu() - ref)[agree].abs().max() < 2e-4
    # get_bn: [B, 256, T], input untouched (clone semantics of parse_wavinfo_wav)
    w2 = wav.clone().to(DEV)
    out = model.get_bn(w2)
    assert out.shape == (2, 256, 250) and torch.equal(w2.cpu(), wav)


def test_f0_norm_transform_matches_golden(gold):
    ops, _ = _ops()
    from satools_amd import f0_transforms
    fx = gold.npz("fx_f0norm.npz")
    for a, b in (("in_1xT", "out_1xT"), ("in_2xT", "out_2xT"), ("in_zero_row", "out_zero_row")):
        x = torch.from_numpy(fx[a].copy()).to(DEV)
        ops.f0_norm_transform_(x)
        assert np.allclose(x.cpu().numpy(), fx[b], atol=2e-6)
    x = torch.from_numpy(fx["in_2xT"].copy()).to(DEV)
    ops.f0_norm_transform_(x, quant_bins=16)
    assert np.abs(x.cpu().numpy() - fx["quant16"][:, 0]).max() < 1e-6
    torch.manual_seed(1234)
    noise = f0_transforms.draw_awgn((2, 1, x.shape[1]), 2)
    x = torch.from_numpy(fx["in_2xT"].copy()).to(DEV)
    ops.f0_norm_transform_(x, quant_bins=16, noise=noise.to(DEV))
    assert np.abs(x.cpu().numpy() - fx["quant16_awgn2_seed1234"][:, 0]).max() < 1e-6


def test_act_split_planes():
    """split planes: hi = f16(x) (round toward zero), lo = f16(x - hi); hi + lo reproduces lrelu(x) to 2^-21"""
    ops, _ = _ops()
    x = _rand(3, 32, 333, seed=7, scale=2.0)
    s = ops.act_split(x.to(DEV), 0.1)
    assert s.shape == (3, 2, 2, 2, 333, 8) and s.dtype == torch.float16
    ref = F.leaky_relu(x, 0.1)
    back = ops.unsplit(s).cpu()
    assert (back - ref).abs().max() <= 2.0 ** -20 * ref.abs().max()
    hi = ops.unsplit(torch.stack([s[:, :, 0], torch.zeros_like(s[:, :, 0])], 2)).cpu()
    assert (hi.abs() <= ref.abs()).all()          # truncation toward zero


@pytest.mark.parametrize("case", [(16, 16, 3, 1, 700), (32, 32, 7, 3, 600), (64, 64, 11, 1, 257), (256, 256, 11, 5, 250), (512, 256, 3, 1, 130)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv1d_split_planes_in_out(case):
    """the split-plane hand-over is the same arithmetic as splitting inside the kernel: a conv fed with
    act_split(lrelu(x)) is bit-identical to the conv that stages the f32 x itself, and the planes it
    writes are bit-identical to act_split of its own f32 output"""
    ops, packing = _ops()
    cin, cout, k, d, T = case
    x = _rand(2, cin, T, seed=1).to(DEV)
    w = packing.pack_conv_weight_f16x3(_rand(cout, cin, k, seed=2, scale=1.0 / np.sqrt(cin * k)).to(DEV))
    b = _rand(cout, seed=3).to(DEV)
    res = _rand(2, cout, T, seed=4).to(DEV)
    pl = (k * d - d) // 2
    y0 = ops.conv1d(x, w, cout, k, bias=b, dilation=d, pad_left=pl, in_lrelu=0.1, res=res, mode=1)
    ys = ops.split_like(2, cout, T, DEV)
    y1 = ops.conv1d(x, w, cout, k, bias=b, dilation=d, pad_left=pl, res=res, mode=1,
                    x_split=ops.act_split(x, 0.1), y_split=ys, y_split_slope=0.1)
    assert torch.equal(y0, y1)
    assert torch.equal(ys, ops.act_split(y0, 0.1))
    # planes only (no f32 store): y untouched
    y2 = torch.full_like(y0, 7.0)
    ys2 = torch.zeros_like(ys)
    ops.conv1d(x, w, cout, k, bias=b, dilation=d, pad_left=pl, res=res, mode=1, x_split=ops.act_split(x, 0.1),
               y_split=ys2, y_split_slope=0.1, no_y=True, out=y2)
    assert torch.equal(ys2, ys) and (y2 == 7.0).all()
    # residual rebuilt from split planes (hi + lo, leaky-relu undone): 2^-21 of the residual
    y3 = ops.conv1d(x, w, cout, k, bias=b, dilation=d, pad_left=pl, mode=1, x_split=ops.act_split(x, 0.1),
                    res_split=ops.act_split(res, 0.1), res_split_slope=0.1)
    assert (y3 - y0).abs().max() <= 2.0 ** -20 * res.abs().max()


def _e4m3(t, exp=0):
    return (t * 2.0 ** exp).clamp(-448, 448).to(torch.float32).to(torch.float8_e4m3fn)


def test_act_split_f8_planes():
    """SAT_SPLIT_F8: hi planes as in SAT_SPLIT_F16; unit 2 = e4m3(hi), unit 3 = e4m3(lo * 2^10), one byte per
    channel, OCP e4m3 round-to-nearest-even like torch.float8_e4m3fn"""
    ops, _ = _ops()
    x = _rand(2, 32, 301, seed=11, scale=3.0)
    s16 = ops.act_split(x.to(DEV), 0.1)
    s8 = ops.act_split(x.to(DEV), 0.1, fmt=1)
    assert torch.equal(s8[:, :, 0], s16[:, :, 0])
    hi = ops.unsplit(torch.stack([s16[:, :, 0], torch.zeros_like(s16[:, :, 0])], 2)).cpu()      # [B][C][T]
    lo = F.leaky_relu(x, 0.1) - hi
    raw = s8.cpu().view(torch.uint8).reshape(2, 2, 2, 2, 301, 16)
    got_h = raw[:, :, 1, 0].permute(0, 1, 3, 2).reshape(2, 32, 301)          # [B][chunk][T][16] -> [B][C][T]
    got_l = raw[:, :, 1, 1].permute(0, 1, 3, 2).reshape(2, 32, 301)
    assert torch.equal(got_h, _e4m3(hi).view(torch.uint8))
    assert torch.equal(got_l, _e4m3(lo, 10).view(torch.uint8))


@pytest.mark.parametrize("case", [(16, 16, 3, 1, 700), (32, 32, 7, 3, 600), (64, 64, 11, 1, 257), (128, 128, 3, 5, 260),
                                  (256, 256, 7, 5, 125), (256, 256, 11, 3, 250), (512, 256, 3, 1, 130)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv1d_f16f8_matches_its_decomposition(case):
    """SAT_CONV_F16F8 = hi*hi (f16 MFMA) + e4m3(W_lo 2^16) e4m3(x_hi) 2^-16 + e4m3(W_hi 2^6) e4m3(x_lo 2^10) 2^-16
    (block-scaled e4m3 MFMA), f32 accumulation.  Tight: against that decomposition evaluated in f64
    from the kernel's own input planes (pins operand order, scales and rounding).  Loose: against the
    exact product, where the e4m3 cross terms leave ~2^-15 per product"""
    ops, packing = _ops()
    cin, cout, k, d, T = case
    x = _rand(2, cin, T, seed=1)
    w = _rand(cout, cin, k, seed=2, scale=1.0 / np.sqrt(cin * k))
    b = _rand(cout, seed=3)
    res = _rand(2, cout, T, seed=4)
    pl = (k * d - d) // 2
    xs = ops.act_split(x.to(DEV), 0.1, fmt=1)
    ys = ops.split_like(2, cout, T, DEV)
    y = ops.conv1d(x.to(DEV), packing.pack_conv_weight_f16f8(w.to(DEV)), cout, k, bias=b.to(DEV), dilation=d, pad_left=pl,
                   res=res.to(DEV), mode=2, x_split=xs, y_split=ys, y_split_slope=0.1).cpu().double()
    xa = F.leaky_relu(x, 0.1)
    xh = ops.unsplit(torch.stack([xs[:, :, 0], torch.zeros_like(xs[:, :, 0])], 2)).cpu()        # the kernel's hi (toward zero)
    xl = xa - xh
    wh = w.to(torch.float16).float()
    wl = w - wh
    conv = lambda a, ww: F.conv1d(a.double(), ww.double(), None, dilation=d, padding=pl)
    emu = (conv(xh, wh) + conv(_e4m3(xh).float(), _e4m3(wl, 16).float() / 2 ** 16)
           + conv(_e4m3(xl, 10).float() / 2 ** 10, _e4m3(wh, 6).float() / 2 ** 6) + b.double()[None, :, None] + res.double())
    ref = conv(xa, w) + b.double()[None, :, None] + res.double()
    e_emu, e_ref = (y - emu).abs().max().item(), (y - ref).abs().max().item()
    print(f"max abs: vs decomposition {e_emu:.2e}, vs exact {e_ref:.2e} (rms {rms(y - ref):.2e})")
    assert e_emu < 1e-5
    assert e_ref < 1e-4 and rms(y - ref) < 2e-5
    assert torch.equal(ys, ops.act_split(y.float().to(DEV), 0.1, fmt=1))        # output planes in the format it reads


@pytest.mark.parametrize("case", [(16, 3, 1, 1000), (16, 7, 3, 700), (16, 11, 5, 2500), (16, 11, 1, 223), (32, 3, 5, 449), (32, 7, 1, 224), (32, 11, 5, 1500)],
                         ids=lambda c: "x".join(map(str, c)))
def test_fused_pair_split_planes(case):
    ops, packing = _ops()
    C, k, d, T = case
    x = _rand(2, C, T, seed=1).to(DEV)
    pk = packing.pack_conv_weight_f16x3
    w1, w2 = pk(_rand(C, C, k, seed=2, scale=0.6 / np.sqrt(C * k)).to(DEV)), pk(_rand(C, C, k, seed=3, scale=0.6 / np.sqrt(C * k)).to(DEV))
    b1, b2 = _rand(C, seed=4, scale=0.1).to(DEV), _rand(C, seed=5, scale=0.1).to(DEV)
    y0 = ops.resblock_pair(x, w1, b1, w2, b2, k, d)
    ys = ops.split_like(2, C, T, DEV)
    y1 = ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=ops.act_split(x, 0.1), y_split=ys, y_split_slope=0.1)
    assert torch.equal(y0, y1)
    assert torch.equal(ys, ops.act_split(y0, 0.1))
    ys2 = torch.zeros_like(ys)
    y2 = torch.full_like(y0, 7.0)
    ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=ops.act_split(x, 0.1), y_split=ys2, y_split_slope=0.1,
                      planes_residual=True, no_y=True, out=y2)
    assert (y2 == 7.0).all()
    assert (ops.unsplit(ys2) - ops.unsplit(ys)).abs().max() <= 2.0 ** -19 * x.abs().max()
    # planes end to end with the MRF sum: out = (acc + pair(x)) / 3 in f32, plus planes of it (C = 16: the
    # 16x16x32-MFMA kernel)
    acc0 = _rand(2, C, T, seed=6).to(DEV)
    out = acc0.clone()
    ys3 = torch.zeros_like(ys)
    ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=ops.act_split(x, 0.1), y_split=ys3, y_split_slope=0.1,
                      planes_residual=True, out=out, accum=True, accum_div=3.0)
    ref = (acc0 + y0) / 3
    assert (out - ref).abs().max() <= 2e-6
    assert (ops.unsplit(ys3) - F.leaky_relu(ref, 0.1)).abs().max() <= 2e-6


def test_generator_split_plane_pipeline_equals_f32_handover(model_f16x3, gold):
    """the generator with split-plane activations between layers vs the same kernels fed with f32
    activations: same arithmetic, different staging"""
    from satools_amd._lib import lib, check
    fx = gold.npz("fx_gen.npz")
    model = model_f16x3
    g = model.hifigan
    if g.precision not in ("f16x3", "f16f8r"):
        pytest.skip("split planes belong to the split-f16 generator")
    x = torch.randn(2, g.imput_dim, 25, generator=torch.Generator().manual_seed(3)).to(DEV)
    with conv_option("convring", 33, 1):
        y_ring = g(x)[0].clone()             # the thick stages on the LDS-DMA ring conv (16x16x32 MFMA shape; + 32: also for this small batch)
    keep_ups = g.ups_ring
    try:
        # (the f32-handover pipeline reads the upsamplers' rows in the (channel, phase) order: ups_ring off for the comparison)
        g.ups_ring = 0
        with conv_option("convring", 0, 1):      # the register-staged tiles below accumulate in ONE order whatever the staging
            _split_plane_pipeline_equals_f32_handover(g, x, lib, check)
            y_lean = g(x)[0].clone()
    finally:
        g.ups_ring = keep_ups
    # K = 32 per instruction associates differently: agreement to f32 rounding of the accumulation
    assert rms((y_ring - y_lean).cpu().numpy()) < 5e-7


def _split_plane_pipeline_equals_f32_handover(g, x, lib, check):
    y2 = g(x)[0].clone()                     # default: planes only, residuals rebuilt from hi + lo (22 bits)
    check(lib().sat_hifigan_set_option(g._handle, b"branch_streams", 0), "set_option")
    y2s = g(x)[0].clone()                    # the three resblock branches of a stage on one stream
    check(lib().sat_hifigan_set_option(g._handle, b"branch_streams", 5), "set_option")
    assert torch.equal(y2, g(x)[0])
    check(lib().sat_hifigan_set_option(g._handle, b"branch_streams", g.branch_streams), "set_option")
    assert torch.equal(y2, y2s)
    check(lib().sat_hifigan_set_option(g._handle, b"planes_residual", 0), "set_option")
    try:
        y1 = g(x)[0].clone()                 # planes + f32 copies for the residuals
        check(lib().sat_hifigan_set_option(g._handle, b"split_acts", 0), "set_option")
        y0 = g(x)[0].clone()                 # f32 activations, split inside every kernel
    finally:
        check(lib().sat_hifigan_set_option(g._handle, b"split_acts", 1), "set_option")
        check(lib().sat_hifigan_set_option(g._handle, b"planes_residual", 1), "set_option")
    assert torch.equal(y0, y1)
    print("planes-only residuals vs f32 residuals: rms", rms((y2 - y1).cpu().numpy()))
    assert rms((y2 - y1).cpu().numpy()) < 5e-7


# ---------------------------------------------------------------------------------------------
# generator and end to end
# ---------------------------------------------------------------------------------------------
def test_generator_matches_golden_teacher_forced(model, gold):
    """generator alone on the reference's own BN/F0/speaker input (n = 8000): 1e-4 RMS bar of the
    north star, measured here against the reference's waveform"""
    fx = gold.npz("fx_gen.npz")
    spk = F.one_hot(torch.from_numpy(fx["spk_argmax"]), len(model.spk))
    f0 = torch.from_numpy(fx["f0_raw"].copy())
    y = model._forward(f0, torch.from_numpy(fx["bn"]), spk)
    assert np.allclose(f0.numpy(), fx["f0_after"], atol=2e-6)      # normalised in place like the reference
    assert y.shape == fx["y"].shape
    err = rms(y.cpu().numpy() - fx["y"])
    print("generator RMS error vs reference:", err, "signal RMS", rms(fx["y"]))
    assert err < 1e-5


def test_generator_f16f8_precision(model, gold):
    """opt-in SAT_CONV_F16F8 generator (cross terms on the e4m3 MFMA): waveform within 2e-5 RMS of the
    reference's on its teacher-forced input (measured ~3e-6; the path's bar is 1e-4)"""
    fx = gold.npz("fx_gen.npz")
    g = model.hifigan
    old = g.precision
    g.precision = "f16f8"
    g.invalidate()
    try:
        spk = F.one_hot(torch.from_numpy(fx["spk_argmax"]), len(model.spk))
        y = model._forward(torch.from_numpy(fx["f0_raw"].copy()), torch.from_numpy(fx["bn"]), spk)
    finally:
        g.precision = old
        g.invalidate()
    err = rms(y.cpu().numpy() - fx["y"])
    print("f16f8 generator RMS error vs reference:", err)
    assert err < 2e-5


def test_convert_matches_golden(model, gold):
    """model.convert on 5 s utterances with the reference's F0 track handed over through set_f0
    (the anonymize pipeline's path, bin/pipeline.py:107-149); YAAPT parity is tested separately"""
    from satools_amd import synthetic
    fx, f0fx = gold.npz("fx_e2e.npz"), gold.npz("fx_f0.npz")
    wav = synthetic.harm_batch([0], 80000)
    keep = wav.clone()
    model.set_f0(torch.from_numpy(f0fx["harm0_80000"].copy()))
    y = model.convert(wav.to(DEV), target=model.spk[3])
    assert y.shape == (1, 80001) and y.dtype == torch.float32       # B == 1 -> [1, n'] (squeeze(0))
    assert torch.equal(wav, keep)
    e1 = rms(y.cpu().numpy() - fx["harm0_80000_str"])
    wav = synthetic.harm_batch([0, 1], 80000)
    model.set_f0(torch.from_numpy(f0fx["harm01_80000_batch"].copy()).to(DEV))
    y = model.convert(wav.to(DEV), target=[model.spk[3], model.spk[10]])
    assert y.shape == (2, 1, 80001)                                  # B > 1 -> [B, 1, n']
    e2 = rms(y.cpu().numpy() - fx["harm01_80000_list"])
    print("convert RMS error vs reference: B=1", e1, "B=2", e2, "signal RMS", rms(fx["harm0_80000_str"]))
    assert e1 < 1e-4 and e2 < 1e-4
    assert model.f0 is None                                          # set_f0 is consumed once


@pytest.mark.parametrize("shape", [(1, 4800), (3, 16123), (2, 31999), (5, 9600)], ids=lambda s: f"B{s[0]}xn{s[1]}")
def test_convert_ragged_sizes_match_oracle(model, fbank_tag_state, shape):
    """sizes off the 5 s / batch-of-32 grid (odd sample counts, sub-second utterances, odd batches): the whole
    convert() with YAAPT on path against the CPU oracle run on the same input"""
    from oracle import convert as oconv
    from oracle import yaapt as oyaapt
    from satools_amd import synthetic
    B, n = shape
    state, _ = fbank_tag_state
    wav = synthetic.harm_batch(list(range(B)), n)
    tg = synthetic.targets(model.spk, list(range(B)))
    nt = torch.get_num_threads()
    torch.set_num_threads(1)          # the reference's own YAAPT setting (frame 0 is thread-count dependent in torch)
    try:
        f0 = oyaapt.yaapt(wav, {"frame_length": 35.0, "frame_space": 20.0, "nccf_thresh1": 0.25, "tda_frame_length": 25.0})
    finally:
        torch.set_num_threads(nt)
    ref = oconv.convert_fbank(state["base_model_state_dict"], model.spk, wav, tg if B > 1 else tg[0], f0)
    y = model.convert(wav.to(DEV), target=tg if B > 1 else tg[0])
    assert y.shape == ref.shape
    err = rms(y.cpu().numpy() - ref.numpy())
    print(f"B={B} n={n}: out {tuple(y.shape)}, RMS error vs oracle {err:.2e}")
    assert err < 1e-5


def test_convert_errors_like_the_reference(model):
    from satools_amd import synthetic
    wav = synthetic.harm_batch([0], 8000).to(DEV)
    model.set_f0(torch.zeros(1, 25))
    with pytest.raises(ValueError):
        model.convert(wav, target="no-such-speaker")
    model.set_f0(torch.zeros(1, 25))
    with pytest.raises(AssertionError):
        model.convert(wav, target=[model.spk[0], model.spk[1]])     # len(target) != len(input_wav)
    assert model.eval() is None                                      # reference quirk: train() returns None


# ---- the ASR half of the fbank-tag net ------------------------------------------------------
@pytest.mark.parametrize("t,d", [(4, 64), (5, 64), (6, 128), (7, 64), (58, 1024)])
def test_tdnnf_unfold15_matches_reference_unfold(t, d):
    """windows / bypass of a subsampling-1.5 TDNNF layer against the reference's formulation
    (chain/nn.py:267-304): F.unfold of the flattened [T*D] input with step int(1.5*D); add_padd"""
    ops, _ = _ops()
    x = torch.randn(2, t, d, generator=torch.Generator().manual_seed(t))
    win, byp = ops.tdnnf_unfold15(x.permute(0, 2, 1).contiguous().cuda())
    want = F.unfold(x.reshape(2, 1, t * d, 1), (d, 1), stride=(int(d * 1.5), 1)).permute(0, 2, 1)     # [B, T', D]
    assert torch.equal(win.cpu().permute(0, 2, 1), want)
    idx = torch.arange(0, 1.6e6, 1.5).long()[:int(t / 1.5)]
    bw = torch.zeros_like(want)
    bw[:, :len(idx)] = x[:, idx]
    assert torch.equal(byp.cpu().permute(0, 2, 1), bw)


def test_log_softmax_channels():
    ops, _ = _ops()
    x = torch.randn(3, 3280, 37, generator=torch.Generator().manual_seed(0)) * 4
    got = ops.log_softmax_channels_(x.cuda().clone()).cpu()
    assert (got - torch.log_softmax(x, dim=1)).abs().max() < 2e-5


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("name,ids,n", [("harm0_16000", [0], 16000), ("harm01_32000", [0, 1], 32000)])
def test_asr_forward_matches_reference(fbank_tag_state, name, ids, n, precision):
    """Net.forward (tdnnf_vq.py:259-284) on the HIP path against outputs of the reference itself"""
    import os
    from conftest import GOLD
    from satools_amd import synthetic
    fx = np.load(os.path.join(GOLD, "fx_asr.npz"))
    state, model = fbank_tag_state
    model.load_state_dict(state["base_model_state_dict"])
    bx = model.bn_extractor.cuda()
    old = bx.precision
    bx.precision = precision
    try:
        wav = synthetic.harm_batch(ids, n)
        chain, xent = bx(wav.clone().cuda())
    finally:
        bx.precision = old
    assert chain.shape == xent.shape and chain.shape[2] == 3280
    for got, key in ((chain, "chain_sub"), (xent, "xent_sub")):
        want = fx[f"{name}/{key}"]
        assert got.shape[:2] == want.shape[:2]
        assert rms(got.cpu().numpy()[..., ::8] - want) <= 1e-4 * max(1.0, rms(want)), key
    assert np.allclose(torch.logsumexp(xent, dim=2).cpu().numpy(), fx[f"{name}/xent_lse"], atol=1e-4)


# ---- BASELINE.json's full size (configs[1]: 32 utterances x 5 s) through size-independent properties ----------
def test_full_size_batch_properties(model, fbank_tag_state):
    """32 x 5 s: (1) the generator treats utterances independently — any slice of the batch gives the same bits as
    the full batch; (2) so does YAAPT; (3) one utterance of the full batch against the CPU oracle's generator
    (teacher-forced on the HIP path's own generator input), 1e-4 RMS bar; (4) shape, range, determinism."""
    from oracle import convert as oconv
    from oracle import hifigan as ohg
    from satools_amd import ops, synthetic
    seeds = list(range(32))
    wav = synthetic.harm_batch(seeds).to(DEV)
    targets = synthetic.targets(model.spk, seeds)
    y = model.convert(wav, target=targets)
    assert y.shape == (32, 1, 80001) and bool(torch.isfinite(y).all()) and float(y.abs().max()) <= 1.0
    assert torch.equal(y, model.convert(wav, target=targets))
    # generator input of the whole batch, as convert() builds it
    f0 = model.get_f0(wav)
    assert f0.shape == (32, 250)
    for i in (0, 13, 31):
        assert torch.equal(model.get_f0(wav[i:i + 1])[0], f0[i])
    bn = model.get_bn(wav)
    spk = model.get_spk_id(wav, targets).to(DEV, torch.float32).contiguous()
    f0n = f0.to(DEV).clone()
    ops.f0_norm_transform_(f0n)
    x = ops.assemble_input(bn, f0n, spk, spk.shape[1])
    assert x.shape == (32, 504, 250)
    full = model.hifigan(x)[0].clone()
    assert torch.equal(full.reshape(y.shape), y)
    # a slice of the batch: the kernels treat utterances independently, so with ONE dispatch the bits are those of the full batch.
    # By default a batch of a few utterances leaves the thick stages on the register-staged tiles (too few tiles for the one-
    # block-per-CU ring conv, csrc/conv_ring16.hip): another accumulation order, agreement to f32 rounding
    with conv_option("convring", 33, 1):                 # (+ 32: the ring conv whatever the number of tiles)
        full_ring = model.hifigan(x)[0].clone()
        assert torch.equal(full_ring, full)
        for sl in (slice(0, 1), slice(5, 9), slice(29, 32)):
            assert torch.equal(model.hifigan(x[sl].contiguous())[0], full[sl])
    # (round 5: with the default "f16f8r" generator a small batch is also another ARITHMETIC on the thick stages — the f16x3 tiles instead
    # of the ring kernel's e4m3 cross terms, 7e-7 RMS apart; both within 1e-5 of the oracle)
    for sl in (slice(0, 1), slice(5, 9)):
        assert rms((model.hifigan(x[sl].contiguous())[0] - full[sl]).cpu().numpy()) < (5e-7 if model.hifigan.precision == "f16x3" else 3e-6)
    _, gen_sd = oconv.split_state_dict(fbank_tag_state[0]["base_model_state_dict"])
    ref = ohg.generator(gen_sd, x[17:18].cpu())
    err = rms((full[17:18].cpu() - ref).numpy())
    print("full-size batch, utterance 17 vs oracle generator: rms", err)
    assert err < 1e-5


def test_converts_in_flight_on_separate_streams_equal_serial(model):
    """the benchmark's mode (bench.py run_steps, the reference's jobs_per_compute_device): eight convert() batches issued
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


def test_weight_caches_follow_in_place_updates():
    """the packed-weight caches are keyed on (data_ptr, _version) of parameters and buffers looked up through a
    module walk done once: an in-place update of a parameter or of a BatchNorm buffer (which `.to()` had replaced
    by a new tensor object) must still invalidate them"""
    from satools_amd import synthetic
    m = _model()
    wav = synthetic.harm_batch([0], 16000).to(DEV)
    bn0 = m.get_bn(wav).clone()
    bx = m.bn_extractor
    with torch.no_grad():
        saved = bx.tdnn1.bn.running_mean.clone()
        bx.tdnn1.bn.running_mean.add_(0.25)
    bn1 = m.get_bn(wav).clone()
    assert not torch.equal(bn0, bn1)
    with torch.no_grad():
        bx.tdnn1.bn.running_mean.copy_(saved)
        assert torch.equal(m.get_bn(wav), bn0)
        x = torch.randn(1, m.hifigan.imput_dim, 50, device=DEV)
        y0 = m.hifigan(x)[0].clone()
        p = next(m.hifigan.parameters())
        p.mul_(1.5)
        y1 = m.hifigan(x)[0].clone()
        p.div_(1.5)
    assert not torch.equal(y0, y1)


def test_c_abi_reports_errors_as_status_codes():
    """the C ABI never throws: a too-small workspace comes back as SAT_ERR_WORKSPACE (-3), bad arguments / unsupported
    shapes as SAT_ERR_INVALID (-1), each with a message behind sat_last_error(); the Python layer turns them into
    SatError"""
    import ctypes as C
    from satools_amd import _lib, f0 as f0_hip, synthetic
    from satools_amd._lib import lib, ptr, stream
    l = lib()
    # YAAPT with a workspace one byte short
    wav = synthetic.harm_batch([0], 16000).to(DEV)
    P = f0_hip.make_plan(16000, {"frame_length": 35.0, "frame_space": 20.0, "nccf_thresh1": 0.25, "tda_frame_length": 25.0})
    hann, kaiser, tw = f0_hip._get_tables(P, wav.device)
    need = l.sat_yaapt_workspace_bytes(C.byref(P), 1)
    ws = torch.empty(need // 4, dtype=torch.float32, device=DEV)
    f0 = torch.empty(1, P.nframes, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    rc = l.sat_yaapt_f32(C.byref(P), ptr(wav), ptr(f0), ptr(status), ptr(hann), ptr(kaiser), ptr(tw), ptr(ws), need - 1, 1, stream())
    assert rc == -3, rc
    assert b"workspace" in l.sat_last_error()
    assert l.sat_yaapt_f32(C.byref(P), ptr(wav), ptr(f0), ptr(status), ptr(hann), ptr(kaiser), ptr(tw), ptr(ws), need, 1, stream()) == 0
    # generator with a workspace that is too small
    m = _model()
    x = torch.randn(1, m.hifigan.imput_dim, 20, device=DEV)
    m.hifigan(x)
    g = m.hifigan
    need = l.sat_hifigan_workspace_bytes(g._handle, 1, 20)
    small = torch.empty(16, dtype=torch.float32, device=DEV)
    y = torch.empty(1, 1, 20 * 320 + 1, device=DEV)
    rc = l.sat_hifigan_forward_f32(g._handle, ptr(x), ptr(y), ptr(small), 64, 1, 20, stream())
    assert rc == -3, rc
    assert need > 64 and l.sat_last_error()
    # null pointers / unsupported values -> SAT_ERR_INVALID
    assert l.sat_f0_stats_f32(None, 10, None, stream()) == -1
    assert l.sat_f0_mean_reversion_f32(ptr(f0), ptr(f0), P.nframes, C.c_float(0.5), 32, stream()) == -1      # out == in
    assert l.sat_conv_set_option(b"no_such_option", 1) == -1 and b"no_such_option" in l.sat_last_error()
    # and through the Python layer
    with pytest.raises(_lib.SatError):
        _lib.check(l.sat_conv_set_option(b"no_such_option", 1), "sat_conv_set_option")


@pytest.mark.parametrize("T,dil", [(129, 3), (200, 5), (256, 1)])
def test_half_width_tiles_of_the_3_tap_planes_conv(T, dil):
    """the 64-row x 128-column tile of the 3-tap planes conv is only dispatched for launches of fewer than 256 blocks
    (rows 128, 129..256 frames, few utterances): pinned here with dilation, residual from planes and MRF-style
    accumulation, against the same conv on the exact-f32 kernel and against torch in float64"""
    ops, packing = _ops()
    B, C = 2, 128
    x, w, b = _rand(B, C, T, seed=1), _rand(C, C, 3, seed=2, scale=0.05), _rand(C, seed=3, scale=0.1)
    r, acc0 = _rand(B, C, T, seed=4), _rand(B, C, T, seed=5)
    xd, rd = x.to(DEV), r.to(DEV)
    xs = ops.act_split(xd, 0.1)
    rs = ops.act_split(rd, 0.1)
    ref = F.conv1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), dilation=dil, padding=dil) + r.double()
    ref = (acc0.double() + ref) / 3.0
    out = acc0.clone().to(DEV)
    y = ops.conv1d(xd, packing.pack_conv_weight_f16x3(w.to(DEV)), C, 3, bias=b.to(DEV), dilation=dil, pad_left=dil, mode=1,
                   x_split=xs, res_split=rs, res_split_slope=0.1, out=out, accum=True, accum_div=3.0)
    err = (y.cpu().double() - ref).abs().max().item()
    assert err < 3e-5, err
    # the same launch with the f32 residual: the planes residual reproduces it to 2^-19
    out2 = acc0.clone().to(DEV)
    y2 = ops.conv1d(xd, packing.pack_conv_weight_f16x3(w.to(DEV)), C, 3, bias=b.to(DEV), dilation=dil, pad_left=dil, mode=1,
                    x_split=xs, res=rd, out=out2, accum=True, accum_div=3.0)
    assert (y2 - y).abs().max().item() <= 2.0 ** -19 * float(r.abs().max())


@pytest.mark.parametrize("cin,cout,T", [(128, 1024, 249), (1024, 1024, 250), (512, 3280, 130), (1024, 256, 500), (64, 1024, 700)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("variant", ["plain", "res", "gelu_planes", "bn_relu_postres"])
def test_1x1_gemm_kernels_on_split_planes(cin, cout, T, variant):
    """the three kernels behind a 1x1 conv on split planes (sat_conv_set_option "k1_gemm": 1 = 128 x 128 register-staged,
    2 = LDS-DMA ring on the 32x32x16 MFMA shape, 3 = the ring on the 16x16x32 shape, the default): 1 and 2 accumulate in
    the same order and agree bit for bit; 3 sums K = 32 inside one instruction and agrees with them to f32 rounding of
    the accumulation; all three against torch in float64 at the split-f16 tolerance.  Ragged time tiles (T not a
    multiple of 256), rows not a multiple of 128, every epilogue the 16x16 kernel carries."""
    ops, packing = _ops()
    from satools_amd import _lib
    B = 3
    x, w, b = _rand(B, cin, T, seed=1), _rand(cout, cin, 1, seed=2, scale=cin ** -0.5), _rand(cout, seed=3)
    r, sc, sh = _rand(B, cout, T, seed=4), torch.rand(cout, generator=torch.Generator().manual_seed(5)) + 0.5, _rand(cout, seed=6)
    xd, rd = x.to(DEV), r.to(DEV)
    xs = ops.act_split(xd, 1.0)
    wp = packing.pack_conv_weight_f16x3(w.to(DEV))
    lin = torch.einsum("oc,bct->bot", w[:, :, 0].double(), x.double()) + b.double()[None, :, None]
    if variant == "plain":
        ref, kw = lin, {}
    elif variant == "res":
        ref, kw = lin + r.double(), dict(res=rd)
    elif variant == "gelu_planes":
        ref, kw = F.gelu(lin), dict(gelu=True)
    else:
        ref, kw = r.double() + F.relu(lin * sc.double()[None, :, None] + sh.double()[None, :, None]), dict(
            post_res=rd, ch_scale=sc.to(DEV), ch_shift=sh.to(DEV), relu=True)
    out = {}
    try:
        for opt in (1, 2, 3):
            _lib.check(_lib.lib().sat_conv_set_option(b"k1_gemm", opt), "sat_conv_set_option")
            ys = ops.split_like(B, cout, T, xd.device) if variant == "gelu_planes" else None
            y = ops.conv1d(xd, wp, cout, 1, bias=b.to(DEV), mode=1, x_split=xs, y_split=ys, **kw)
            out[opt] = (y, ys)
    finally:
        _lib.check(_lib.lib().sat_conv_set_option(b"k1_gemm", 3), "sat_conv_set_option")
    scale = float(ref.abs().max())
    for opt, (y, ys) in out.items():
        assert (y.cpu().double() - ref).abs().max().item() < 1e-5 * scale, (opt, variant)
    assert torch.equal(out[1][0], out[2][0])
    assert (out[3][0] - out[1][0]).abs().max().item() < 2e-6 * scale
    if variant == "gelu_planes":
        # the planes written next to y are split(y): hi + lo reproduces y to 2^-21 (read back through a 1x1 identity product)
        assert torch.equal(out[1][1], out[2][1])
        eye = torch.eye(cout)[:128].reshape(128, cout, 1).contiguous().to(DEV)
        y3, ys3 = out[3]
        back = ops.conv1d(y3, packing.pack_conv_weight_f16x3(eye), 128, 1, mode=1, x_split=ys3)
        assert (back - y3[:, :128]).abs().max().item() <= 2.0 ** -20 * float(y3.abs().max())


@pytest.mark.parametrize("C,T", [(64, 701), (512, 1000), (128, 256)], ids=lambda v: str(v))
def test_stride2_three_tap_conv_as_one_wrapped_product(C, T):
    """sat_conv1d_desc.x_wrap_channels: a stride-2 3-tap conv over [even | odd] phase-split planes as ONE 1x1 product
    over 3 C input channels, the last C of them (tap 2) reading the even phase one frame later — against
    torch.nn.functional.conv1d(stride=2) in float64 and against the two-tap polyphase form (one zero tap) the other
    kernels run; odd and even input lengths (the last even frame exists / is the zero pad)."""
    ops, packing = _ops()
    from satools_amd import _lib
    from satools_amd.wav2vec2 import _polyphase_stride2_weight
    B, cout = 2, 128
    x, w, b = _rand(B, C, T, seed=1), _rand(cout, C, 3, seed=2, scale=(3 * C) ** -0.5), _rand(cout, seed=3)
    ref = F.conv1d(x.double(), w.double(), b.double(), stride=2)
    Tq = ref.shape[2]
    Th = (T + 1) // 2
    ph = torch.zeros(B, 2 * C, Th)
    ph[:, :C, :] = x[:, :, 0::2]
    ph[:, C:, :T // 2] = x[:, :, 1::2]
    phd = ph.to(DEV)
    xs = ops.act_split(phd, 1.0)
    ww = packing.pack_conv_weight_f16x3(torch.cat([w[:, :, 0], w[:, :, 1], w[:, :, 2]], 1).unsqueeze(-1).contiguous().to(DEV))
    y = ops.conv1d(phd, ww, cout, 1, bias=b.to(DEV), t_out=Tq, mode=1, x_split=xs, x_wrap_channels=2 * C, c_in=3 * C)
    assert y.shape == ref.shape
    scale = float(ref.abs().max())
    assert (y.cpu().double() - ref).abs().max().item() < 1e-5 * scale
    wc, kp = _polyphase_stride2_weight(w.to(DEV))
    y2 = ops.conv1d(phd, packing.pack_conv_weight_f16x3(wc), cout, kp, bias=b.to(DEV), pad_left=0, pad_right=0, t_out=Tq, mode=1, x_split=xs)
    assert (y - y2).abs().max().item() < 2e-6 * scale
    # refused where the GEMM path cannot serve it: rows not a multiple of 128
    w96 = packing.pack_conv_weight_f16x3(torch.cat([w[:96, :, 0], w[:96, :, 1], w[:96, :, 2]], 1).unsqueeze(-1).contiguous().to(DEV))
    with pytest.raises(_lib.SatError):
        ops.conv1d(phd, w96, 96, 1, bias=b[:96].to(DEV), t_out=Tq, mode=1, x_split=xs, x_wrap_channels=2 * C, c_in=3 * C)


@pytest.mark.parametrize("C,T,dil,k", [(64, 700, 1, 3), (128, 513, 5, 3), (256, 250, 3, 3), (64, 700, 5, 7), (256, 250, 1, 7), (64, 700, 5, 11), (256, 250, 1, 11), (128, 513, 3, 11)], ids=lambda v: str(v))
def test_three_blocks_per_cu_form_of_the_conv_tile_gives_the_same_bits(C, T, dil, k):
    """sat_conv_set_option("lean3" / "lean7" / "lean11"): the 3-, 7- and 11-tap convs on split planes through the two-block form
    (double-buffered fragments, residual prefetch) and through the three-blocks-per-CU form (the default): same
    accumulation order, same epilogue arithmetic — same bits, with the generator's conv2 epilogue (residual from
    planes, MRF accumulate / 3, f32 + planes out)"""
    ops, packing = _ops()
    from satools_amd import _lib
    B = 3
    x, w, b = _rand(B, C, T, seed=1).to(DEV), _rand(C, C, k, seed=2, scale=(k * C) ** -0.5).to(DEV), _rand(C, seed=3).to(DEV)
    r, acc0 = _rand(B, C, T, seed=4).to(DEV), _rand(B, C, T, seed=5).to(DEV)
    xs, rs = ops.act_split(x, 0.1), ops.act_split(r, 0.1)
    wp = packing.pack_conv_weight_f16x3(w)
    out = {}
    try:
        _lib.check(_lib.lib().sat_conv_set_option(b"convring", 0), "sat_conv_set_option")     # (else C >= 128 goes to conv_ring16.hip)
        for v in (0, 1):
            _lib.check(_lib.lib().sat_conv_set_option(b"lean%d" % k, v), "sat_conv_set_option")
            ys = ops.split_like(B, C, T, DEV)
            y = ops.conv1d(x, wp, C, k, bias=b, dilation=dil, pad_left=dil * (k - 1) // 2, mode=1, x_split=xs, res_split=rs, res_split_slope=0.1,
                           out=acc0.clone(), accum=True, accum_div=3.0, y_split=ys, y_split_slope=0.1)
            out[v] = (y, ys)
    finally:
        _lib.check(_lib.lib().sat_conv_set_option(b"lean%d" % k, 1), "sat_conv_set_option")
        _lib.check(_lib.lib().sat_conv_set_option(b"convring", 1), "sat_conv_set_option")
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    ref = (acc0.double().cpu() + F.conv1d(F.leaky_relu(x.double().cpu(), 0.1), w.double().cpu(), b.double().cpu(), dilation=dil, padding=dil * (k - 1) // 2)
           + r.double().cpu()) / 3.0
    assert (out[1][0].cpu().double() - ref).abs().max().item() < 3e-5


@pytest.mark.parametrize("T,dil,k", [(700, 1, 3), (1000, 3, 3), (126, 5, 3), (127, 1, 3), (379, 5, 3), (700, 1, 7), (122, 5, 7), (123, 3, 7),
                                     (1000, 5, 11), (118, 1, 11), (119, 3, 11)], ids=lambda v: str(v))
def test_fused_resblock_step_c64_equals_the_two_launch_path(T, dil, k):
    """resblock_pair64_kernel (C = 64, 3 / 7 / 11 taps, planes end to end: the generator's third stage) against the two conv
    launches it replaces — conv1 (dilation d) writing split(lrelu(t1)) planes, conv2 (dilation 1) adding the residual
    rebuilt from the input planes — bit for bit, planes-only and with the MRF accumulation into an f32 tensor; tile
    edges (128 - (k - 1) output positions per block), utterance edges and every dilation of the generator"""
    ops, packing = _ops()
    C, B = 64, 2
    x = _rand(B, C, T, seed=1).to(DEV)
    pk = packing.pack_conv_weight_f16x3
    w1, w2 = pk(_rand(C, C, k, seed=2, scale=0.6 / np.sqrt(C * k)).to(DEV)), pk(_rand(C, C, k, seed=3, scale=0.6 / np.sqrt(C * k)).to(DEV))
    b1, b2 = _rand(C, seed=4, scale=0.1).to(DEV), _rand(C, seed=5, scale=0.1).to(DEV)
    xs = ops.act_split(x, 0.1)
    acc0 = _rand(B, C, T, seed=6).to(DEV)

    def two_launches(accum):
        t1s, ys = ops.split_like(B, C, T, DEV), ops.split_like(B, C, T, DEV)
        ops.conv1d(x, w1, C, k, bias=b1, dilation=dil, pad_left=dil * (k - 1) // 2, mode=1, x_split=xs, y_split=t1s, y_split_slope=0.1, no_y=True)
        if accum:
            y = ops.conv1d(x, w2, C, k, bias=b2, pad_left=(k - 1) // 2, mode=1, x_split=t1s, res_split=xs, res_split_slope=0.1,
                           y_split=ys, y_split_slope=0.1, out=acc0.clone(), accum=True, accum_div=3.0)
            return y, ys
        ops.conv1d(x, w2, C, k, bias=b2, pad_left=(k - 1) // 2, mode=1, x_split=t1s, res_split=xs, res_split_slope=0.1,
                   y_split=ys, y_split_slope=0.1, no_y=True)
        return None, ys

    def fused(accum):
        ys = ops.split_like(B, C, T, DEV)
        if accum:
            y = ops.resblock_pair(x, w1, b1, w2, b2, k, dil, x_split=xs, y_split=ys, y_split_slope=0.1, planes_residual=True,
                                  out=acc0.clone(), accum=True, accum_div=3.0)
            return y, ys
        y = torch.full((B, C, T), 7.0, device=DEV)
        ops.resblock_pair(x, w1, b1, w2, b2, k, dil, x_split=xs, y_split=ys, y_split_slope=0.1, planes_residual=True, no_y=True, out=y)
        assert (y == 7.0).all()
        return None, ys

    for accum in (False, True):
        ya, sa = two_launches(accum)
        yb, sb = fused(accum)
        assert torch.equal(sa, sb), accum
        if accum:
            assert torch.equal(ya, yb)
    # and against torch in float64 (the planes residual carries 22 bits of x)
    ref = F.conv1d(F.leaky_relu(F.conv1d(F.leaky_relu(x.double().cpu(), 0.1), _rand(C, C, k, seed=2, scale=0.6 / np.sqrt(C * k)).double(),
                                         b1.double().cpu(), dilation=dil, padding=dil * (k - 1) // 2), 0.1),
                   _rand(C, C, k, seed=3, scale=0.6 / np.sqrt(C * k)).double(), b2.double().cpu(), padding=(k - 1) // 2) + x.double().cpu()
    y, _ = fused(True)
    assert (y.cpu().double() - (acc0.double().cpu() + ref) / 3).abs().max().item() < 1e-5


def _mrf_weights(C, ks, seed):
    """seeded weights of a ResBlock1 branch: three steps of (w1, b1, w2, b2), f32 on the CPU"""
    steps = []
    for i in range(3):
        s = seed + 10 * i
        steps.append((_rand(C, C, ks, seed=s, scale=0.7 / np.sqrt(C * ks)), _rand(C, seed=s + 1, scale=0.1),
                      _rand(C, C, ks, seed=s + 2, scale=0.7 / np.sqrt(C * ks)), _rand(C, seed=s + 3, scale=0.1)))
    return steps


def _mrf_launch_by_launch(ops, pk, x, xs, branches, out_div, y_split):
    """today's path: per branch three sat_resblock_pair_f16x3 launches with planes end to end, the last one accumulating
    into the f32 MRF sum (hifigan.hip's loop)"""
    B, C, T = x.shape
    acc = torch.empty(B, C, T, device=DEV)
    nb = len(branches)
    for j, (k, steps) in enumerate(branches):
        cur = xs
        for i, (w1, b1, w2, b2) in enumerate(steps):
            a = (pk(w1.to(DEV)), b1.to(DEV), pk(w2.to(DEV)), b2.to(DEV))
            if i < 2:
                nxt = ops.split_like(B, C, T, DEV)
                ops.resblock_pair(x, *a, k, 2 * i + 1, x_split=cur, y_split=nxt, y_split_slope=0.1, planes_residual=True, no_y=True, out=acc)
                cur = nxt
            else:
                last = j == nb - 1
                ops.resblock_pair(x, *a, k, 2 * i + 1, x_split=cur, y_split=y_split if last else None, y_split_slope=0.1,
                                  planes_residual=True, out=acc, accum=j > 0, accum_div=out_div if last else 0.0)
    return acc


@pytest.mark.parametrize("T,B", [(700, 2), (512, 1), (513, 2), (1600, 3), (60, 2), (11, 1), (4099, 2)], ids=lambda v: str(v))
def test_fused_mrf_block_c16_equals_the_nine_launch_path(T, B):
    """mrf16_kernel (the whole MRF block of the C = 16 stage in one launch: 3 branches x 3 steps x 2 convs + the mean,
    hifigan/archi.py:82-86) against the nine fused-step launches it replaces: bit for bit on the f32 mean and on its
    split planes; tile edges (512 outputs per tile), utterances shorter than the 60-position halo, ragged tails"""
    ops, packing = _ops()
    C = 16
    pk = packing.pack_conv_weight_f16x3
    x = _rand(B, C, T, seed=1).to(DEV)
    xs = ops.act_split(x, 0.1)
    branches = [(3, _mrf_weights(C, 3, 100)), (7, _mrf_weights(C, 7, 200)), (11, _mrf_weights(C, 11, 300))]
    ys_ref = ops.split_like(B, C, T, DEV)
    ref = _mrf_launch_by_launch(ops, pk, x, xs, branches, 3.0, ys_ref)
    packed = [(k, [(pk(w1.to(DEV)), b1.to(DEV), pk(w2.to(DEV)), b2.to(DEV)) for (w1, b1, w2, b2) in steps]) for k, steps in branches]
    ys = ops.split_like(B, C, T, DEV)
    y = ops.resblock_mrf(xs, B, C, T, packed, out=torch.empty(B, C, T, device=DEV), y_split=ys, y_split_slope=0.1, out_div=3.0,
                         residual_from_planes=True)
    assert torch.equal(y, ref)
    assert torch.equal(ys, ys_ref)
    # one branch alone (no mean): each kernel size against its three launches
    for k, steps in branches:
        r1 = _mrf_launch_by_launch(ops, pk, x, xs, [(k, steps)], 0.0, None)
        p1 = [(k, [(pk(w1.to(DEV)), b1.to(DEV), pk(w2.to(DEV)), b2.to(DEV)) for (w1, b1, w2, b2) in steps])]
        assert torch.equal(ops.resblock_mrf(xs, B, C, T, p1, residual_from_planes=True), r1), k
    # float64 reference of the block
    xd = ops.unsplit(xs).double().cpu()
    xd = torch.where(xd > 0, xd, xd * 10.0)           # what the planes carry: 22 bits of x
    tot = 0
    for k, steps in branches:
        v = xd
        for i, (w1, b1, w2, b2) in enumerate(steps):
            d = 2 * i + 1
            t1 = F.conv1d(F.leaky_relu(v, 0.1), w1.double(), b1.double(), dilation=d, padding=d * (k - 1) // 2)
            v = v + F.conv1d(F.leaky_relu(t1, 0.1), w2.double(), b2.double(), padding=(k - 1) // 2)
        tot = tot + v
    ref64 = tot / 3
    e_planes = (y.cpu().double() - ref64).abs().max().item()
    # default mode: the residual of steps 2 and 3 stays in f32 registers instead of being rebuilt from its 22-bit split —
    # within f32 rounding of the bit-identical mode, and no further from float64 than it
    ys2 = ops.split_like(B, C, T, DEV)
    y2 = ops.resblock_mrf(xs, B, C, T, packed, out=torch.empty(B, C, T, device=DEV), y_split=ys2, y_split_slope=0.1, out_div=3.0)
    e_exact = (y2.cpu().double() - ref64).abs().max().item()
    assert e_planes < 2e-5 and e_exact < 2e-5, (e_planes, e_exact)
    assert e_exact <= 1.5 * e_planes + 1e-7, (e_planes, e_exact)
    assert (y2 - y).abs().max().item() < 4e-6
    assert (ops.unsplit(ys2) - torch.where(y2 > 0, y2, y2 * 0.1)).abs().max().item() < 4e-6


# ---------------------------------------------------------------------------------------------
# split-f16 over the range of a real checkpoint: the per-layer power-of-two weight scale (packing.py) keeps the
# 22-bit split whatever the magnitude of a layer's weights; activations carry 22 bits down to |x| = 0.125 and an
# ABSOLUTE 2^-24 below (f16 subnormals), and must stay below 65504
# ---------------------------------------------------------------------------------------------
REL_2M18 = 2.0 ** -18


@pytest.mark.parametrize("wscale", [1e-5, 1e-3, 3e-2, 1.0, 10.0, 1e3], ids=lambda v: f"w{v:g}")
def test_split_f16_weight_scale_sweep(wscale):
    """a 7-tap conv on split planes (the generator's tile) and a 1024 -> 1024 1x1 conv (the ring GEMM of the wav2vec2
    encoder) against float64, weights of every magnitude: relative error <= 2^-18 of the output scale.  Without the
    layer scale a layer of 1e-5-sized weights only carries ~8 bits (asserted too: the scale is what fixes it)."""
    ops, packing = _ops()
    for (cin, cout, k, T) in [(64, 64, 7, 700), (1024, 1024, 1, 256)]:
        x = _rand(2, cin, T, seed=1).to(DEV)
        w = _rand(cout, cin, k, seed=2, scale=wscale / np.sqrt(cin * k))
        b = _rand(cout, seed=3, scale=wscale)
        ref = F.conv1d(x.double().cpu(), w.double(), b.double(), padding=(k - 1) // 2)
        xs = ops.act_split(x, 1.0)
        wp = packing.pack_conv_weight_f16x3(w.to(DEV))
        y = ops.conv1d(x, wp, cout, k, bias=b.to(DEV), pad_left=(k - 1) // 2, mode=1, x_split=xs)
        rel = (y.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        assert rel <= REL_2M18, (cin, k, wscale, rel)
        if wscale <= 1e-3:
            wp0 = packing.pack_conv_weight_f16x3(w.to(DEV), scale=False)
            assert wp0.w_descale == 1.0
            y0 = ops.conv1d(x, wp0, cout, k, bias=b.to(DEV), pad_left=(k - 1) // 2, mode=1, x_split=xs)
            rel0 = (y0.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
            assert rel0 > 4 * rel, (wscale, rel0, rel)


@pytest.mark.parametrize("xscale", [1e-4, 1e-2, 0.1, 1.0, 1e2, 1e4], ids=lambda v: f"x{v:g}")
def test_split_f16_activation_scale_sweep(xscale):
    """the same two layers with activations of every magnitude.  From 0.1 to 1e4 (|x| < 65504) the result is within
    2^-18 of the output scale; below, the split of x is an ABSOLUTE 2^-24 per element (lo is an f16 subnormal), so the
    error bound is 2^-24 * sum |w| — both asserted, the second is the documented limit of the representation."""
    ops, packing = _ops()
    for (cin, cout, k, T) in [(64, 64, 7, 700), (1024, 1024, 1, 256)]:
        x = (_rand(2, cin, T, seed=1).clamp(-5, 5) * xscale).to(DEV)
        w = _rand(cout, cin, k, seed=2, scale=1.0 / np.sqrt(cin * k))
        ref = F.conv1d(x.double().cpu(), w.double(), None, padding=(k - 1) // 2)
        wp = packing.pack_conv_weight_f16x3(w.to(DEV))
        for xs in (ops.act_split(x, 1.0), None):           # planes input, and f32 input split in the kernel (3 / 7 / 11 taps)
            if xs is None and k == 1:
                continue
            y = ops.conv1d(x, wp, cout, k, pad_left=(k - 1) // 2, mode=1, x_split=xs)
            err = (y.cpu().double() - ref).abs().max().item()
            bound = REL_2M18 * ref.abs().max().item() + 2.0 ** -24 * w.abs().sum(dim=(1, 2)).max().item()
            assert err <= bound, (cin, k, xscale, err, bound)
            if xscale >= 0.1:
                assert err <= REL_2M18 * ref.abs().max().item(), (cin, k, xscale, err)


def test_generator_survives_rescaled_layer_pairs():
    """leaky_relu is positively homogeneous: scaling conv1 (weight and bias) of every ResBlock1 step by 2^8 and its conv2
    weight by 2^-8 leaves the generator's function unchanged while the inner activations t1 grow 256-fold and the conv2
    weights shrink 256-fold (and the other way round).  The reference's f32 path does not care; here the layer scale of
    the packed weights absorbs it: the waveform still matches the CPU oracle at < 1e-5 RMS (bar of the path: 1e-4)."""
    import satools_amd
    from satools_amd import synthetic
    from oracle import hifigan as ogen
    tag = "hifigan_bn_tdnnf_600h_vq_48_v1"
    state, _ = synthetic.checkpoint(tag)
    sd0 = state["base_model_state_dict"]
    x = _rand(2, 504, 40, seed=7)
    for shift in (8, -8):
        sd = {k: v.clone() for k, v in sd0.items()}
        for k in list(sd):
            if k.startswith("hifigan.resblocks.") and k.endswith("weight_g"):
                which = "convs1" if ".convs1." in k else "convs2"
                sd[k] = sd[k] * (2.0 ** shift if which == "convs1" else 2.0 ** -shift)
            if k.startswith("hifigan.resblocks.") and ".convs1." in k and k.endswith(".bias"):
                sd[k] = sd[k] * 2.0 ** shift
        model = satools_amd.load_model("synthetic:" + tag)
        model.load_state_dict(sd)
        model.to(DEV)
        y = model.hifigan(x.to(DEV))[0]
        gsd = {k[len("hifigan."):]: v for k, v in sd.items() if k.startswith("hifigan.")}
        ref = ogen.generator(gsd, x)
        err = rms(y.cpu().numpy() - ref.numpy())
        assert err < 1e-5, (shift, err)


@pytest.mark.parametrize("cin,T,B", [(32, 700, 2), (32, 496, 1), (32, 497, 2), (32, 3000, 2), (64, 240, 1), (64, 241, 2), (64, 1000, 3), (32, 5, 1)],
                         ids=lambda v: str(v))
def test_streaming_upsampler_matches_the_polyphase_conv_and_torch(cin, T, B):
    """ups2_kernel (ConvTranspose1d(C -> C / 2, k 4, stride 2, padding 1) on split planes, the generator's last two
    upsamplers) against the polyphase conv tile it replaces (same split-f16 products, another accumulation order: f32
    rounding) and against torch.conv_transpose1d in float64; tile edges (496 / 240 input positions per tile)"""
    ops, packing = _ops()
    cout = cin // 2
    x = _rand(B, cin, T, seed=1).to(DEV)
    w = _rand(cin, cout, 4, seed=2, scale=0.8 / np.sqrt(cin * 2))
    b = _rand(cout, seed=3, scale=0.1).to(DEV)
    wc, kp, pl = packing.convtranspose_as_phase_conv(w.to(DEV), 2, 1)
    wp = packing.pack_conv_weight_f16x3(wc, up=2)
    xs = ops.act_split(x, 0.1)
    ys = ops.upsample2(xs, wp, b, B, cin, T, y_split_slope=0.1)
    ys_ref = ops.split_like(B, cout, 2 * T, DEV)
    ops.conv1d(x, wp, cout, kp, bias=b, pad_left=pl, up=2, mode=1, x_split=xs, y_split=ys_ref, y_split_slope=0.1, no_y=True)
    got, ref = ops.unsplit(ys), ops.unsplit(ys_ref)
    assert (got - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())
    xd = F.leaky_relu(x.double().cpu(), 0.1)
    y64 = F.leaky_relu(F.conv_transpose1d(xd, w.double(), b.double().cpu(), stride=2, padding=1), 0.1)
    assert (got.cpu().double() - y64).abs().max().item() < 1e-5


@pytest.mark.parametrize("C,k", [(32, 3), (32, 7), (32, 11), (64, 3)])
@pytest.mark.parametrize("T,d,B", [(1, 1, 1), (5, 5, 2), (111, 3, 1), (112, 1, 2), (113, 5, 2), (239, 3, 1), (240, 1, 2), (241, 5, 2), (1000, 3, 3), (4099, 5, 2)], ids=lambda v: str(v))
def test_streaming_resblock_step_matches_the_general_fused_step(T, d, B, C, k):
    """pair32s_kernel (C = 32, 3 taps: both convs' fragments resident) and pairw_kernel (C = 32 at 7 / 11 taps, C = 64 at 3 taps:
    waves specialised by conv, a two-stage pipeline over tiles) — the ResBlock1 steps of the 32- and 64-channel stages
    (`sat_conv_set_option("pair32s" / "pair32w" / "pair64w")`) against the general fused steps they replace (same split-f16
    products, another accumulation order: f32 rounding) and against torch in float64, in the three output forms — planes only
    (the first two steps of a branch), f32 with the MRF accumulation, both; tile edges at 240 (C = 32) and 112 (C = 64)"""
    ops, packing = _ops()
    from satools_amd import _lib
    x = _rand(B, C, T, seed=1).to(DEV)
    pk = packing.pack_conv_weight_f16x3
    w1f, w2f = _rand(C, C, k, seed=2, scale=0.6 / np.sqrt(C * k)), _rand(C, C, k, seed=3, scale=0.6 / np.sqrt(C * k))
    w1, w2 = pk(w1f.to(DEV)), pk(w2f.to(DEV))
    b1, b2 = _rand(C, seed=4, scale=0.1).to(DEV), _rand(C, seed=5, scale=0.1).to(DEV)
    xs = ops.act_split(x, 0.1)
    acc0 = _rand(B, C, T, seed=6).to(DEV)

    def run(opt):
        for name in (b"pair32s", b"pair32w", b"pair64w"):
            _lib.check(_lib.lib().sat_conv_set_option(name, opt), "sat_conv_set_option")
        try:
            ys = ops.split_like(B, C, T, DEV)
            ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=xs, y_split=ys, y_split_slope=0.1, planes_residual=True, no_y=True)
            name_planes = _lib.lib().sat_last_dispatch_name().decode()
            yf = acc0.clone()
            ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=xs, planes_residual=True, out=yf, accum=True, accum_div=3.0)
            yb, ysb = torch.full_like(acc0, 7.0), ops.split_like(B, C, T, DEV)
            ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=xs, y_split=ysb, y_split_slope=0.1, planes_residual=True, out=yb)
            return ops.unsplit(ys), yf, yb, ops.unsplit(ysb), name_planes
        finally:
            for name, default in ((b"pair32s", 1), (b"pair32w", 1), (b"pair64w", 0)):
                _lib.check(_lib.lib().sat_conv_set_option(name, default), "sat_conv_set_option")

    new, old = run(1), run(0)
    kern = "pair32s_kernel" if (C, k) == (32, 3) else "pairw_kernel"
    assert kern in new[4] and kern not in old[4]
    for a, b in zip(new[:4], old[:4]):
        assert (a - b).abs().max().item() < 2e-6 * max(1.0, b.abs().max().item())
    xd = x.double().cpu()
    t1 = F.conv1d(F.leaky_relu(xd, 0.1), w1f.double(), b1.double().cpu(), dilation=d, padding=d * (k - 1) // 2)
    y64 = xd + F.conv1d(F.leaky_relu(t1, 0.1), w2f.double(), b2.double().cpu(), padding=(k - 1) // 2)
    assert (new[2].cpu().double() - y64).abs().max().item() < 1e-5
    assert (new[1].cpu().double() - (acc0.cpu().double() + y64) / 3).abs().max().item() < 1e-5
    assert (new[0].cpu().double() - F.leaky_relu(y64, 0.1)).abs().max().item() < 1e-5
    assert torch.equal(new[3], ops.unsplit(ops.act_split(new[2], 0.1)))       # the planes are the split of the f32 written beside them


@pytest.mark.parametrize("C,T,k,dil", [(256, 1250, 11, 5), (128, 1000, 7, 3), (128, 256, 3, 1), (128, 100, 3, 5), (128, 385, 11, 1), (256, 1100, 7, 1)], ids=lambda v: str(v))
def test_balanced_grid_of_the_conv_tile_gives_the_same_bits(C, T, k, dil):
    """sat_conv_set_option("lean_balance"): the ragged end of every row as 128-column half tiles dispatched after the full
    tiles (1 = only where one half tile covers it, the default; 2 = always, two half tiles if need be) — the same tiles'
    arithmetic, same bits"""
    ops, packing = _ops()
    from satools_amd import _lib
    B = 2
    x, w, b = _rand(B, C, T, seed=1).to(DEV), _rand(C, C, k, seed=2, scale=(k * C) ** -0.5).to(DEV), _rand(C, seed=3).to(DEV)
    xs = ops.act_split(x, 0.1)
    wp = packing.pack_conv_weight_f16x3(w)
    out = {}
    try:
        _lib.check(_lib.lib().sat_conv_set_option(b"convring", 0), "sat_conv_set_option")     # (else C >= 128 goes to conv_ring16.hip)
        for v in (0, 1, 2):
            _lib.check(_lib.lib().sat_conv_set_option(b"lean_balance", v), "sat_conv_set_option")
            ys = ops.split_like(B, C, T, DEV)
            y = ops.conv1d(x, wp, C, k, bias=b, dilation=dil, pad_left=dil * (k - 1) // 2, mode=1, x_split=xs, y_split=ys, y_split_slope=0.1)
            assert T < 1000 or "lean_kernel" in _lib.lib().sat_last_dispatch_name().decode()
            out[v] = (y, ys)
    finally:
        _lib.check(_lib.lib().sat_conv_set_option(b"lean_balance", 1), "sat_conv_set_option")
        _lib.check(_lib.lib().sat_conv_set_option(b"convring", 1), "sat_conv_set_option")
    for v in (1, 2):
        assert torch.equal(out[0][0], out[v][0]) and torch.equal(out[0][1], out[v][1])
    ref = F.conv1d(F.leaky_relu(x.double().cpu(), 0.1), w.double().cpu(), b.double().cpu(), dilation=dil, padding=dil * (k - 1) // 2)
    assert (out[1][0].cpu().double() - ref).abs().max().item() < 3e-5


# ---------------------------------------------------------------------------------------------
# round 4: the LDS-DMA ring conv on the 16x16x32 MFMA shape (csrc/conv_ring16.hip) and sat_conv1d_multi_f32
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,T,k,dil", [(256, 1250, 11, 5), (256, 333, 7, 3), (128, 700, 3, 5), (128, 97, 11, 1), (192, 401, 7, 1), (512, 200, 3, 1),
                                       (320, 161, 11, 3), (128, 5000, 7, 5), (256, 159, 3, 3), (128, 321, 3, 1)], ids=lambda v: str(v))
def test_ring_conv_on_split_planes(C, T, k, dil):
    """conv1d_f16x3_ring16_kernel (C_out > 64, taps >= 3: the generator's resblock convs of the 256- and 128-channel stages,
    reference hifigan/nn.py:96-175) against float64 and against the register-staged tile it replaces: tile edges of both
    layouts (256 x 160, 128 x 320), row counts that do not fill a tile, every dilation of the generator; the three
    epilogues of a ResBlock (planes only; residual from planes; residual + MRF accumulate / 3 with f32 and planes out).
    K = 32 per MFMA associates differently from the 32x32x16 tile: agreement to f32 rounding, not bit for bit; the planes
    it writes are exactly the split of the f32 values it writes beside them"""
    ops, packing = _ops()
    from satools_amd import _lib
    B = 3
    x, w, b = _rand(B, C, T, seed=1).to(DEV), _rand(C, C, k, seed=2, scale=(k * C) ** -0.5).to(DEV), _rand(C, seed=3).to(DEV)
    r, acc0 = _rand(B, C, T, seed=4).to(DEV), _rand(B, C, T, seed=5).to(DEV)
    xs, rs = ops.act_split(x, 0.1), ops.act_split(r, 0.1)
    wp = packing.pack_conv_weight_f16x3(w)
    kw = dict(bias=b, dilation=dil, pad_left=dil * (k - 1) // 2, mode=1, x_split=xs, y_split_slope=0.1)
    conv64 = F.conv1d(F.leaky_relu(x.double().cpu(), 0.1), w.double().cpu(), b.double().cpu(), dilation=dil, padding=dil * (k - 1) // 2)
    r64 = r.double().cpu()
    res = {}
    for ring in (0, 1):
        with conv_option("convring", 33 * ring, 1):          # (+ 32: also for the handful of tiles of a three-utterance batch)
            ys1 = ops.split_like(B, C, T, DEV).zero_()
            ops.conv1d(x, wp, C, k, y_split=ys1, no_y=True, **kw)
            name = _lib.lib().sat_last_dispatch_name().decode()
            assert ("ring16" in name) == bool(ring), name
            ys2 = ops.split_like(B, C, T, DEV).zero_()
            y2 = ops.conv1d(x, wp, C, k, y_split=ys2, res_split=rs, res_split_slope=0.1, **kw)
            ys3 = ops.split_like(B, C, T, DEV).zero_()
            y3 = ops.conv1d(x, wp, C, k, y_split=ys3, res_split=rs, res_split_slope=0.1, out=acc0.clone(), accum=True, accum_div=3.0, **kw)
            res[ring] = (ops.unsplit(ys1), y2, ops.unsplit(ys2), y3, ops.unsplit(ys3), ys2, ys3)
    new, old = res[1], res[0]
    scale = float(conv64.abs().max())
    assert (new[0].cpu().double() - F.leaky_relu(conv64, 0.1)).abs().max().item() < 3e-5
    assert (new[1].cpu().double() - (conv64 + r64)).abs().max().item() < 3e-5
    assert (new[3].cpu().double() - (acc0.double().cpu() + conv64 + r64) / 3).abs().max().item() < 3e-5
    for a, bb in zip(new[:5], old[:5]):
        assert (a - bb).abs().max().item() <= 3e-6 * scale
    assert torch.equal(new[5], ops.act_split(new[1], 0.1)) and torch.equal(new[6], ops.act_split(new[3], 0.1))


@pytest.mark.parametrize("C,T", [(256, 1250), (128, 700), (192, 333)], ids=lambda v: str(v))
def test_conv1d_multi_equals_the_single_calls(C, T):
    """sat_conv1d_multi_f32: the i-th conv of the three branches of an MRF block (kernel sizes 3 / 7 / 11, reference
    hifigan/archi.py:82-86) in ONE launch of the ring kernel — independent jobs (the order rotates from block to block), jobs
    chained through the MRF accumulator (index order), and a set the ring kernel does not serve (falls back to the single
    calls): the bits of the single calls every time"""
    ops, packing = _ops()
    from satools_amd import _lib
    B, ks, dils = 4, (3, 7, 11), (1, 3, 5)
    x = _rand(B, C, T, seed=1).to(DEV)
    xs = ops.act_split(x, 0.1)
    ws = [packing.pack_conv_weight_f16x3(_rand(C, C, k, seed=10 + k, scale=(k * C) ** -0.5).to(DEV)) for k in ks]
    bs = [_rand(C, seed=20 + k).to(DEV) for k in ks]

    def jobs(kind, ys, acc):
        out = []
        for j, k in enumerate(ks):
            kw = dict(bias=bs[j], dilation=dils[j], pad_left=dils[j] * (k - 1) // 2, mode=1, x_split=xs, y_split_slope=0.1)
            if kind == "conv1":
                kw.update(y_split=ys[j], no_y=True)
            elif kind == "conv2":
                kw.update(y_split=ys[j], no_y=True, res_split=xs, res_split_slope=0.1)
            else:
                kw.update(res_split=xs, res_split_slope=0.1, out=acc, accum=j > 0, accum_div=3.0 if j == 2 else 0.0, y_split=ys[2] if j == 2 else None)
            out.append((x, ws[j], C, k, kw))
        return out

    from satools_amd import _lib as L
    L.check(L.lib().sat_conv_set_option(b"convring", 33), "sat_conv_set_option")      # (+ 32: also for the few tiles of this batch)
    try:
        _multi_equals_singles(ops, packing, _lib, jobs, ks, B, C, T, x, xs, bs)
    finally:
        L.check(L.lib().sat_conv_set_option(b"convring", 1), "sat_conv_set_option")


def _multi_equals_singles(ops, packing, _lib, jobs, ks, B, C, T, x, xs, bs):
    for kind in ("conv1", "conv2", "last"):
        got = {}
        for how in ("single", "multi"):
            ys = [ops.split_like(B, C, T, DEV).zero_() for _ in ks]
            acc = torch.full((B, C, T), 7.0, device=DEV)
            if how == "single":
                for (xx, w, c, k, kw) in jobs(kind, ys, acc):
                    ops.conv1d(xx, w, c, k, **kw)
            else:
                ops.conv1d_multi(jobs(kind, ys, acc))
                assert "ring16" in _lib.lib().sat_last_dispatch_name().decode()
            got[how] = (ys, acc)
        for a, bb in zip(got["single"][0], got["multi"][0]):
            assert torch.equal(a, bb), kind
        assert torch.equal(got["single"][1], got["multi"][1]), kind
    # two jobs; and a set with a 1-tap member: served by the single calls
    ys = [ops.split_like(B, C, T, DEV).zero_() for _ in range(2)]
    ops.conv1d_multi(jobs("conv1", ys + [None], None)[:2])
    ys1 = [ops.split_like(B, C, T, DEV).zero_() for _ in range(2)]
    for (xx, w, c, k, kw) in jobs("conv1", ys1 + [None], None)[:2]:
        ops.conv1d(xx, w, c, k, **kw)
    assert all(torch.equal(a, bb) for a, bb in zip(ys, ys1))
    w1 = packing.pack_conv_weight_f16x3(_rand(C, C, 1, seed=5, scale=C ** -0.5).to(DEV))
    ya, yb = ops.split_like(B, C, T, DEV).zero_(), ops.split_like(B, C, T, DEV).zero_()
    j3 = jobs("conv1", [ya, None, None], None)[0]
    j1 = (x, w1, C, 1, dict(bias=bs[0], mode=1, x_split=xs, y_split=yb, y_split_slope=0.1, no_y=True))
    ops.conv1d_multi([j3, j1])
    yc = ops.split_like(B, C, T, DEV).zero_()
    ops.conv1d(x, w1, C, 1, bias=bs[0], mode=1, x_split=xs, y_split=yc, y_split_slope=0.1, no_y=True)
    assert torch.equal(yb, yc)
    with pytest.raises(_lib.SatError):
        ops.conv1d_multi([j3] * 4)


def test_generator_multi_branch_launches_give_the_same_bits(model_f16x3):
    """generator option "multi_branch": the i-th conv of all three MRF branches of the thick stages as one
    sat_conv1d_multi_f32 call against the eighteen single launches per stage"""
    from satools_amd._lib import lib, check
    model = model_f16x3
    g = model.hifigan
    if g.precision not in ("f16x3", "f16f8r"):
        pytest.skip("the ring conv belongs to the split-f16 generator")
    x = torch.randn(3, g.imput_dim, 57, generator=torch.Generator().manual_seed(5)).to(DEV)
    with conv_option("convring", 33, 1):                 # (the ring kernel also for this small batch)
        y1 = g(x)[0].clone()
        check(lib().sat_hifigan_set_option(g._handle, b"multi_branch", 0), "set_option")
        try:
            y0 = g(x)[0].clone()
        finally:
            check(lib().sat_hifigan_set_option(g._handle, b"multi_branch", g.multi_branch), "set_option")
    assert torch.equal(y0, y1)


def test_generator_stride4_upsamplers_on_the_ring(model_f16x3):
    """generator attribute `ups_ring`: the two stride-4 upsamplers packed with their rows grouped by phase and run by the LDS-DMA ring
    (another accumulation order than the 64 x 256 tile: f32 rounding), the row order kept by the frozen export, and the pipelines that
    cannot read that order refusing it"""
    from satools_amd._lib import lib, check, SatError
    model = model_f16x3
    g = model.hifigan
    if g.precision not in ("f16x3", "f16f8r") or not g.split_acts:
        pytest.skip("the grouped rows belong to the split-f16 generator on the split-plane pipeline")
    x = torch.randn(2, g.imput_dim, 41, generator=torch.Generator().manual_seed(6)).to(DEV)
    keep = g.ups_ring
    try:
        g.ups_ring = 1
        y1 = g(x)[0].clone()
        assert g._packed_ups_grouped
        check(lib().sat_hifigan_set_option(g._handle, b"planes_residual", 0), "set_option")
        try:
            with pytest.raises(SatError):
                g(x)
            check(lib().sat_hifigan_set_option(g._handle, b"split_acts", 0), "set_option")
            with pytest.raises(SatError):
                g(x)
        finally:
            check(lib().sat_hifigan_set_option(g._handle, b"split_acts", 1), "set_option")
            check(lib().sat_hifigan_set_option(g._handle, b"planes_residual", 1), "set_option")
        assert torch.equal(y1, g(x)[0])
        g.ups_ring = 0
        y0 = g(x)[0].clone()
        assert not g._packed_ups_grouped
    finally:
        g.ups_ring = keep
    assert rms((y1 - y0).cpu().numpy()) < 5e-7


@pytest.mark.parametrize("B,T", [(3, 249), (2, 500), (5, 250)], ids=lambda v: str(v))
def test_persistent_ring_gemm_gives_the_bits_of_the_ring_gemm(B, T):
    """gemm_f16x3_walk16_kernel (csrc/gemm_walk16.hip: the Linear layers of the wav2vec2 encoder, reference
    tdnnf_wav2vec2_vq.py:39-56 through torchaudio) — a block walks several tiles and up to three GEMMs of one shape (q | k | v)
    with the loop of gemm_f16x3_ring16_kernel: the same bits as that kernel, for every epilogue it carries (bias; GELU + planes;
    f32 residual; f32 output with a row pitch) and for rows / columns that do not fill a tile"""
    ops, packing = _ops()
    from satools_amd import _lib
    tp = (T + 63) // 64 * 64
    for cin, cout, kind in ((512, 256, "plain"), (1024, 384, "gelu_planes"), (256, 128, "res"), (512, 256, "qkv"), (1024, 640, "qkv")):
        x = _rand(B, cin, T, seed=1).to(DEV)
        xs = ops.act_split(x, 1.0)
        nj = 3 if kind == "qkv" else 1
        ws = [packing.pack_conv_weight_f16x3(_rand(cout, cin, 1, seed=10 + j, scale=cin ** -0.5).to(DEV)) for j in range(nj)]
        bs = [_rand(cout, seed=20 + j).to(DEV) for j in range(nj)]
        res = _rand(B, cout, T, seed=30).to(DEV)

        def run(multi):
            if kind == "qkv":
                qs, ks = ops.split_like(B, cout, T, DEV).zero_(), ops.split_like(B, cout, T, DEV).zero_()
                v = torch.zeros(B, cout, tp, device=DEV)
                jobs = [(x, ws[0], cout, 1, dict(bias=bs[0], mode=1, x_split=xs, y_split=qs, y_split_slope=1.0, no_y=True)),
                        (x, ws[1], cout, 1, dict(bias=bs[1], mode=1, x_split=xs, y_split=ks, y_split_slope=1.0, no_y=True)),
                        (x, ws[2], cout, 1, dict(bias=bs[2], mode=1, x_split=xs, out=v[:, :, :T]))]
                if multi:
                    ops.conv1d_multi(jobs)
                else:
                    for (xx, w, c, k, kw) in jobs:
                        ops.conv1d(xx, w, c, k, **kw)
                return (qs, ks, v)
            if kind == "plain":
                return (ops.conv1d(x, ws[0], cout, 1, bias=bs[0], mode=1, x_split=xs),)
            if kind == "res":
                return (ops.conv1d(x, ws[0], cout, 1, bias=bs[0], mode=1, x_split=xs, res=res),)
            ys = ops.split_like(B, cout, T, DEV).zero_()
            ops.conv1d(x, ws[0], cout, 1, bias=bs[0], gelu=True, mode=1, x_split=xs, y_split=ys, y_split_slope=1.0, no_y=True)
            return (ys,)

        with conv_option("gemm_walk", 0, 1):
            old = run(False)
            assert "ring16" in _lib.lib().sat_last_dispatch_name().decode()
        with conv_option("gemm_walk", 3, 1):             # (+ 2: also for a single GEMM with fewer tiles than CUs)
            new = run(True)
            name = _lib.lib().sat_last_dispatch_name().decode()
            assert "walk16" in name, (kind, name)
        for a, bb in zip(old, new):
            assert torch.equal(a, bb), (cin, cout, kind)
        if kind == "plain":
            ref = F.conv1d(x.double().cpu(), _rand(cout, cin, 1, seed=10, scale=cin ** -0.5).double(), bs[0].double().cpu())
            assert (new[0].cpu().double() - ref).abs().max().item() < 3e-5
