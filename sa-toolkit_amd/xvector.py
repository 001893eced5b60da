"""ASV x-vector extractor (ECAPA-TDNN) behind the reference's `Net` interface
(reference: egs/asv/voxceleb/local/tuning/ecapa_tdnn.py:18-81; SURVEY row aX / §8 f3): `model(wav)` returns
`((loss, logits), x_vector)` like the reference's forward with `target=None` — loss = NaN, logits = None, the
L2-normalised 192-dim embedding — computed on the HIP kernels:

  front end   pre-emphasis + 1024-point power spectrum + 80 mel + log, fused   csrc/xvector.hip  melspec_logmel_kernel
              InstanceNorm1d over time                                        instnorm_rows_kernel
  ECAPA body  every Conv1d / Linear on the fused conv kernel (exact f32 MFMA), ReLU-then-BatchNorm epilogue
              Res2Net partial sums, SE gate (+ the block's skip connections)   add3_kernel, se_gate_add_kernel
  pooling     tanh, softmax over time + weighted mean / std                    tanh_kernel, attentive_stats_kernel
  head        Linear(3072 -> 192) + BatchNorm on the conv kernel, L2 norm      l2norm_rows_kernel

The parameter tree carries the reference's state-dict keys (torchaudio's `MelSpec.spectrogram.window` /
`mel_scale.fb` buffers included), so a reference checkpoint loads with `load_state_dict`.  No CPU fallback."""
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib, ops, packing


class _Conv1dReluBn(nn.Module):
    def __init__(self, cin, cout, k=1):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, k, bias=False)
        self.bn = nn.BatchNorm1d(cout)


class _Res2Conv1dReluBn(nn.Module):
    def __init__(self, channels, k, scale):
        super().__init__()
        self.scale, self.width, self.nums = scale, channels // scale, scale - 1
        self.convs = nn.ModuleList([nn.Conv1d(self.width, self.width, k, bias=False) for _ in range(self.nums)])
        self.bns = nn.ModuleList([nn.BatchNorm1d(self.width) for _ in range(self.nums)])


class _SEConnect(nn.Module):
    def __init__(self, channels, s=2):
        super().__init__()
        self.linear1 = nn.Linear(channels, channels // s)
        self.linear2 = nn.Linear(channels // s, channels)


def _se_res2block(channels, k, scale):
    return nn.Sequential(_Conv1dReluBn(channels, channels), _Res2Conv1dReluBn(channels, k, scale),
                         _Conv1dReluBn(channels, channels), _SEConnect(channels))


class _PreEcapaTDNN(nn.Module):
    def __init__(self, in_feature=80, channels=512):
        super().__init__()
        self.layer1 = _Conv1dReluBn(in_feature, channels, 5)
        self.layer2 = _se_res2block(channels, 3, 8)
        self.layer3 = _se_res2block(channels, 3, 8)
        self.layer4 = _se_res2block(channels, 3, 8)
        self.conv = nn.Conv1d(channels * 3, channels * 3, 1)


class _Buffers(nn.Module):
    def __init__(self, **bufs):
        super().__init__()
        for k, v in bufs.items():
            self.register_buffer(k, v)


class _MelSpec(nn.Module):
    def __init__(self):
        super().__init__()
        self.spectrogram = _Buffers(window=torch.hann_window(400, periodic=True))
        self.mel_scale = _Buffers(fb=mel_filterbank())


class _MelSpecFrontEnd(nn.Module):
    def __init__(self):
        super().__init__()
        self.PreEmphasis = _Buffers(flipped_filter=torch.tensor([[[-0.97, 1.0]]]))
        self.MelSpec = _MelSpec()


class _AttentiveStatsPool(nn.Module):
    def __init__(self, in_dim, bottleneck):
        super().__init__()
        self.linear1 = nn.Conv1d(in_dim, bottleneck, 1)
        self.linear2 = nn.Conv1d(bottleneck, in_dim, 1)


class _ArcMargin(nn.Module):
    def __init__(self, emb, n):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(n, emb))


def mel_filterbank(n_freqs=513, f_min=90.0, f_max=7600.0, n_mels=80, sample_rate=16000):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") as configured by
    sidekit/preprocessor.py:181-213 — [n_freqs, n_mels]; third party, restated (parity unpinned)"""
    hz2mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def build(args=None):
    """same contract as the reference's model-config `build(args)`: returns the Net class"""

    class Net(nn.Module):
        #: arithmetic of the wide 1x1 convs: "f16x3" (split-f16 on the f16 matrix cores) or "f32" (exact f32 MFMA)
        precision = os.environ.get("SATOOLS_AMD_XVECTOR_PRECISION", "f16x3")
        #: the Res2Net chain of a block as one launch (ops.res2_chain) instead of conv by conv
        res2_chain = os.environ.get("SATOOLS_AMD_XVECTOR_RES2_CHAIN", "1") != "0"

        def __init__(self, num_speakers=1):
            super().__init__()
            self.preprocessor = _MelSpecFrontEnd()
            self.sequence_network = _PreEcapaTDNN(80, 512)
            self.embedding_size = 192
            self.before_speaker_embedding = nn.Sequential(OrderedDict([
                ("lin", nn.Linear(3072, self.embedding_size, bias=False)), ("bn2", nn.BatchNorm1d(self.embedding_size))]))
            self.stat_pooling = _AttentiveStatsPool(1536, 128)
            self.after_speaker_embedding = _ArcMargin(self.embedding_size, num_speakers)
            self._cache, self._cache_key = None, None
            super().eval()

        def train(self, mode=True):
            if mode:
                raise _lib.SatError("the MI355X x-vector extractor is inference only")
            return super().train(False)

        # ---- kernel-ready weights ----------------------------------------------------------------
        def _prepare(self, device):
            key = (self.precision,) + tuple((p.data_ptr(), p._version, str(p.device)) for p in list(self.parameters()) + list(self.buffers()))
            if self._cache_key == key:
                return self._cache
            _lib.cache_rebuild_begin(device, self._cache is not None)
            f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
            # the wide 1x1 convs (block in / out, the 1536 -> 1536 conv before pooling, the attention MLP) run as
            # split-f16 products on the f16 matrix cores (~2^-21 per product) unless precision is "f32"
            split = self.precision == "f16x3"
            pack1 = packing.pack_conv_weight_f16x3 if split else packing.pack_conv_weight
            m1 = _lib.CONV_F16X3 if split else _lib.CONV_F32

            def bn_affine(bn):
                s = f32(bn.weight) / torch.sqrt(f32(bn.running_var) + bn.eps)
                return s.contiguous(), (f32(bn.bias) - f32(bn.running_mean) * s).contiguous()

            def crb(m):
                sc, sh = bn_affine(m.bn)
                k = m.conv.weight.shape[2]
                wide = k == 1 and m.conv.weight.shape[1] % 16 == 0
                return {"w": (pack1 if wide else packing.pack_conv_weight)(f32(m.conv.weight)), "k": k, "scale": sc, "shift": sh,
                        "cout": m.conv.weight.shape[0], "mode": m1 if wide else _lib.CONV_F32}

            W = {"layer1": crb(self.sequence_network.layer1), "blocks": []}
            for lay in (self.sequence_network.layer2, self.sequence_network.layer3, self.sequence_network.layer4):
                res2 = []
                for conv, bn in zip(lay[1].convs, lay[1].bns):
                    sc, sh = bn_affine(bn)
                    res2.append({"w": packing.pack_conv_weight(f32(conv.weight)), "scale": sc, "shift": sh})
                chain = None
                if all(tuple(c.weight.shape) == (64, 64, 3) and c.bias is None for c in lay[1].convs):
                    # the whole Res2Conv1dReluBn in one launch (ops.res2_chain): weights [piece][tap][ci][co], the BatchNorms as affines
                    aff = [bn_affine(bn) for bn in lay[1].bns]
                    chain = {"w": torch.stack([f32(c.weight).permute(2, 1, 0) for c in lay[1].convs]).contiguous(),
                             "scale": torch.stack([a[0] for a in aff]).contiguous(), "shift": torch.stack([a[1] for a in aff]).contiguous()}
                W["blocks"].append({
                    "in": crb(lay[0]), "res2": res2, "chain": chain, "out": crb(lay[2]),
                    "se1_w": f32(lay[3].linear1.weight), "se1_b": f32(lay[3].linear1.bias),
                    "se2_w": f32(lay[3].linear2.weight), "se2_b": f32(lay[3].linear2.bias)})
            sn = self.sequence_network
            W["cat"] = {"w": pack1(f32(sn.conv.weight)), "b": f32(sn.conv.bias)}
            W["mode1"] = m1
            sp = self.stat_pooling
            W["asp1"] = {"w": pack1(f32(sp.linear1.weight)), "b": f32(sp.linear1.bias)}
            W["asp2"] = {"w": pack1(f32(sp.linear2.weight)), "b": f32(sp.linear2.bias)}
            sc, sh = bn_affine(self.before_speaker_embedding.bn2)
            W["emb"] = {"w": f32(self.before_speaker_embedding.lin.weight), "scale": sc, "shift": sh}
            W["window"] = f32(self.preprocessor.MelSpec.spectrogram.window)
            W["fb"] = f32(self.preprocessor.MelSpec.mel_scale.fb).t().contiguous()      # [80][513]
            W["coef"] = float(-self.preprocessor.PreEmphasis.flipped_filter.reshape(-1)[0])
            self._cache, self._cache_key = W, key
            _lib.cache_rebuild_end(device)
            return W

        # ---- forward -----------------------------------------------------------------------------
        def features(self, x):
            """[B, n] -> [B, 80, 1 + n // 160]: log-mel front end + InstanceNorm (sidekit/preprocessor.py:223-236)"""
            W = self._prepare(x.device)
            return ops.instnorm_rows(ops.melspec_logmel(x, W["window"], W["fb"], W["coef"]))

        def _crb(self, x, e, pad=0, dil=1, out=None):
            return ops.conv1d(x, e["w"], e["cout"], e["k"], pad_left=pad, dilation=dil, ch_scale=e["scale"], ch_shift=e["shift"],
                              relu=True, relu_first=True, out=out, mode=e["mode"])

        def _block(self, x, blk, dil, skips, out):
            """SE_Res2Block (sidekit/nn.py:142-154) + the skip connections of PreEcapaTDNN.forward; writes `out`"""
            y = self._crb(x, blk["in"])
            B, C, T = y.shape
            width = C // 8
            if blk["chain"] is not None and self.res2_chain and dil * len(blk["res2"]) <= 32 and dil <= 4:
                z_in = ops.res2_chain(y, blk["chain"]["w"], blk["chain"]["scale"], blk["chain"]["shift"], dil)
            else:
                z_in = torch.empty_like(y)
                prev = None
                for i, e in enumerate(blk["res2"]):
                    piece = y[:, i * width:(i + 1) * width]
                    inp = piece if i == 0 else ops.add3(prev, piece)
                    prev = ops.conv1d(inp, e["w"], width, 3, pad_left=dil, dilation=dil, ch_scale=e["scale"], ch_shift=e["shift"],
                                      relu=True, relu_first=True, out=z_in[:, i * width:(i + 1) * width])
                z_in[:, 7 * width:].copy_(y[:, 7 * width:])
            z = self._crb(z_in, blk["out"])
            m = ops.row_mean(z)                                                    # [B, C, 1]
            g = ops.linear_rows(m, blk["se1_w"], bias=blk["se1_b"], relu=True)          # (the pooled frame: matrix-vector products)
            g = ops.linear_rows(g, blk["se2_w"], bias=blk["se2_b"])
            return ops.se_gate_add(z, g, skips, out=out)

        def embed(self, feats):
            W = self._prepare(feats.device)
            out1 = self._crb(feats, W["layer1"], pad=2)
            B, C, T = out1.shape
            cat = torch.empty(B, 3 * C, T, dtype=torch.float32, device=feats.device)
            out2 = self._block(out1, W["blocks"][0], 2, [out1], cat[:, :C])
            out3 = self._block(ops.add3(out1, out2), W["blocks"][1], 3, [out1, out2], cat[:, C:2 * C])
            self._block(ops.add3(out1, out2, out3), W["blocks"][2], 4, [out1, out2, out3], cat[:, 2 * C:])
            h = ops.conv1d(cat, W["cat"]["w"], 3 * C, 1, bias=W["cat"]["b"], relu=True, mode=W["mode1"])
            a = ops.tanh_(ops.conv1d(h, W["asp1"]["w"], 128, 1, bias=W["asp1"]["b"], mode=W["mode1"]))
            logits = ops.conv1d(a, W["asp2"]["w"], 3 * C, 1, bias=W["asp2"]["b"], mode=W["mode1"])
            pooled = ops.attentive_stats(h, logits)                                 # [B, 2 * 3C, 1]
            e = ops.linear_rows(pooled, W["emb"]["w"], ch_scale=W["emb"]["scale"], ch_shift=W["emb"]["shift"])
            return ops.l2norm_rows(e.reshape(B, self.embedding_size))

        def forward(self, x, target=None):
            if target is not None:
                raise _lib.SatError("the MI355X x-vector extractor is inference only (target must be None)")
            if not x.is_cuda:
                raise _lib.SatError("x-vector extraction runs on the HIP device only (no CPU fallback)")
            x = x.to(torch.float32)
            if x.dim() == 1:
                x = x.unsqueeze(0)
            xv = self.embed(self.features(x.contiguous()))
            return (torch.tensor(float("nan")), None), xv

    return Net
