"""Parameter trees with the reference's state-dict key names.

The reference checkpoint (`base_model_state_dict`) is part of the drop-in boundary
(reference: satools/satools/infer_helper.py:57, hifigan/model.py:146-160).  These holder
modules carry the tensors under the exact attribute paths the reference's modules use, so that
`load_state_dict` / `state_dict` round-trip a reference checkpoint unchanged.  They hold no
torch compute: every forward goes through the HIP library (see hifigan.py / asrbn.py).

Key inventory was taken from the reference model built by import (427 keys for the fbank tag);
tests/test_state_dict_keys.py pins it against tests/golden/state_dict_keys_*.json.
"""
import torch
import torch.nn as nn


def _p(*shape):
    return nn.Parameter(torch.zeros(*shape), requires_grad=False)


class WeightNormConv(nn.Module):
    """Holder for a `torch.nn.utils.weight_norm`-wrapped conv: bias, weight_g, weight_v
    (reference: hifigan/archi.py:40,50,70; hifigan/nn.py:98-163).  After
    remove_weight_norm() the reference stores a plain `weight`; both forms are accepted."""

    def __init__(self, v_shape, n_bias):
        super().__init__()
        self.bias = _p(n_bias)
        self.weight_g = _p(v_shape[0], 1, 1)
        self.weight_v = _p(*v_shape)

    def folded_weight(self) -> torch.Tensor:
        """w = g * v / ||v||, norm over every dim but 0 (for ConvTranspose1d dim 0 is the
        INPUT channel axis, as in torch's weight_norm(dim=0)).  Same arithmetic as
        torch._weight_norm so that folding at load is exact."""
        if hasattr(self, "weight") and self.weight is not None and "weight" in self._parameters:
            return self.weight.detach()
        return torch._weight_norm(self.weight_v.detach(), self.weight_g.detach(), 0)

    def remove_weight_norm(self):
        if "weight_v" not in self._parameters:
            return
        w = self.folded_weight().clone()
        del self._parameters["weight_g"]
        del self._parameters["weight_v"]
        # reference order after remove_weight_norm: bias, weight
        self.weight = nn.Parameter(w, requires_grad=False)


class ResBlock1Params(nn.Module):
    def __init__(self, ch, k):
        super().__init__()
        self.convs1 = nn.ModuleList([WeightNormConv((ch, ch, k), ch) for _ in range(3)])
        self.convs2 = nn.ModuleList([WeightNormConv((ch, ch, k), ch) for _ in range(3)])


class CoreHifiGanParams(nn.Module):
    """reference: satools/satools/hifigan/archi.py:21-75"""

    def __init__(self, imput_dim, upsample_rates=(5, 4, 4, 2, 2), upsample_kernel_sizes=(11, 8, 8, 4, 4),
                 upsample_initial_channel=512, resblock_kernel_sizes=(3, 7, 11),
                 resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5))):
        super().__init__()
        self.imput_dim = imput_dim
        self.upsample_rates = tuple(upsample_rates)
        self.upsample_kernel_sizes = tuple(upsample_kernel_sizes)
        self.upsample_initial_channel = upsample_initial_channel
        self.resblock_kernel_sizes = tuple(resblock_kernel_sizes)
        self.resblock_dilation_sizes = tuple(tuple(d) for d in resblock_dilation_sizes)
        c0 = upsample_initial_channel
        self.conv_pre = WeightNormConv((c0, imput_dim, 7), c0)
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
            self.ups.append(WeightNormConv((cin, cout, k), cout))
        self.resblocks = nn.ModuleList()
        ch = c0
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for k in resblock_kernel_sizes:
                self.resblocks.append(ResBlock1Params(ch, k))
        self.conv_post = WeightNormConv((1, ch, 7), 1)

    def remove_weight_norm(self):
        for m in self.modules():
            if isinstance(m, WeightNormConv):
                m.remove_weight_norm()


class _InnerNat(nn.Module):
    def __init__(self, feat, out):
        super().__init__()
        self.weight = _p(out, feat)
        self.bias = _p(1, out)


class _OrthoLinear(nn.Module):
    def __init__(self, feat, out):
        super().__init__()
        self.inner_nat = _InnerNat(feat, out)


class _Linear(nn.Module):
    def __init__(self, feat, out):
        super().__init__()
        self.weight = _p(out, feat)
        self.bias = _p(out)


class _Embedding(nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.weight = _p(n, d)


class _VQEma(nn.Module):
    """reference: satools/satools/chain/nn.py:377-400 (state: _ema_w, _ema_cluster_size, _embedding.weight)"""

    def __init__(self, n, d):
        super().__init__()
        self._ema_w = _p(n, d)
        self.register_buffer("_ema_cluster_size", torch.zeros(n))
        self._embedding = _Embedding(n, d)


class _VQLayer(nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.quant = _VQEma(n, d)


class _TDNNF(nn.Module):
    def __init__(self, feat, out, bottleneck, ctx, vq=None):
        super().__init__()
        if vq is not None:
            self.bottleneck_func = vq
        self.linearB = _OrthoLinear(feat * ctx, bottleneck)
        self.linearA = _Linear(bottleneck, out)


class _BNStats(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class TDNNFBatchNormParams(nn.Module):
    """reference: satools/satools/chain/nn.py:308-347 (+ TDNNF :197-304).  The VQ module is
    registered twice in the reference (on TDNNFBatchNorm and on its inner TDNNF), which
    duplicates its three tensors in the state dict; the same object is shared here too."""

    def __init__(self, feat, out, bottleneck, ctx=1, sub=1, bypass_scale=0.66, vq_codes=0):
        super().__init__()
        self.feat_dim, self.out_dim, self.bottleneck_dim = feat, out, bottleneck
        self.context_len, self.subsampling_factor = ctx, sub
        self.bypass_scale = bypass_scale
        self.use_bypass = bypass_scale > 0.0 and feat == out
        vq = _VQLayer(vq_codes, bottleneck) if vq_codes else None
        if vq is not None:
            self.bottleneck_func = vq
        self.tdnn = _TDNNF(feat, out, bottleneck, ctx, vq)
        self.bn = _BNStats(out)


class _Identity(nn.Module):
    """placeholder for the reference's Dropout entries (odd indices of `tdnnfs`)"""


def tdnnf_stack(in_dim, hidden, bottleneck, prefinal_bottleneck, kernel_sizes, subs, vq_codes):
    """(tdnn1, tdnnfs) as the reference constructs them:
    tdnnf_vq.py:49-115 / tdnnf_wav2vec2_vq.py:66-129."""
    tdnn1 = TDNNFBatchNormParams(in_dim, hidden, bottleneck, kernel_sizes[0], subs[0])
    seq = []
    n = len(kernel_sizes)
    for i in range(1, n - 1):
        seq += [TDNNFBatchNormParams(hidden, hidden, bottleneck, kernel_sizes[i], subs[i]), _Identity()]
    seq += [TDNNFBatchNormParams(hidden, hidden, prefinal_bottleneck, kernel_sizes[n - 1], subs[n - 1],
                                 bypass_scale=0.0, vq_codes=vq_codes), _Identity()]
    return tdnn1, nn.Sequential(*seq)
