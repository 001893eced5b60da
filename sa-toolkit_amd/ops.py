"""Thin Python wrappers over the C ABI (tensor checks + descriptor filling).
All tensors are f32, channel-major [B, C, T], on the HIP device."""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, check, lib, ptr, stream


def _f32c(t):
    if t.dtype != torch.float32:
        raise _lib.SatError(f"expected float32 tensor, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def conv1d(x, w_packed, c_out, ksize, *, bias=None, dilation=1, stride=1, pad_left=0, pad_right=None, groups=1,
           up=1, in_lrelu=None, res=None, res_scale=1.0, res_toff=0, res_tstride=1, ch_scale=None, ch_shift=None,
           relu=False, out=None, accum=False, accum_div=0.0, mode=0):
    """Fused conv (see include/satools_hip.h sat_conv1d_f32).  `pad_right` defaults to the
    'same'-style value implied by pad_left for stride 1; T_q is derived like torch does:
    T_q = (T_in + pad_left + pad_right - dilation*(ksize-1) - 1)//stride + 1."""
    x = _f32c(x)
    B, c_in, t_in = x.shape
    if pad_right is None:
        pad_right = dilation * (ksize - 1) - pad_left if up == 1 and stride == 1 else 0
    if up > 1:
        t_q = t_in
    else:
        t_q = (t_in + pad_left + pad_right - dilation * (ksize - 1) - 1) // stride + 1
    if t_q <= 0:
        raise _lib.SatError("conv1d: input too short for this kernel")
    if out is None:
        out = torch.empty(B, c_out, t_q * up, dtype=torch.float32, device=x.device)
    d = ConvDesc()
    d.B, d.C_in, d.T_in, d.C_out, d.T_q = B, c_in, t_in, c_out, t_q
    d.ksize, d.dilation, d.stride, d.pad_left, d.groups, d.up = ksize, dilation, stride, pad_left, groups, up
    d.mode = int(mode)
    d.in_lrelu = 0 if in_lrelu is None else 1
    d.in_slope = 0.0 if in_lrelu is None else float(in_lrelu)
    d.relu = int(relu)
    d.accum = int(accum)
    d.accum_div = float(accum_div)
    d.res_scale, d.res_toff, d.res_tstride = float(res_scale), int(res_toff), int(res_tstride)
    d.x_bstride, d.x_cstride = x.stride(0), x.stride(1)
    d.y_bstride, d.y_cstride = out.stride(0), out.stride(1)
    if res is not None:
        res = _f32c(res)
        d.res_bstride, d.res_cstride = res.stride(0), res.stride(1)
    d.bias = ptr(bias)
    d.res = ptr(res)
    d.ch_scale = ptr(ch_scale)
    d.ch_shift = ptr(ch_shift)
    check(lib().sat_conv1d_f32(C.byref(d), ptr(x), ptr(w_packed), ptr(out), stream()), "sat_conv1d_f32")
    return out


def convpost(x, w, bias):
    x = _f32c(x)
    B, c, t = x.shape
    y = torch.empty(B, 1, t + 1, dtype=torch.float32, device=x.device)
    check(lib().sat_hifigan_convpost_f32(ptr(x), ptr(w), ptr(bias), ptr(y), B, c, t, stream()),
          "sat_hifigan_convpost_f32")
    return y


def fbank_cmvn_pad(wav, window, mel, mel_lo, mel_hi, *, scale=32768.0, pad=0, cmvn=True):
    wav = _f32c(wav)
    B, n = wav.shape
    n_mel = mel.shape[0]
    m = (n + 80) // 160
    ws_bytes = lib().sat_fbank_workspace_bytes(B, n)
    ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=wav.device)
    out = torch.empty(B, n_mel, m + 2 * pad, dtype=torch.float32, device=wav.device)
    check(lib().sat_fbank_cmvn_pad_f32(ptr(wav), ptr(out), ptr(window), ptr(mel), ptr(mel_lo), ptr(mel_hi), ptr(ws),
                                       ws_bytes, B, n, float(scale), n_mel, pad, int(cmvn), stream()),
          "sat_fbank_cmvn_pad_f32")
    return out


def vq(z, codebook, want_dist=False):
    z = _f32c(z)
    B, D, T = z.shape
    n_codes = codebook.shape[0]
    q = torch.empty_like(z)
    idx = torch.empty(B, T, dtype=torch.int32, device=z.device)
    dist = torch.empty(B, T, n_codes, dtype=torch.float32, device=z.device) if want_dist else None
    check(lib().sat_vq_argmin_gather_f32(ptr(z), ptr(codebook), ptr(q), ptr(idx), ptr(dist), B, D, T, n_codes,
                                         stream()), "sat_vq_argmin_gather_f32")
    return q, idx, dist


def pad_replicate(x, left, right, interleave_right=False):
    x = _f32c(x)
    B, c, t = x.shape
    y = torch.empty(B, c, left + t + right, dtype=torch.float32, device=x.device)
    check(lib().sat_pad_replicate_f32(ptr(x), ptr(y), B, c, t, left, right, int(interleave_right), stream()),
          "sat_pad_replicate_f32")
    return y


def f0_norm_transform_(f0, quant_bins=0, noise=None):
    """in place on a contiguous device tensor: batch-coupled mean/var normalisation over the
    non-zero entries, optional quantisation and additive noise"""
    if not f0.is_contiguous():
        raise _lib.SatError("f0 must be contiguous for the in-place normalisation")
    n = f0.numel()
    stats = torch.empty(2, dtype=torch.float32, device=f0.device)
    check(lib().sat_f0_stats_f32(ptr(f0), n, ptr(stats), stream()), "sat_f0_stats_f32")
    if noise is not None:
        noise = _f32c(noise)
        assert noise.numel() == n
    check(lib().sat_f0_apply_f32(ptr(f0), n, ptr(stats), int(quant_bins), ptr(noise), stream()), "sat_f0_apply_f32")
    return f0


def assemble_input(bn, f0, spk_idx, n_spk):
    bn = _f32c(bn)
    f0 = _f32c(f0)
    B, c_bn, T = bn.shape
    t_f0 = f0.shape[-1]
    assert f0.numel() == B * t_f0
    x = torch.empty(B, c_bn + 1 + n_spk, T, dtype=torch.float32, device=bn.device)
    check(lib().sat_assemble_input_f32(ptr(bn), ptr(f0), ptr(spk_idx), ptr(x), B, c_bn, T, t_f0, n_spk, stream()),
          "sat_assemble_input_f32")
    return x
