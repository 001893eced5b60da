"""lowpass_biquad / highpass_biquad following torchaudio 2.1's published algorithm:
coefficients from the RBJ cookbook in the waveform dtype (f32); `_lfilter` divides b and a by a0
FIRST, takes the FIR part with torch's own `conv1d` over the 2-sample left-padded input and the
flipped b' (as torchaudio's DifferentiableFIR does), then runs the sequential recursion of
`_lfilter_core_loop`  y[t] = fir[t]; y[t] -= a2'*y[t-2]; y[t] -= a1'*y[t-1]  in f32 (separate multiply
and subtract, that order), and clamps to [-1, 1].  Same arithmetic as oracle/biquad.py order="torchaudio"
(which emulates conv1d's FMA chain exactly).  The recursion runs in numpy float32 scalars behind
torch.jit.ignore so that the scripted reference can call it."""
import math

import numpy as np
import torch
from torch import Tensor


def _coeffs(kind: str, sample_rate: int, cutoff_freq: float, Q: float):
    f32 = torch.float32
    cutoff = torch.as_tensor(cutoff_freq, dtype=f32)
    q = torch.as_tensor(Q, dtype=f32)
    w0 = 2 * math.pi * cutoff / sample_rate
    alpha = torch.sin(w0) / 2 / q
    if kind == "lp":
        b0 = (1 - torch.cos(w0)) / 2
        b1 = 1 - torch.cos(w0)
    else:
        b0 = (1 + torch.cos(w0)) / 2
        b1 = -1 - torch.cos(w0)
    b2 = b0
    a0 = 1 + alpha
    a1 = -2 * torch.cos(w0)
    a2 = 1 - alpha
    return (torch.stack([b0, b1, b2]).to(f32), torch.stack([a0, a1, a2]).to(f32))


def _lfilter(x: Tensor, b: Tensor, a: Tensor) -> Tensor:
    shape = x.shape
    xn = x.reshape(-1, shape[-1]).to(torch.float32).numpy()
    bn = b.numpy().astype(np.float32)
    an = a.numpy().astype(np.float32)
    c1, c2 = np.float32(an[1] / an[0]), np.float32(an[2] / an[0])
    out = np.empty_like(xn)
    # FIR part as in torchaudio: conv1d of the padded waveform with the flipped b / a0
    bnorm = (b.to(torch.float32) / a.to(torch.float32)[0:1]).flip(0).contiguous().view(1, 1, 3)
    xpad = torch.nn.functional.pad(x.reshape(-1, 1, shape[-1]).to(torch.float32), (2, 0))
    fir = torch.nn.functional.conv1d(xpad, bnorm).reshape(-1, shape[-1]).numpy()
    for r in range(xn.shape[0]):
        f = fir[r]
        o = out[r]
        y1 = np.float32(0.0)
        y2 = np.float32(0.0)
        for t in range(f.shape[0]):
            v = f[t] - c2 * y2
            v = v - c1 * y1
            o[t] = v
            y2 = y1
            y1 = v
    y = torch.from_numpy(out)
    return torch.clamp(y, -1.0, 1.0).reshape(shape)


@torch.jit.ignore
def _lp(waveform: Tensor, sample_rate: int, cutoff_freq: float, Q: float) -> Tensor:
    b, a = _coeffs("lp", sample_rate, cutoff_freq, Q)
    return _lfilter(waveform, b, a)


@torch.jit.ignore
def _hp(waveform: Tensor, sample_rate: int, cutoff_freq: float, Q: float) -> Tensor:
    b, a = _coeffs("hp", sample_rate, cutoff_freq, Q)
    return _lfilter(waveform, b, a)


def lowpass_biquad(waveform: Tensor, sample_rate: int, cutoff_freq: float, Q: float = 0.707) -> Tensor:
    return _lp(waveform, sample_rate, cutoff_freq, Q)


def highpass_biquad(waveform: Tensor, sample_rate: int, cutoff_freq: float, Q: float = 0.707) -> Tensor:
    return _hp(waveform, sample_rate, cutoff_freq, Q)
