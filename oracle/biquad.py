"""Band-limiting biquads used by YAAPT (reference call site: satools/satools/hifigan/yaapt.py:42-51
-> torchaudio.functional.lowpass_biquad / highpass_biquad).

THIRD-PARTY, PARITY UNPINNED: torchaudio is a dependency of the reference whose source is not under
/root/reference and which is not installed here (its version is not pinned by the reference either,
install.sh:115-116 -> 2.1.x).  This restates torchaudio 2.1's published algorithm:
  * RBJ cookbook coefficients computed in the waveform dtype (f32), Q = 0.707;
  * lfilter: FIR part = conv1d of the 2-sample left-padded input with [b2, b1, b0], divided by a0;
    a-coefficients divided by a0; sequential recursion in f32
        y[t] = fir[t] - a2'*y[t-2] - a1'*y[t-1]     (multiply, subtract; that order; no fma)
  * output clamped to [-1, 1] (lfilter(clamp=True)); the recursion itself runs on unclamped values.
The FIR sum order is fixed here as (b2*x[t-2] + b1*x[t-1]) + b0*x[t] without fma."""
import math

import numpy as np
import torch


def coeffs(kind, sample_rate, cutoff, Q=0.707):
    f32 = torch.float32
    w0 = 2 * math.pi * torch.as_tensor(cutoff, dtype=f32) / sample_rate
    alpha = torch.sin(w0) / 2 / torch.as_tensor(Q, dtype=f32)
    if kind == "lp":
        b0 = (1 - torch.cos(w0)) / 2
        b1 = 1 - torch.cos(w0)
    else:
        b0 = (1 + torch.cos(w0)) / 2
        b1 = -1 - torch.cos(w0)
    a0, a1, a2 = 1 + alpha, -2 * torch.cos(w0), 1 - alpha
    b = np.array([float(b0), float(b1), float(b0)], dtype=np.float32)
    a = np.array([float(a0), float(a1), float(a2)], dtype=np.float32)
    return b, a


def kernel_constants(kind, sample_rate, cutoff):
    """(b0, b1, b2, a0, c1 = a1/a0, c2 = a2/a0) as f32 — what the recursion uses"""
    b, a = coeffs(kind, sample_rate, cutoff)
    return b[0], b[1], b[2], a[0], np.float32(a[1] / a[0]), np.float32(a[2] / a[0])


def biquad(x, kind, sample_rate, cutoff):
    """x: 1-D float32 numpy array -> filtered, clamped float32 array"""
    b0, b1, b2, a0, c1, c2 = kernel_constants(kind, sample_rate, cutoff)
    x = np.asarray(x, dtype=np.float32)
    xp = np.concatenate([np.zeros(2, np.float32), x])
    fir = ((b2 * xp[:-2] + b1 * xp[1:-1]) + b0 * xp[2:]) / a0
    out = np.empty_like(x)
    y1 = np.float32(0)
    y2 = np.float32(0)
    for t in range(x.shape[0]):
        v = fir[t] - c2 * y2
        v = v - c1 * y1
        out[t] = v
        y2 = y1
        y1 = v
    return np.clip(out, np.float32(-1), np.float32(1))


def band_limit(x, sample_rate=16000, bp_low=50.0, bp_high=1500.0):
    """SignalObj.filtered_version (yaapt.py:42-51): low-pass at bp_low THEN high-pass at bp_high
    (yes, as written in the reference), each through its own clamped lfilter"""
    return biquad(biquad(x, "lp", sample_rate, bp_low), "hp", sample_rate, bp_high)
