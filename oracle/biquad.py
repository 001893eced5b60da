"""Band-limiting biquads used by YAAPT (reference call site: satools/satools/hifigan/yaapt.py:42-51
-> torchaudio.functional.lowpass_biquad / highpass_biquad).  Test infrastructure only.

THIRD-PARTY, PARITY UNPINNED: torchaudio is a dependency of the reference whose source is not under
/root/reference and which is not installed here (its version is not pinned by the reference either,
install.sh:115-116 -> 2.1.x).  This restates torchaudio 2.1's published algorithm
(`functional/filtering.py`: `lowpass_biquad` -> `biquad` -> `lfilter` -> `_lfilter`; the recursion is
`_lfilter_core_loop`, csrc/lfilter.cpp `host_lfilter_core_loop`):
  * RBJ cookbook coefficients computed in the waveform dtype (f32), Q = 0.707;
  * `_lfilter` normalises FIRST: b' = b / a0, a' = a / a0 (f32 divisions of the coefficients), then
    FIR part = `conv1d` of the 2-sample left-padded input with the flipped [b2', b1', b0'];
  * sequential recursion in f32, exactly the C++ loop's order
        y[t] = fir[t];  y[t] -= a2'*y[t-2];  y[t] -= a1'*y[t-1]     (multiply, subtract; no fma)
  * output clamped to [-1, 1] (lfilter(clamp=True)); the recursion itself runs on unclamped values.

FIR summation order (`order=`):
  "torchaudio" (default, shipped in csrc/yaapt.hip): what torch's CPU `conv1d` evaluates for this 1-channel
      3-tap kernel — an FMA chain in tap order, acc = b2'*x[t-2]; acc = fma(b1', x[t-1], acc);
      acc = fma(b0', x[t], acc) — measured bit-identical to `F.conv1d` on 80 000 samples for both filters in the
      build container (oneDNN, AVX-512 host; tests/test_oracle_yaapt.py pins it).  The FMA is emulated here
      exactly (round-to-odd in float64) so the oracle gives the same bits on any host.
  "raw_b_then_divide": round 1's order, ((b2*x[t-2] + b1*x[t-1]) + b0*x[t]) / a0 with the raw b; kept as the
      variant of the rounding-order study (tests/golden/make_biquad_order_study.py, fx_biquad_order.json)."""
import math

import numpy as np
import torch

ORDERS = ("torchaudio", "raw_b_then_divide")


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
    """(b0' = b0/a0, b1' = b1/a0, b2' = b2/a0, a0, c1 = a1/a0, c2 = a2/a0) as f32 — what the FIR and the
    recursion use (a0 itself is no longer used by the kernel; kept in the tuple for the plan layout)"""
    b, a = coeffs(kind, sample_rate, cutoff)
    return (np.float32(b[0] / a[0]), np.float32(b[1] / a[0]), np.float32(b[2] / a[0]), a[0],
            np.float32(a[1] / a[0]), np.float32(a[2] / a[0]))


def fma32(a, b, c):
    """exact f32 fused multiply-add on numpy arrays: the f32 x f32 product is exact in float64; the float64 sum
    is rounded to odd (TwoSum error term), which makes the final rounding to f32 the correctly rounded result"""
    p = np.asarray(a, np.float32).astype(np.float64) * np.asarray(b, np.float32).astype(np.float64)
    c64 = np.asarray(c, np.float32).astype(np.float64)
    s = p + c64
    bb = s - p
    e = (p - (s - bb)) + (c64 - bb)
    t = np.nextafter(s, np.where(e > 0, np.inf, -np.inf))
    odd = (s.view(np.int64) & 1) == 1
    s = np.where((e != 0) & ~odd, t, s)
    return s.astype(np.float32)


def fir(x, kind, sample_rate, cutoff, order="torchaudio"):
    """FIR half of lfilter on a 1-D f32 array (2-sample zero history)"""
    x = np.asarray(x, dtype=np.float32)
    xp = np.concatenate([np.zeros(2, np.float32), x])
    x2, x1, x0 = xp[:-2], xp[1:-1], xp[2:]
    if order == "torchaudio":
        b0, b1, b2, _, _, _ = kernel_constants(kind, sample_rate, cutoff)
        return fma32(b0, x0, fma32(b1, x1, b2 * x2))
    if order == "raw_b_then_divide":
        b, a = coeffs(kind, sample_rate, cutoff)
        return ((b[2] * x2 + b[1] * x1) + b[0] * x0) / a[0]
    raise ValueError(order)


#: evaluation orders of the RECURSION half (the study of tests/golden/make_biquad_iir_crosscheck.py; "torchaudio" is shipped):
#:   "torchaudio"  v = f - c2*y2; v = v - c1*y1      the C++ loop as published: multiply, subtract, a2 term first
#:   "c1_first"    v = f - c1*y1; v = v - c2*y2
#:   "fma"         v = fma(-c2, y2, f); v = fma(-c1, y1, v)   the same loop compiled with FMA contraction (-ffp-contract=fast, e.g. aarch64 builds)
#:   "sum_first"   v = f - (c1*y1 + c2*y2)
IIR_ORDERS = ("torchaudio", "c1_first", "fma", "sum_first")


def _fma32_scalar(a, b, c):
    """correctly rounded f32 fma of three python floats holding f32 values (exact product, round-to-odd float64 sum)"""
    import struct
    p = a * b
    s = p + c
    bb = s - p
    e = (p - (s - bb)) + (c - bb)
    if e != 0.0 and not (struct.unpack("<q", struct.pack("<d", s))[0] & 1):
        s = math.nextafter(s, math.inf if e > 0 else -math.inf)
    return float(np.float32(s))


def biquad(x, kind, sample_rate, cutoff, order="torchaudio", iir="torchaudio", clamp=True):
    """x: 1-D float32 numpy array -> filtered, clamped float32 array.  `order`: FIR summation order (ORDERS), `iir`: evaluation
    order of the recursion (IIR_ORDERS)"""
    _, _, _, _, c1, c2 = kernel_constants(kind, sample_rate, cutoff)
    f = fir(x, kind, sample_rate, cutoff, order)
    out = np.empty_like(f)
    if iir == "fma":
        fc1, fc2 = float(c1), float(c2)
        y1 = y2 = 0.0
        fl = f.astype(np.float64).tolist()
        res = []
        for t in range(len(fl)):
            v = _fma32_scalar(-fc2, y2, fl[t])
            v = _fma32_scalar(-fc1, y1, v)
            res.append(v)
            y2 = y1
            y1 = v
        out[:] = np.asarray(res, dtype=np.float32)
    else:
        y1 = np.float32(0)
        y2 = np.float32(0)
        for t in range(f.shape[0]):
            if iir == "torchaudio":
                v = f[t] - c2 * y2
                v = v - c1 * y1
            elif iir == "c1_first":
                v = f[t] - c1 * y1
                v = v - c2 * y2
            elif iir == "sum_first":
                v = f[t] - (c1 * y1 + c2 * y2)
            else:
                raise ValueError(iir)
            out[t] = v
            y2 = y1
            y1 = v
    return np.clip(out, np.float32(-1), np.float32(1)) if clamp else out


def band_limit(x, sample_rate=16000, bp_low=50.0, bp_high=1500.0, order="torchaudio", iir="torchaudio"):
    """SignalObj.filtered_version (yaapt.py:42-51): low-pass at bp_low THEN high-pass at bp_high
    (yes, as written in the reference), each through its own clamped lfilter"""
    return biquad(biquad(x, "lp", sample_rate, bp_low, order, iir), "hp", sample_rate, bp_high, order, iir)
