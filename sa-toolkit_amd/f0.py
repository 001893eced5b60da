"""YAAPT F0 extraction on the HIP device behind the reference's `yaapt(wav, opts)` call
(satools/satools/hifigan/yaapt.py:946-951; options as passed by
egs/vc/libritts/local/tuning/hifigan.py:31-36).

The host derives the integer/scalar plan exactly the way the reference derives it from its
option dict (Python float arithmetic, math.floor/ceil, torch f32 rounds where the reference uses
them) and builds the three tables (hann, kaiser, FFT twiddles); everything else runs in
sat_yaapt_f32.  Unlike the reference (serial over the batch, forced to the CPU, yaapt.py:798-799)
the whole batch is processed in one call on the GPU."""
import ctypes as C
import functools
import math
import threading

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream

DEFAULTS = dict(sr=16000.0, frame_length=35.0, tda_frame_length=35.0, frame_space=10.0, f0_min=60.0, f0_max=400.0,
                fft_length=8192.0, bp_low=50.0, bp_high=1500.0, nlfer_thresh1=0.75, nlfer_thresh2=0.1,
                shc_numharms=3.0, shc_window=40.0, shc_maxpeaks=4.0, shc_pwidth=50.0, shc_thresh1=5.0,
                shc_thresh2=1.25, f0_double=150.0, f0_half=150.0, dp5_k1=11.0, nccf_thresh1=0.3,
                nccf_thresh2=0.9, nccf_maxcands=3.0, nccf_pwidth=5.0, merit_boost=0.20, merit_pivot=0.99,
                merit_extra=0.4, median_value=7.0, dp_w1=0.15, dp_w2=0.5, dp_w3=0.1, dp_w4=0.9,
                spec_pitch_min_std=0.05)  # yaapt.py:815-859


class YaaptPlan(C.Structure):
    """mirror of sat_yaapt_plan (include/satools_hip.h)"""
    _fields_ = [(k, C.c_int32) for k in (
        "n", "pad", "L", "Lz", "nfft", "frame_size", "frame_jump", "nframes", "nl_lo", "nl_hi", "nframe_size",
        "half_wl", "wl", "max_shc", "min_shc", "nharm", "maxpeaks", "pk_center", "pk_min_lag", "pk_max_lag",
        "tda_len", "tda_nframes", "maxcands", "nccf_center", "median_value")] + [(k, C.c_float) for k in (
        "fs", "delta", "nlfer_thresh1", "nlfer_thresh2", "shc_thresh1", "inv_shc_thresh1", "shc_thresh2",
        "f0_double", "f0_half", "merit_extra", "dp5_k1", "f0_min", "f0_max", "spec_pitch_min_std",
        "nccf_thresh1", "nccf_thresh2", "merit_boost1", "merit_pivot", "dp_w1", "dp_w2", "dp_w3", "dp_w4")] + [
        ("lp", C.c_float * 6), ("hp", C.c_float * 6)]


@functools.lru_cache(maxsize=64)
def biquad_constants(kind, sample_rate, cutoff, Q=0.707):
    """RBJ biquad as torchaudio.functional.{lowpass,highpass}_biquad computes it (f32 tensors), then
    the normalisation `_lfilter` applies FIRST to both coefficient vectors: {b0/a0, b1/a0, b2/a0, a0, a1/a0, a2/a0}
    (f32 divisions; a0 itself is unused by the kernel).  torchaudio is third-party to the reference
    (yaapt.py:46-47) and absent here: restated from its published algorithm, the FIR order a tested decision
    (oracle/biquad.py, tests/golden/fx_biquad_order.json)."""
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
    v = np.array([float(b0), float(b1), float(b0), float(a0), float(a1), float(a2)], dtype=np.float32)
    return [float(np.float32(v[0] / v[3])), float(np.float32(v[1] / v[3])), float(np.float32(v[2] / v[3])), float(v[3]),
            float(np.float32(v[4] / v[3])), float(np.float32(v[5] / v[3]))]


def make_plan(n, opts):
    p = dict(DEFAULTS)
    if "frame_lengtht" in opts and "tda_frame_length" not in opts:   # yaapt.py:802-810
        opts = dict(opts)
        opts["tda_frame_length"] = opts.pop("frame_lengtht")
    p.update(opts)
    fs = p["sr"]
    P = YaaptPlan()
    P.n = n
    P.pad = int(p["frame_length"] / 1000 * int(p["sr"])) // 2                       # yaapt.py:868
    P.L = n + 2 * P.pad
    P.nfft = int(p["fft_length"])
    P.frame_size = int(math.floor(p["frame_length"] * fs / 1000))                  # :882
    P.frame_jump = int(math.floor(p["frame_space"] * fs / 1000))                   # :883
    half = P.frame_size // 2
    P.nframes = len(range(half, P.L - half, P.frame_jump))                         # nlfer :163-165
    P.nl_lo = int(torch.round(torch.tensor(p["f0_min"] * 2 / float(fs)) * P.nfft)) - 1   # :156, slice start N-1
    P.nl_hi = int(torch.round(torch.tensor(p["f0_max"] / float(fs)) * P.nfft))          # :157
    P.nframe_size = P.frame_size * 2                                               # spec_track :189
    delta = fs / P.nfft
    wl = math.floor(p["shc_window"] / delta)
    P.half_wl = math.floor(float(wl) / 2)
    P.wl = wl + 1 if wl % 2 == 0 else wl
    P.max_shc = math.floor((p["f0_max"] + p["shc_pwidth"] * 2) / delta)
    P.min_shc = math.ceil(p["f0_min"] / delta)
    P.nharm = int(p["shc_numharms"])
    P.maxpeaks = int(p["shc_maxpeaks"])
    w = math.floor(p["shc_pwidth"] / delta)                                        # peaks :396-406
    width = w + 1 if w % 2 == 0 else w
    P.pk_center = math.ceil(width / 2)
    P.pk_min_lag = max(1, math.floor(p["f0_min"] / delta - P.pk_center))
    P.pk_max_lag = min(math.floor(p["f0_max"] / delta + P.pk_center), P.max_shc - width)
    P.tda_len = int(p["tda_frame_length"] * fs / 1000)                             # time_track :686-694
    P.tda_nframes = min(int((P.L - (P.tda_len - P.frame_jump)) / P.frame_jump), P.nframes)
    P.maxcands = int(p["nccf_maxcands"])
    P.nccf_center = math.floor(p["nccf_pwidth"] / 2.0)
    P.median_value = int(p["median_value"])
    need = P.nframe_size + (P.nframes - 1) * P.frame_jump                          # zero extension, :206-210
    P.Lz = max(P.L, need)
    P.fs, P.delta = fs, delta
    for k in ("nlfer_thresh1", "nlfer_thresh2", "shc_thresh1", "shc_thresh2", "f0_double", "f0_half", "merit_extra",
              "dp5_k1", "f0_min", "f0_max", "spec_pitch_min_std", "nccf_thresh1", "nccf_thresh2", "merit_pivot",
              "dp_w1", "dp_w2", "dp_w3", "dp_w4"):
        setattr(P, k, p[k])
    P.inv_shc_thresh1 = 1 / p["shc_thresh1"]
    P.merit_boost1 = 1 + p["merit_boost"]
    lp = biquad_constants("lp", int(fs), p["bp_low"])
    hp = biquad_constants("hp", int(fs), p["bp_high"])
    for i in range(6):
        P.lp[i], P.hp[i] = lp[i], hp[i]
    return P


def length_dims(P, n):
    """[n, L, nframes, tda_nframes] of make_plan(n, opts) from a plan P of the same options at another length: the only fields of a
    plan that depend on the length (a ragged batch of 32 lengths was 32 full plans, 4 ms of host time per batch in the batch job's
    launching thread; tests/test_host_logic.py holds the two against each other)"""
    L = n + 2 * P.pad
    half = P.frame_size // 2
    nframes = len(range(half, L - half, P.frame_jump))
    return [n, L, nframes, min(int((L - (P.tda_len - P.frame_jump)) / P.frame_jump), nframes)]


_tables = {}


def _get_tables(P, device):
    key = (P.frame_size, str(device))
    if key not in _tables:
        hann = torch.hann_window(P.frame_size + 2)[1:-1].contiguous()              # nlfer :159
        kaiser = torch.kaiser_window(P.nframe_size, periodic=True, beta=0.5)       # spec_track :213
        k = np.arange(4096, dtype=np.float64)
        tw = np.stack([np.cos(2 * np.pi * k / 8192.0), -np.sin(2 * np.pi * k / 8192.0)], 1).astype(np.float32)
        _tables[key] = (hann.to(device), kaiser.to(device), torch.from_numpy(tw).to(device))
    return _tables[key]


class _PinnedInts:
    """rows of ONE page-locked int32 block for the words that travel with a yaapt launch (its status, the per-utterance dimensions of a
    ragged batch).  A `tensor.pin_memory()` per call asks torch's host allocator for a block whose previous copy has completed; a
    launching thread that runs ahead of the GPU (the batch job with deferred status) finds none and pays a hipHostMalloc per launch,
    ~50 ms each at the start of a job (0.67 s for the first four batches)."""
    ROWS, COLS = 64, 1024

    def __init__(self):
        self.buf, self.free, self.lock = None, list(range(self.ROWS)), threading.Lock()

    def take(self, n):
        """-> (pinned int32 view of n words, row to give back or None)"""
        with self.lock:
            if self.buf is None:
                self.buf = torch.empty(self.ROWS, self.COLS, dtype=torch.int32, pin_memory=True)
            row = self.free.pop() if (self.free and n <= self.COLS) else None
        if row is None:
            return torch.empty(n, dtype=torch.int32).pin_memory(), None
        return self.buf[row, :n], row

    def give(self, row):
        if row is not None:
            with self.lock:
                self.free.append(row)


_pinned_ints = _PinnedInts()


class F0Status:
    """deferred error report of one yaapt launch (checked without stalling the launch stream); `also` = rows of _pinned_ints that the
    launch read (a ragged batch's dimensions), free once the status has arrived"""

    def __init__(self, dev_status, B, also=()):
        self.host, row = _pinned_ints.take(B)
        self.rows = [row] + list(also)
        self.host.copy_(dev_status, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def _release(self):
        rows, self.rows = self.rows, []
        for r in rows:
            _pinned_ints.give(r)

    def __del__(self):
        try:
            if self.rows:
                self.event.synchronize()        # (never checked: the copies must have landed before the rows are reused)
                self._release()
        except Exception:
            pass

    def check(self):
        if self.host is not None:
            self.event.synchronize()
            self.words, self.host = self.host.clone(), None
            self._release()
        st = self.words
        if (st == 1).any():
            # the reference fails inside medfilt/unfold when no frame of an utterance is voiced
            raise RuntimeError("yaapt: no voiced frame in utterance(s) %s (the reference's spec_track fails on an "
                               "empty candidate list)" % (st == 1).nonzero().flatten().tolist())
        if (st == 2).any():
            raise AssertionError("ERROR: Negative index in the cross correlation calculation of the pYAAPT time domain "
                                 "analysis. Please try to increase the value of the \"tda_frame_length\" parameter.")


def workspace_views(ws, P, B):
    """named views into the sat_yaapt_f32 workspace (layout of csrc/yaapt.hip), for diagnostics/tests"""
    nf, Lz = P.nframes, P.Lz
    out, off = {}, 0
    for name, shape, dt in (("filt", (B, 2, Lz), torch.float32), ("e_raw", (B, nf), torch.float32),
                            ("energy", (B, nf), torch.float32), ("vuv", (B, nf), torch.int32),
                            ("cand", (B, 8, nf), torch.float32), ("spec_pitch", (B, nf), torch.float32),
                            ("scal", (B, 4), torch.float32), ("fmean", (B, 2, nf), torch.float32),
                            ("tp", (B, 2, nf), torch.float32), ("tm", (B, 2, nf), torch.float32)):
        cnt = int(np.prod(shape))
        out[name] = ws[off:off + cnt].view(dt).view(*shape)
        off += cnt
    return out


def yaapt_ragged(wav, lengths, opts, defer_status=False):
    """wav [B, n_max] zero-padded on the HIP device, lengths [B] -> F0 [B, nframes(n_max)]: every utterance tracked
    at its own length (zero past its frames), all in one launch sequence"""
    if not wav.is_cuda or wav.dim() != 2:
        raise _lib.SatError("yaapt_ragged expects [B, samples] on the HIP device")
    wav = wav.to(torch.float32).contiguous()
    B, n_max = wav.shape
    lens = [int(v) for v in lengths]
    if len(lens) != B or max(lens) > n_max or min(lens) <= 0:
        raise _lib.SatError("yaapt_ragged: lengths do not fit the batch")
    P = make_plan(n_max, dict(opts))
    dims = [length_dims(P, n) for n in lens]
    if max(d[2] for d in dims) > P.nframes or max(d[1] for d in dims) > P.Lz:
        raise _lib.SatError("yaapt_ragged: an utterance exceeds the batch plan")
    hann, kaiser, tw = _get_tables(P, wav.device)
    ws_bytes = lib().sat_yaapt_workspace_bytes(C.byref(P), B)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=wav.device)
    f0 = torch.empty(B, P.nframes, dtype=torch.float32, device=wav.device)
    status = torch.empty(B, dtype=torch.int32, device=wav.device)
    ud_host, ud_row = _pinned_ints.take(4 * B)
    ud_host.copy_(torch.tensor(dims, dtype=torch.int32).reshape(-1))
    ud = ud_host.to(wav.device, non_blocking=True).view(B, 4)
    check(lib().sat_yaapt_ragged_f32(C.byref(P), ptr(wav), ptr(ud), ptr(f0), ptr(status), ptr(hann), ptr(kaiser), ptr(tw),
                                     ptr(ws), ws_bytes, B, stream()), "sat_yaapt_ragged_f32")
    st = F0Status(status, B, also=(ud_row,))
    if defer_status:
        return f0, st
    st.check()
    return f0


def yaapt(wav, opts, defer_status=False, return_aux=False):
    """wav [B, n] on the HIP device -> F0 [B, nframes] on the same device"""
    if not wav.is_cuda:
        raise _lib.SatError("yaapt runs on the HIP device only (no CPU fallback); move the input to 'cuda'")
    if wav.dim() != 2:
        raise _lib.SatError("yaapt expects [B, samples]")
    wav = wav.to(torch.float32).contiguous()
    B, n = wav.shape
    P = make_plan(n, dict(opts))
    hann, kaiser, tw = _get_tables(P, wav.device)
    ws_bytes = lib().sat_yaapt_workspace_bytes(C.byref(P), B)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=wav.device)
    f0 = torch.empty(B, P.nframes, dtype=torch.float32, device=wav.device)
    status = torch.empty(B, dtype=torch.int32, device=wav.device)
    check(lib().sat_yaapt_f32(C.byref(P), ptr(wav), ptr(f0), ptr(status), ptr(hann), ptr(kaiser), ptr(tw), ptr(ws),
                              ws_bytes, B, stream()), "sat_yaapt_f32")
    st = F0Status(status, B)
    if defer_status:
        return f0, st
    st.check()
    if return_aux:
        return f0, workspace_views(ws, P, B)
    return f0
