"""YAAPT pitch tracker, CPU restatement at the configuration the anonymizer uses
(reference: satools/satools/hifigan/yaapt.py:795-951 `_yaapt`/`yaapt`, called with the options
of egs/vc/libritts/local/tuning/hifigan.py:31-36).

Written frame-parallel (the way the HIP kernels are organised) instead of the reference's
per-frame Python loops; every comparison, tie rule and f32 operation order that decides a
candidate follows the reference line cited next to it.  The biquads are third-party
(oracle/biquad.py, parity unpinned)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import biquad

F32 = torch.float32

DEFAULTS = dict(sr=16000.0, frame_length=35.0, tda_frame_length=35.0, frame_space=10.0, f0_min=60.0, f0_max=400.0,
                fft_length=8192.0, bp_low=50.0, bp_high=1500.0, nlfer_thresh1=0.75, nlfer_thresh2=0.1,
                shc_numharms=3.0, shc_window=40.0, shc_maxpeaks=4.0, shc_pwidth=50.0, shc_thresh1=5.0,
                shc_thresh2=1.25, f0_double=150.0, f0_half=150.0, dp5_k1=11.0, nccf_thresh1=0.3,
                nccf_thresh2=0.9, nccf_maxcands=3.0, nccf_pwidth=5.0, merit_boost=0.20, merit_pivot=0.99,
                merit_extra=0.4, median_value=7.0, dp_w1=0.15, dp_w2=0.5, dp_w3=0.1, dp_w4=0.9,
                spec_pitch_min_std=0.05)  # yaapt.py:815-859


class Plan:
    """integer / scalar constants derived from the options (yaapt.py:868-886, :156-157, :190-204,
    :396-406, :686-688)"""

    def __init__(self, n, opts):
        p = dict(DEFAULTS)
        p.update(opts)
        self.p = p
        self.fs = p["sr"]
        self.pad = int(p["frame_length"] / 1000 * int(p["sr"])) // 2
        self.L = n + 2 * self.pad
        self.nfft = int(p["fft_length"])
        self.frame_size = int(math.floor(p["frame_length"] * self.fs / 1000))
        self.frame_jump = int(math.floor(p["frame_space"] * self.fs / 1000))
        assert 15 < self.frame_size < 2048
        half = self.frame_size // 2
        self.nframes = len(range(half, self.L - half, self.frame_jump))
        # nlfer bins (torch f32 round, yaapt.py:156-157)
        self.nl_lo = int(torch.round(torch.tensor(p["f0_min"] * 2 / float(self.fs)) * self.nfft)) - 1
        self.nl_hi = int(torch.round(torch.tensor(p["f0_max"] / float(self.fs)) * self.nfft))
        # spectral track
        self.nframe_size = self.frame_size * 2
        self.delta = self.fs / self.nfft
        wl = math.floor(p["shc_window"] / self.delta)
        self.half_wl = math.floor(float(wl) / 2)
        self.wl = wl + 1 if wl % 2 == 0 else wl
        self.max_shc = math.floor((p["f0_max"] + p["shc_pwidth"] * 2) / self.delta)
        self.min_shc = math.ceil(p["f0_min"] / self.delta)
        self.nharm = int(p["shc_numharms"])
        self.maxpeaks = int(p["shc_maxpeaks"])
        # peaks()
        w = math.floor(p["shc_pwidth"] / self.delta)
        self.pk_width = w + 1 if w % 2 == 0 else w
        self.pk_center = math.ceil(self.pk_width / 2)
        self.pk_min_lag = max(1, math.floor(p["f0_min"] / self.delta - self.pk_center))
        self.pk_max_lag = min(math.floor(p["f0_max"] / self.delta + self.pk_center), self.max_shc - self.pk_width)
        # time track
        self.tda_len = int(p["tda_frame_length"] * self.fs / 1000)
        self.tda_nframes = min(int((self.L - (self.tda_len - self.frame_jump)) / self.frame_jump), self.nframes)
        self.maxcands = int(p["nccf_maxcands"])
        self.nccf_center = math.floor(p["nccf_pwidth"] / 2.0)


def medfilt(x, k):
    """yaapt.py:54-69: zero-padded sliding median (odd k -> the middle order statistic)"""
    pad = k // 2
    return F.pad(x, (pad, pad)).unfold(0, k, 1).median(dim=-1)[0]


# --------------------------------------------------------------------------------------------
def nlfer(filt, plan):
    """yaapt.py:148-176 -> (energy [nframes] normalised by its mean, vuv)"""
    win = torch.hann_window(plan.frame_size + 2)[1:-1]
    idx = (torch.arange(plan.nframes) * plan.frame_jump).unsqueeze(1) + torch.arange(plan.frame_size).unsqueeze(0)
    fr = filt[idx] * win
    spec = torch.fft.rfft(fr, plan.nfft)
    e = torch.abs(spec[:, plan.nl_lo:plan.nl_hi]).sum(1).to(F32)
    energy = e / torch.mean(e)
    return energy, energy > plan.p["nlfer_thresh1"]


def shc_frames(filt2, vuv, plan):
    """spectral harmonics correlation of every voiced frame (yaapt.py:209-235) -> SHC [nv, max_shc]"""
    need = plan.nframe_size + (plan.nframes - 1) * plan.frame_jump - plan.L
    data = torch.cat((filt2, torch.zeros(need, dtype=F32)))
    win = torch.kaiser_window(plan.nframe_size, periodic=True, beta=0.5)
    frames = torch.where(vuv)[0]
    idx = (frames * plan.frame_jump).unsqueeze(1) + torch.arange(plan.nframe_size).unsqueeze(0)
    sl = data[idx] * win
    sl = sl - sl.mean(dim=1, keepdim=True)
    mag = torch.zeros(frames.numel(), plan.half_wl + plan.nfft // 2 + 1)
    mag[:, plan.half_wl:] = torch.abs(torch.fft.rfft(sl, plan.nfft))
    rows = plan.max_shc - plan.min_shc + 1
    r = torch.arange(rows).unsqueeze(1)
    w = torch.arange(plan.wl).unsqueeze(0)
    prod = torch.ones(frames.numel(), rows, plan.wl)
    for h in range(plan.nharm + 1):
        prod = prod * mag[:, plan.min_shc * (h + 1) + r * (h + 1) + w]
    shc = torch.zeros(frames.numel(), plan.max_shc)
    shc[:, plan.min_shc - 1:plan.max_shc] = prod.sum(2)
    return frames, shc


def peaks(shc, plan):
    """yaapt.py:383-497 on one SHC vector -> (pitch[4], merit[4])"""
    p = plan.p
    mp = plan.maxpeaks
    zeros, ones = torch.zeros(mp), torch.ones(mp)
    lo, hi, c = plan.pk_min_lag, plan.pk_max_lag, plan.pk_center
    data = shc
    mx = torch.max(data[lo:hi + 1])
    if mx > 1e-14:
        data = data / mx
    avg = torch.mean(data[lo:hi + 1])
    if avg > 1 / p["shc_thresh1"]:
        return zeros, ones
    mid = data[lo + c + 1:hi - c + 1]
    flag = (mid > data[lo + c:hi - c]) & (mid > data[lo + c + 2:hi - c + 2]) & (mid > p["shc_thresh2"] * avg)
    pitch, merit = [], []
    for n in (flag.nonzero().flatten() + lo + c + 1).tolist():
        if int(torch.argmax(data[n - c:n + c + 1])) == c:
            pitch.append(float(n) * plan.delta)
            merit.append(float(data[n]))
    numpeaks = len(pitch)
    pt = torch.tensor(pitch + [0.0] * max(0, mp - numpeaks), dtype=F32)
    mt = torch.tensor(merit + [0.0] * max(0, mp - numpeaks), dtype=F32)
    if torch.max(mt) / avg < p["shc_thresh1"]:
        return zeros, ones
    order = (-mt).argsort()
    mt, pt = mt[order], pt[order]
    numpeaks = min(numpeaks, mp)
    pt = torch.cat((pt[:numpeaks], torch.zeros(mp - numpeaks)))
    mt = torch.cat((mt[:numpeaks], torch.zeros(mp - numpeaks)))
    if numpeaks > 0:
        if pt[0] > p["f0_double"]:
            numpeaks = min(numpeaks + 1, mp)
            pt[numpeaks - 1] = pt[0] / 2.0
            mt[numpeaks - 1] = p["merit_extra"]
        if pt[0] < p["f0_half"]:
            numpeaks = min(numpeaks + 1, mp)
            pt[numpeaks - 1] = pt[0] * 2.0
            mt[numpeaks - 1] = p["merit_extra"]
        if numpeaks < mp:
            pt[numpeaks:mp] = pt[0]
            mt[numpeaks:mp] = mt[0]
        return pt, mt
    return zeros, ones


def path1(local, trans):
    """yaapt.py:530-570.  local [C, T], trans [C, C, T].  aux[i][j] = PCOST[j] + trans[i][j][t]
    (the 1-D PCOST broadcasts over the LAST axis); ties go to the LAST minimum (argmin on a flip)."""
    C, T = local.shape
    pred = torch.zeros((C, T), dtype=torch.long)
    p_small = torch.zeros(T, dtype=torch.long)
    pcost = local[:, 0].clone()
    ar = torch.arange(C)
    for t in range(1, T):
        aux = pcost + trans[:, :, t]
        k = C - torch.argmin(torch.flip(aux, [1]), 1) - 1
        pred[:, t] = k
        ccost = pcost[k] + trans[k, ar, t]
        ccost = ccost + local[:, t]
        pcost = ccost
        p_small[t] = C - torch.argmin(torch.flip(ccost, dims=[0]), dim=0) - 1
    path = torch.ones(T, dtype=torch.long)
    path[-1] = p_small[-1]
    for t in range(T - 2, -1, -1):
        path[t] = pred[path[t + 1], t + 1]
    return path


def dynamic5(pitch, merit, k1, f0_min):
    """yaapt.py:506-523"""
    C, T = pitch.shape
    local = 1 - merit
    trans = torch.zeros((C, C, T))
    trans[:, :, 1:] = abs(pitch[:, 1:].reshape(1, C, T - 1) - pitch[:, :-1].reshape(C, 1, T - 1)) / f0_min
    trans[:, :, 1:] = 0.05 * trans[:, :, 1:] + trans[:, :, 1:] ** 2
    trans = k1 * trans
    path = path1(local, trans)
    return pitch[path, torch.arange(T)]


def spec_track(filt2, energy, vuv, plan, aux=None):
    """yaapt.py:184-312 -> (spec_pitch [nframes], pitch_std)"""
    p = plan.p
    nf = plan.nframes
    cand_pitch = torch.zeros((plan.maxpeaks, nf))
    cand_merit = torch.ones((plan.maxpeaks, nf))
    frames, shc = shc_frames(filt2, vuv, plan)
    for i, f in enumerate(frames.tolist()):
        cand_pitch[:, f], cand_merit[:, f] = peaks(shc[i], plan)
    if aux is not None:
        aux.update(cand_pitch=cand_pitch.clone(), cand_merit=cand_merit.clone())
    spec_pitch = cand_pitch[0, :].clone()
    voiced = cand_pitch[0, :] > 0.0
    vcp, vcm = cand_pitch[:, voiced].clone(), cand_merit[:, voiced].clone()
    nv = vcp.shape[1]
    avg_v, std_v = torch.mean(vcp[0, :]), torch.std(vcp[0, :])
    d1 = abs(vcp - 0.8 * avg_v) * (3 - vcm)
    index = d1.argmin(0)
    ar = torch.arange(nv)
    pk, mr = vcp[index, ar], vcm[index, ar]
    med_k = max(1, int(p["median_value"]) - 2)
    pk = medfilt(pk, med_k) if nv > 0 else pk
    vcp[index, ar] = pk
    vcm[index, ar] = mr
    wtrans = p["dp5_k1"] * std_v / avg_v
    first_cleared = False
    if nv > 2:
        vpitch = medfilt(dynamic5(vcp, vcm, wtrans, p["f0_min"]), med_k)
    elif nv > 0:
        vpitch = torch.ones(nv) * 150.0
    else:
        vpitch = torch.tensor([150.0])
        first_cleared = True
    pitch_avg = torch.mean(vpitch)
    pitch_std = torch.maximum(torch.std(vpitch), pitch_avg * torch.tensor(p["spec_pitch_min_std"]))
    if not first_cleared:
        spec_pitch[voiced] = vpitch
    if spec_pitch[0] < pitch_avg / 2:
        spec_pitch[0] = pitch_avg
    if spec_pitch[-1] < pitch_avg / 2:
        spec_pitch[-1] = pitch_avg
    nz = spec_pitch[torch.nonzero(spec_pitch).squeeze()]
    spec_pitch = F.interpolate(nz.unsqueeze(0).unsqueeze(0), size=(nf,), mode="linear").squeeze()
    spec_pitch[0] = spec_pitch[2]
    spec_pitch[1] = spec_pitch[3]
    return spec_pitch, pitch_std


def frame_means(filt, plan):
    """time_track subtracts each 400-sample frame's mean IN PLACE on overlapping views
    (yaapt.py:711-714 + crs_corr :589): frame k's first 80 samples already carry frame k-1's
    subtraction.  Returns the de-meaned frames [nframes, 400] exactly as crs_corr sees them."""
    T, n, hop = plan.tda_nframes, plan.tda_len, plan.frame_jump
    ov = n - hop
    out = torch.empty(T, n)
    prev_mean = None
    for k in range(T):
        cur = filt[k * hop:k * hop + n].clone()
        if prev_mean is not None:
            cur[:ov] = cur[:ov] - prev_mean
        m = torch.mean(cur)
        out[k] = cur - m
        prev_mean = m
    return out


def nccf_candidates(frames, lag_min, lag_max, plan):
    """crs_corr + cmp_rate (yaapt.py:577-673) for one de-meaned frame -> (pitch, merit) of the at
    most one candidate the reference's cmp_rate can return"""
    p = plan.p
    n = frames.numel()
    N = n - lag_max
    assert N > 0
    x = frames[:N]
    pw = torch.dot(x, x)
    rows = frames[lag_min:lag_max + N].unfold(0, N, 1)[:lag_max - lag_min]
    phi = torch.zeros(n)
    nume = torch.matmul(rows, x.unsqueeze(0).T).squeeze()          # [lags, N] @ [N, 1] like the reference
    phi[lag_min:lag_max] = nume / torch.sqrt(torch.sum(rows * rows, 1) * pw + 0.0)
    c = plan.nccf_center
    mid = phi[lag_min + c:lag_max - c + 1]
    flag = (mid > phi[lag_min + c - 1:lag_max - c]) & (mid > phi[lag_min + c + 1:lag_max - c + 2]) & \
           (mid > p["nccf_thresh1"])
    nzs = flag.nonzero()
    if nzs.shape[0] == 0:
        return 0.0, 0.0
    npk = int(nzs[0]) + lag_min + c
    if torch.amax(phi) > p["nccf_thresh2"]:
        return plan.fs / float(npk + 1), float(phi[npk])
    if int(torch.argmax(phi[npk - c:npk + c + 1])) == c:
        return plan.fs / float(npk + 1), float(phi[npk])
    return 0.0, 0.0


def time_track(filt, spec_pitch, pitch_std, plan):
    """yaapt.py:680-729 -> (time_pitch [3, T], time_merit [3, T])"""
    p = plan.p
    T = plan.tda_nframes
    sp = spec_pitch[:T]
    freq_thresh = 5.0 * pitch_std
    lo = torch.max(sp - 2.0 * pitch_std, torch.tensor(p["f0_min"]))
    hi = torch.min(sp + 2.0 * pitch_std, torch.tensor(p["f0_max"]))
    tp = torch.zeros((plan.maxcands, T))
    tm = torch.zeros((plan.maxcands, T))
    frames = frame_means(filt, plan)
    a = torch.floor(plan.fs / hi)
    b = torch.floor(plan.fs / lo)
    for k in range(T):
        if not math.isnan(float(a[k])) and not math.isnan(float(b[k])):
            lag_min = int(a[k]) - plan.nccf_center
            lag_max = int(b[k]) + plan.nccf_center
            pit, mer = nccf_candidates(frames[k], lag_min, lag_max, plan)
            tp[0, k], tm[0, k] = pit, mer
    mx = torch.amax(tm, 0)
    tm = torch.where(mx > 1.0, tm / mx, tm)
    diff = torch.abs(tp - sp)
    match = (1 - diff / freq_thresh) * (diff < freq_thresh)
    tm = ((1 + p["merit_boost"]) * tm) * match
    return tp, tm


def refine(tp1, tm1, tp2, tm2, spec_pitch, energy, vuv, plan):
    """yaapt.py:732-784"""
    p = plan.p
    nf = plan.nframes
    tp = torch.cat((tp1, tp2), 0)
    tm = torch.cat((tm1, tm2), 0)
    C = tp.shape[0]
    idx = torch.argsort(-tm, dim=0)
    tm = torch.flip(torch.sort(tm, dim=0)[0], dims=[0])
    tp = tp[idx, torch.arange(nf)]
    best = medfilt(tp[0, :], int(p["median_value"])) * vuv
    i1 = energy <= p["nlfer_thresh2"]
    i2 = (energy > p["nlfer_thresh2"]) & (tp[0, :] > 0)
    i3 = (energy > p["nlfer_thresh2"]) & (tp[0, :] <= 0)
    mm = (tp[1:C - 1, :] == 0) & i2
    mm = torch.cat((torch.zeros((1, nf), dtype=torch.bool), mm, torch.zeros((1, nf), dtype=torch.bool)), 0)
    tp[:, i1] = 0
    tm[:, i1] = p["merit_pivot"]
    tp[C - 1, i2] = 0.0
    tm[C - 1, i2] = 1.0 - tm[0, i2]
    tm[mm] = 0.0
    tp[0, i3] = spec_pitch[i3]
    tm[0, i3] = torch.minimum(torch.tensor(1), energy[i3] / 2.0)
    tp[1:C, i3] = 0.0
    tm[1:C, i3] = 1.0 - tm[0, i3]
    tp[C - 2, :] = best
    nzf = best > 0.0
    tm[C - 2, nzf] = tm[0, nzf]
    tm[C - 2, ~nzf] = 1.0 - torch.minimum(torch.tensor(1), energy[~nzf] / 2.0)
    tp[C - 3, :] = spec_pitch
    tm[C - 3, :] = energy / 5.0
    return tp, tm


def dynamic(rp, rm, energy, plan):
    """yaapt.py:321-370"""
    p = plan.p
    C, T = rp.shape
    best = rp[C - 2, :]
    mean_pitch = torch.mean(best[best > 0])
    local = 1 - rm
    trans = torch.ones((C, C, T))
    m1 = torch.zeros((C, C, T))
    m2 = torch.zeros((C, C, T))
    m1[:, :, 1:] = rp[:, 1:].reshape(1, C, T - 1).expand(C, C, T - 1)
    m2[:, :, 1:] = rp[:, :-1].reshape(C, 1, T - 1).expand(C, C, T - 1)
    i1 = torch.zeros((C, C, T), dtype=torch.bool)
    i2 = torch.zeros_like(i1)
    i3 = torch.zeros_like(i1)
    i1[:, :, 1:] = (m1[:, :, 1:] > 0) & (m2[:, :, 1:] > 0)
    i2[:, :, 1:] = ((m1[:, :, 1:] == 0) & (m2[:, :, 1:] > 0)) | ((m1[:, :, 1:] > 0) & (m2[:, :, 1:] == 0))
    i3[:, :, 1:] = (m1[:, :, 1:] == 0) & (m2[:, :, 1:] == 0)
    v1 = torch.abs(m1 - m2) / mean_pitch
    b2 = torch.cat((torch.tensor([0]), torch.minimum(torch.tensor(1), torch.abs(energy[:-1] - energy[1:]))))
    b2 = b2.repeat(C * C).reshape(C, C, -1)
    trans[i1] = p["dp_w1"] * v1[i1]
    trans[i2] = p["dp_w2"] * (1 - b2[i2])
    trans[i3] = p["dp_w3"]
    trans = trans / p["dp_w4"]
    path = path1(local, trans)
    return rp[path, torch.arange(T)]


def yaapt_one(x, opts, aux=None, biquad_order="torchaudio", biquad_iir="torchaudio"):
    """x [n] f32 -> final pitch [nframes] (Hz, 0 = unvoiced)   (`_yaapt`, yaapt.py:795-944).
    `biquad_order`: FIR summation order of the third-party biquads (oracle/biquad.py)"""
    plan = Plan(x.numel(), opts)
    sig = F.pad(x.to(F32), (plan.pad, plan.pad))
    bl = lambda v: torch.from_numpy(biquad.band_limit(v.numpy(), int(plan.fs), plan.p["bp_low"], plan.p["bp_high"], biquad_order, biquad_iir))
    filt = bl(sig)
    filt2 = bl(sig ** 2)
    energy, vuv = nlfer(filt, plan)
    spec_pitch, pitch_std = spec_track(filt2, energy, vuv, plan, aux=aux)
    tp1, tm1 = time_track(filt, spec_pitch, pitch_std, plan)
    tp2, tm2 = time_track(filt2, spec_pitch, pitch_std, plan)
    if tp1.shape[1] < spec_pitch.numel():
        padn = spec_pitch.numel() - tp1.shape[1]
        tp1, tp2, tm1, tm2 = [torch.cat((t, torch.zeros((3, padn))), 1) for t in (tp1, tp2, tm1, tm2)]
    rp, rm = refine(tp1, tm1, tp2, tm2, spec_pitch, energy, vuv, plan)
    final = dynamic(rp, rm, energy, plan)
    if aux is not None:
        aux.update(filt=filt, filt2=filt2, energy=energy, vuv=vuv, spec_pitch=spec_pitch, pitch_std=pitch_std,
                   tp1=tp1, tm1=tm1, tp2=tp2, tm2=tm2, ref_pitch=rp, ref_merit=rm, final=final)
    return final


def yaapt(wav, opts, biquad_order="torchaudio"):
    """wav [B, n] -> [B, nframes]   (`yaapt`, yaapt.py:946-951: a serial loop over the batch)"""
    return torch.stack([yaapt_one(wav[i], opts, biquad_order=biquad_order) for i in range(wav.shape[0])], 0)
