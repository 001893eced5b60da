"""Kaldi-compatible fbank, CPU restatement (reference: satools/satools/kaldifeature.py:461-593,
helpers :80-122, :200-264, :386-457), for the call made on the hot path:
fbank(x, num_mel_bins=80, snip_edges=False) with every other argument at its default."""
import math

import torch


def povey_window(n=400):
    # kaldifeature.py:144-146
    return torch.hann_window(n, periodic=False, dtype=torch.float32).pow(0.85)


def mel_banks(num_bins=80, n_fft=512, sample_freq=16000.0, low_freq=20.0, high_freq=0.0):
    # kaldifeature.py:386-457 (vtln_warp == 1 branch) + zero column appended at :576
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    width = sample_freq / n_fft
    mel_lo = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_hi = 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = (mel_hi - mel_lo) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left, center, right = mel_lo + b * delta, mel_lo + (b + 1.0) * delta, mel_lo + (b + 2.0) * delta
    mel = (1127.0 * (1.0 + (width * torch.arange(n_fft / 2)) / 700.0).log()).unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    bins = torch.max(torch.zeros(1), torch.min(up, down))
    return torch.nn.functional.pad(bins, (0, 1))


def frames(wave, win=400, shift=160):
    """snip_edges=False framing (kaldifeature.py:104-122): m = (n + shift/2)//shift frames over
    [reversed first 120 samples | wave | whole wave reversed]"""
    n = wave.numel()
    m = (n + shift // 2) // shift
    pad = win // 2 - shift // 2
    rev = torch.flip(wave, [0])
    padded = torch.cat((rev[-pad:], wave, rev), 0)
    idx = (torch.arange(m) * shift).unsqueeze(1) + torch.arange(win).unsqueeze(0)
    return padded[idx]


def fbank(x, num_mel_bins=80):
    """x [B, n] (already scaled by 32768) -> [B, m, num_mel_bins]"""
    assert x.dim() == 2 and x.shape[1] >= 400, "choose a window size 400 that is [2, {}]".format(x.shape[1])
    B = x.shape[0]
    fr = torch.cat([frames(w) for w in x], 0)                       # [B*m, 400]
    fr = fr - fr.mean(dim=1, keepdim=True)                           # remove_dc_offset
    prev = torch.cat((fr[:, :1], fr[:, :-1]), 1)                     # replicate left neighbour
    fr = fr - 0.97 * prev                                            # pre-emphasis
    fr = fr * povey_window().unsqueeze(0)
    fr = torch.nn.functional.pad(fr, (0, 112))                       # 400 -> 512
    spec = torch.fft.rfft(fr).abs().pow(2.0)                         # power spectrum [.., 257]
    mel = torch.mm(spec, mel_banks(num_mel_bins).T)
    mel = torch.max(mel, torch.tensor(1e-6)).log()
    return mel.view(B, -1, num_mel_bins)
