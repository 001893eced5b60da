"""torchaudio.transforms.MelSpectrogram as the reference configures it (satools/satools/sidekit/preprocessor.py:
164-236: sample_rate 16000, n_fft 1024, win_length 400, hop_length 160, hann window, power 2, 80 mel, 90-7600 Hz),
CPU restatement.

THIRD-PARTY, PARITY UNPINNED: torchaudio is not under /root/reference and not installed here.  This restates the
published torchaudio 2.1 defaults: Spectrogram(center=True, pad_mode="reflect", normalized=False, onesided=True,
window = hann_window(win_length, periodic=True) zero-padded to n_fft by torch.stft) -> |X|^power;
MelScale(norm=None, mel_scale="htk") with melscale_fbanks over n_fft//2+1 linear frequencies.
It doubles as the MelSpectrogram of the fixture generator's torchaudio stand-in."""
import math

import torch


def hz_to_mel_htk(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks(n_freqs=513, f_min=90.0, f_max=7600.0, n_mels=80, sample_rate=16000):
    """[n_freqs, n_mels] triangular filters, htk scale, no normalisation"""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def spectrogram(x, window, n_fft=1024, hop=160, win_length=400, power=2.0):
    """x [..., n] -> [..., n_fft//2+1, 1 + n//hop]"""
    shape = x.shape
    x = x.reshape(-1, shape[-1])
    s = torch.stft(x, n_fft=n_fft, hop_length=hop, win_length=win_length, window=window, center=True, pad_mode="reflect",
                   normalized=False, onesided=True, return_complex=True)
    s = s.reshape(shape[:-1] + s.shape[-2:])
    return s.abs().pow(power)


def melspectrogram(x, window=None, fb=None):
    if window is None:
        window = torch.hann_window(400, periodic=True)
    if fb is None:
        fb = melscale_fbanks()
    spec = spectrogram(x, window)
    return torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)
