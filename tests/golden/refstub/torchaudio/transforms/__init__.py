"""stand-in for torchaudio.transforms (not installed): MelSpectrogram as oracle/melspec.py restates it, with
torchaudio's module / buffer names (`spectrogram.window`, `mel_scale.fb`) so that state dicts keep their keys;
the masking transforms are training-only and pass through."""
import torch

from oracle import melspec as _ms


class Spectrogram(torch.nn.Module):
    def __init__(self, n_fft, win_length, hop_length, window_fn, power):
        super().__init__()
        self.n_fft, self.win_length, self.hop_length, self.power = n_fft, win_length, hop_length, power
        self.register_buffer("window", window_fn(win_length))

    def forward(self, x):
        return _ms.spectrogram(x, self.window, self.n_fft, self.hop_length, self.win_length, self.power)


class MelScale(torch.nn.Module):
    def __init__(self, n_mels, sample_rate, f_min, f_max, n_stft):
        super().__init__()
        self.register_buffer("fb", _ms.melscale_fbanks(n_stft, f_min, f_max, n_mels, sample_rate))

    def forward(self, spec):
        return torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)


class MelSpectrogram(torch.nn.Module):
    def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, f_min=0.0, f_max=None, pad=0,
                 n_mels=128, window_fn=torch.hann_window, power=2.0, **kw):
        super().__init__()
        assert not kw, f"unsupported MelSpectrogram arguments {sorted(kw)}"
        win_length = win_length or n_fft
        hop_length = hop_length or win_length // 2
        self.spectrogram = Spectrogram(n_fft, win_length, hop_length, window_fn, power)
        self.mel_scale = MelScale(n_mels, sample_rate, float(f_min), float(f_max if f_max is not None else sample_rate // 2),
                                  n_fft // 2 + 1)

    def forward(self, x):
        return self.mel_scale(self.spectrogram(x))


class _Identity(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, x):
        return x


TimeMasking = FrequencyMasking = _Identity
