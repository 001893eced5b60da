"""F0 normalisation and transformations, CPU restatement.
Reference: UttCMVN(var_norm=True, keep_zeros=True) satools/satools/cmvn.py:143-155;
quantize_f0 / awgn_f0 satools/satools/hifigan/nn.py:28-62."""
import torch


def norm_keep_zeros_(x):
    """in place: over the non-zero entries of the WHOLE tensor subtract their mean and divide by
    sqrt(unbiased var + 1e-6); zeros stay zero"""
    uv, vv = x == 0, x != 0
    mean = x[vv].mean()
    std = torch.sqrt(x[vv].var() + 1e-6)
    x[vv] = x[vv] - mean
    x[vv] /= std
    x[uv] = 0
    return x


def quantize(x, bins):
    flat = x.reshape(-1).clone()
    uv = flat == 0
    flat = torch.round(flat * bins) / bins
    flat[uv] = 0
    return flat.view(x.shape)


def awgn(pitch, noise):
    """noise drawn by the caller with torch.normal(mean=tensor(0.), std=sqrt(tensor(10**(db/10))),
    size=pitch.shape) on the CPU global generator"""
    ii = pitch == 0
    pitch = pitch + noise.to(pitch.dtype)
    pitch[ii] = 0
    return pitch
