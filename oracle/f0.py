"""F0 normalisation and transformations, CPU restatement.
Reference: UttCMVN(var_norm=True, keep_zeros=True) satools/satools/cmvn.py:143-155;
quantize_f0 / awgn_f0 satools/satools/hifigan/nn.py:28-62; moving_average_f0 / mean_reverv_f0 nn.py:64-90."""
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


def parse_mean_reverv(spec):
    """"mean-reverv_<alpha>:<n>" -> (alpha, n): alpha = the digits and dots of the token's part before ':', n = the
    digits of the part after it (nn.py:83-87)"""
    tok = spec[spec.index("mean-reverv"):].split("_")[1]
    alpha = float("".join(c for c in tok.split(":")[0] if c.isdigit() or c == "."))
    n = int("".join(c for c in tok.split(":")[1] if "0" <= c <= "9"))
    return alpha, n


def mean_reversion(f0, alpha, n):
    """f0 [1, 1, T] -> (1 - alpha) * f0 + alpha * moving_average(f0, n).  The reference squeezes the padded
    [B, 1, T + 2 (n // 2)] tensor to 2-D and hands it to conv1d, which reads it as ONE unbatched sequence of B
    channels against a 1-channel window: it works for B = 1 only and raises RuntimeError otherwise (nn.py:64-76);
    the window sum runs over f0[t - n // 2 .. t - n // 2 + n - 1] (zeros outside, unvoiced zeros included)."""
    import torch.nn.functional as F
    if f0.shape[0] != 1:
        raise RuntimeError(f"mean-reverv: expected a batch of 1 (the reference's conv1d reads [B, T] as B channels), got {f0.shape[0]}")
    pad = n // 2
    fp = F.pad(f0, (pad, pad), mode="constant")
    window = torch.ones(n) / n
    avg = F.conv1d(fp.squeeze(1), window.view(1, 1, -1))[..., :f0.shape[-1]]
    return (1 - alpha) * f0 + alpha * avg
