"""String-driven F0 transformations (option `f0_transformation`, e.g. "quant_16_awgn_2").
Parsing follows satools/satools/hifigan/nn.py:19-62: the number is the digits of the token after
"quant" / "awgn"."""
import torch


def _digits(s):
    return "".join(ch for ch in s if "0" <= ch <= "9")


def parse_quant_bins(spec: str) -> int:
    tok = spec[spec.index("quant"):].split("_")[1]
    return int(_digits(tok))


def parse_awgn_db(spec: str) -> int:
    tok = spec[spec.index("awgn"):].split("_")[1]
    return int(_digits(tok))


def parse_mean_reverv(spec: str):
    """"mean-reverv_<alpha>:<n>" -> (alpha, n)   (hifigan/nn.py:83-87)"""
    tok = spec[spec.index("mean-reverv"):].split("_")[1]
    alpha = float("".join(ch for ch in tok.split(":")[0] if "0" <= ch <= "9" or ch == "."))
    return alpha, int(_digits(tok.split(":")[1]))


def draw_awgn(shape, target_noise_db: int) -> torch.Tensor:
    """the reference draws on the CPU global generator with exactly this call
    (hifigan/nn.py:49-57); keeping the call identical keeps the draws identical under a seed"""
    target_noise_watts = 10 ** (target_noise_db / 10)
    return torch.normal(mean=torch.tensor(0.0), std=torch.sqrt(torch.tensor(target_noise_watts)), size=shape)
