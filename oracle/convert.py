"""End-to-end `Net.convert` / `_forward`, CPU restatement
(reference: egs/vc/libritts/local/tuning/hifigan.py:58-102)."""
import torch
import torch.nn.functional as F

from . import f0 as f0mod
from . import hifigan, tdnnf


def split_state_dict(sd):
    asr = {k[len("bn_extractor."):]: v for k, v in sd.items() if k.startswith("bn_extractor.")}
    gen = {k[len("hifigan."):]: v for k, v in sd.items() if k.startswith("hifigan.")}
    return asr, gen


def spk_one_hot(spk_list, target):
    tg = [target] if isinstance(target, str) else target
    return F.one_hot(torch.tensor([spk_list.index(t) for t in tg]), num_classes=len(spk_list))


def forward(gen_sd, f0, bn, spk_id, quant_bins=0, noise=None, hook=None):
    """f0 [1,B,T] or [B,T] (normalised IN PLACE), bn [B,256,T], spk_id [B,N] int64 -> [B,1,n']"""
    f0 = f0mod.norm_keep_zeros_(f0)
    if f0.dim() == 2:
        f0 = f0.unsqueeze(0)
    f0 = f0.permute(1, 0, 2)
    if quant_bins:
        f0 = f0mod.quantize(f0, quant_bins)
    if noise is not None:
        f0 = f0mod.awgn(f0, noise)
    f0i = F.interpolate(f0, bn.shape[-1])
    x = torch.cat([bn, f0i], dim=1)
    s = F.interpolate(spk_id.unsqueeze(2).to(torch.float32), x.shape[-1])
    x = torch.cat([x, s], dim=1)
    if hook:
        hook("gen_in", x)
    return hifigan.generator(gen_sd, x, hook=hook)


def convert_fbank(sd, spk_list, wav, target, f0, quant_bins=0, noise=None):
    """fbank-tag convert with F0 given ([B,T] Hz, 0 = unvoiced); returns like the reference:
    [1, n'] for B == 1, [B, 1, n'] otherwise (the `.squeeze(0)` of hifigan.py:71)"""
    asr, gen = split_state_dict(sd)
    bn = tdnnf.extract_bn_fbank(asr, wav).permute(0, 2, 1)
    y = forward(gen, f0.clone().unsqueeze(0), bn, spk_one_hot(spk_list, target), quant_bins, noise)
    return y.squeeze(0)


def convert_w2v2(sd, spk_list, wav, target, f0, quant_bins=0, noise=None, model=None):
    """wav2vec2-tag convert with F0 given; `model` = a prebuilt oracle.wav2vec2.Wav2Vec2Restated holding the
    `preprocessor.*` weights (315 M parameters: build it once outside a timed region)"""
    asr, gen = split_state_dict(sd)
    bn = tdnnf.extract_bn_w2v2(asr, wav, model=model).permute(0, 2, 1)
    y = forward(gen, f0.clone().unsqueeze(0), bn, spk_one_hot(spk_list, target), quant_bins, noise)
    return y.squeeze(0)
