"""ASV x-vector forward (ECAPA-TDNN), CPU restatement over the reference's state dict — SURVEY row aX / f3.
Reference: egs/asv/voxceleb/local/tuning/ecapa_tdnn.py:18-81 (Net), satools/satools/sidekit/preprocessor.py:164-236
(MelSpecFrontEnd), satools/satools/augmentation.py:219-244 (PreEmphasis), sidekit/archi.py:163-189 (PreEcapaTDNN),
sidekit/nn.py:75-154 (Res2Conv1dReluBn, Conv1dReluBn, SE_Connect, SE_Res2Block), sidekit/pooling.py:141-155
(AttentiveStatsPool).  Eval mode: SpecAugment and the masking transforms are training-only.
The mel spectrogram itself is torchaudio's (oracle/melspec.py: third party, parity unpinned).
Test infrastructure only."""
import torch
import torch.nn.functional as F

from . import melspec


def pre_emphasis(x, coef=0.97):
    """[B, n] -> [B, n]: y[t] = x[t] - coef * x[t-1] with x[-1] = x[1] (reflect pad of one sample)"""
    xp = F.pad(x.unsqueeze(1), (1, 0), "reflect")
    w = torch.tensor([[[-coef, 1.0]]], dtype=x.dtype)
    return F.conv1d(xp, w).squeeze(1)


def front_end(x):
    """[B, n] -> [B, 80, 1 + n // 160]: pre-emphasis, mel spectrogram + 1e-6, log, InstanceNorm1d over time"""
    m = melspec.melspectrogram(pre_emphasis(x)) + 1e-6
    return F.instance_norm(torch.log(m), eps=1e-5)


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, 1e-5)


def _conv_relu_bn(sd, p, x, pad=0, dil=1):
    return _bn(sd, p + "bn.", F.relu(F.conv1d(x, sd[p + "conv.weight"], None, padding=pad, dilation=dil)))


def _se_res2block(sd, p, x, dil, scale=8):
    y = _conv_relu_bn(sd, p + "0.", x)
    width = y.shape[1] // scale
    spx = torch.split(y, width, 1)
    out, sp = [], spx[0]
    for i in range(scale - 1):
        if i >= 1:
            sp = sp + spx[i]
        sp = F.conv1d(sp, sd[f"{p}1.convs.{i}.weight"], None, padding=dil, dilation=dil)
        sp = _bn(sd, f"{p}1.bns.{i}.", F.relu(sp))
        out.append(sp)
    out.append(spx[scale - 1])
    z = _conv_relu_bn(sd, p + "2.", torch.cat(out, dim=1))
    g = z.mean(dim=2)
    g = F.relu(F.linear(g, sd[p + "3.linear1.weight"], sd[p + "3.linear1.bias"]))
    g = torch.sigmoid(F.linear(g, sd[p + "3.linear2.weight"], sd[p + "3.linear2.bias"]))
    return z * g.unsqueeze(2)


def sequence_network(sd, x, hook=None, p="sequence_network."):
    out1 = _conv_relu_bn(sd, p + "layer1.", x, pad=2)
    out2 = _se_res2block(sd, p + "layer2.", out1, 2) + out1
    out3 = _se_res2block(sd, p + "layer3.", out1 + out2, 3) + out1 + out2
    out4 = _se_res2block(sd, p + "layer4.", out1 + out2 + out3, 4) + out1 + out2 + out3
    if hook:
        hook("out1", out1), hook("out2", out2), hook("out3", out3), hook("out4", out4)
    out = torch.cat([out2, out3, out4], dim=1)
    return F.relu(F.conv1d(out, sd[p + "conv.weight"], sd[p + "conv.bias"]))


def attentive_stats_pool(sd, x, p="stat_pooling."):
    a = torch.tanh(F.conv1d(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"]))
    a = torch.softmax(F.conv1d(a, sd[p + "linear2.weight"], sd[p + "linear2.bias"]), dim=2)
    mean = torch.sum(a * x, dim=2)
    resid = torch.sum(a * x ** 2, dim=2) - mean ** 2
    return torch.cat([mean, torch.sqrt(resid.clamp(min=1e-9))], dim=1)


def xvector(sd, wav, hook=None):
    """wav [n] or [B, n] f32 in [-1, 1] -> L2-normalised x-vectors [B, 192]"""
    if wav.dim() == 1:
        wav = wav.unsqueeze(0)
    feats = front_end(wav)
    if hook:
        hook("feats", feats)
    h = sequence_network(sd, feats, hook)
    if hook:
        hook("seq", h)
    pooled = attentive_stats_pool(sd, h)
    if hook:
        hook("pooled", pooled)
    e = F.linear(pooled, sd["before_speaker_embedding.lin.weight"])
    e = F.batch_norm(e, sd["before_speaker_embedding.bn2.running_mean"], sd["before_speaker_embedding.bn2.running_var"],
                     sd["before_speaker_embedding.bn2.weight"], sd["before_speaker_embedding.bn2.bias"], False, 0.0, 1e-5)
    return F.normalize(e, dim=1)
