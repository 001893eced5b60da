"""TDNNF-VQ bottleneck extractor, CPU restatement.
Reference: egs/asr/librispeech/local/chain/tuning/tdnnf_vq.py:228-257 (extract_bn, pad_input),
satools/satools/chain/nn.py:267-292 (TDNNF), :338-347 (TDNNFBatchNorm), :402-476 (VQ eval),
satools/satools/chain/objf.py:137-144 (affine), satools/satools/cmvn.py:157-165 (UttCMVN())."""
import torch
import torch.nn.functional as F

from . import fbank as fb


def pad_input(x, pad):
    """tdnnf_vq.py:228-234; x [N, T, C].  Left: the first frame of each utterance `pad` times.
    Right: the reference builds it as `x[:, -1, :].repeat(1, pad, 1).reshape(N, -1, C)`; the
    2-D slice is tiled as ONE sequence [last_0, last_1, ..., last_{N-1}, last_0, ...] of N*pad
    frames and then cut into N pieces, so for N > 1 right-pad frame p of utterance b is the last
    frame of utterance (b*pad + p) mod N.  Restated as is (it is what the reference computes)."""
    if pad <= 0:
        return x
    N = x.shape[0]
    left = x[:, :1].expand(-1, pad, -1)
    src = (torch.arange(N * pad) % N).view(N, pad)
    right = x[:, -1, :][src]
    return torch.cat([left, x, right], 1)


def vq(z, codebook):
    """chain/nn.py:424-459 eval branch.  z [N, T, D] -> (quantized, indices [N*T], distances)"""
    flat = z.reshape(-1, z.shape[-1])
    dist = (torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1)
            - 2 * torch.matmul(flat, codebook.t()))
    idx = torch.argmin(dist, dim=1)
    enc = torch.zeros(idx.shape[0], codebook.shape[0], dtype=torch.float32)
    enc.scatter_(1, idx.unsqueeze(1), 1)
    quant = torch.matmul(enc, codebook).view(z.shape)
    quant = z + (quant - z)
    return quant, idx, dist


def tdnnf_layer(sd, prefix, x, ctx, sub, bypass, return_bottleneck=False, aux=None):
    """one TDNNFBatchNorm; x [N, T, D].  prefix e.g. 'tdnnfs.0.'"""
    N, T, D = x.shape
    win = x.reshape(N, -1).unfold(1, D * ctx, int(D * sub)).contiguous()      # sub = 1.5: every other window straddles two frames
    wB, bB = sd[prefix + "tdnn.linearB.inner_nat.weight"], sd[prefix + "tdnn.linearB.inner_nat.bias"]
    z = win.matmul(wB.t())
    z = z + bB
    cb_key = prefix + "bottleneck_func.quant._embedding.weight"
    if cb_key in sd:
        zq, idx, dist = vq(z, sd[cb_key])
        if aux is not None:
            aux.update(z=z, idx=idx.view(N, -1), dist=dist.view(N, z.shape[1], -1))
        z = zq
    if return_bottleneck:
        return z
    y = F.linear(z, sd[prefix + "tdnn.linearA.weight"], sd[prefix + "tdnn.linearA.bias"])
    if bypass and sub == 1.5:
        # add_padd (chain/nn.py:294-304): frames 0, 1, 3, 4, 6, 7, ... of the input, the first int(T / 1.5) of them,
        # zero-padded at the end to the length of y
        idx = torch.arange(0, 16000 * 100, 1.5).long()[:int(T / 1.5)]
        byp = torch.index_select(x, 1, idx) * 0.66
        if y.shape[1] < byp.shape[1]:
            y = F.pad(y, [0, 0, 0, byp.shape[1] - y.shape[1], 0, 0])
        else:
            byp = F.pad(byp, [0, 0, 0, y.shape[1] - byp.shape[1], 0, 0])
        y = y + byp
    elif bypass:
        l = ctx // 2 if ctx > 1 else 0
        r = -l if (ctx > 1 and ctx % 2 == 1) else None
        y = y + x[:, l:r:sub, :] * 0.66
    y = F.batch_norm(y.permute(0, 2, 1), sd[prefix + "bn.running_mean"], sd[prefix + "bn.running_var"],
                     None, None, False, 0.1, 1e-5).permute(0, 2, 1)
    return F.relu(y)


FBANK_KS = [3, 3, 3, 1, 3, 3, 3, 3, 3, 3, 3, 3]
FBANK_SUB = [1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1]
W2V2_KS = [3, 3, 3]
W2V2_SUB = [1, 1, 1]


def run_stack(sd, x, ks, subs, aux=None, hook=None):
    """tdnn1, tdnnfs[0,2,...] and the bottleneck of the last one (tdnnf_vq.py:250-256)"""
    feat = x.shape[-1]
    x = tdnnf_layer(sd, "tdnn1.", x, ks[0], subs[0], bypass=(feat == 1024))
    if hook:
        hook("tdnn1", x)
    for i in range(1, len(ks) - 1):
        name = f"tdnnfs.{2 * (i - 1)}."
        x = tdnnf_layer(sd, name, x, ks[i], subs[i], bypass=True)
        if hook:
            hook(name[:-1], x)
    last = f"tdnnfs.{2 * (len(ks) - 2)}."
    return tdnnf_layer(sd, last, x, ks[-1], subs[-1], bypass=False, return_bottleneck=True, aux=aux)


def extract_bn_fbank(sd, wav, aux=None, hook=None):
    """sd: state dict of the ASR-BN net (keys without the 'bn_extractor.' prefix).
    wav [N, n] in [-1, 1]  ->  [N, T, 256]"""
    x = wav.detach().clone() * 32768
    x = fb.fbank(x, 80)
    if hook:
        hook("fbank", x)
    x = x - x.mean(dim=1).unsqueeze(1)          # UttCMVN(): mean over frames
    x = pad_input(x, 19)
    return run_stack(sd, x, FBANK_KS, FBANK_SUB, aux=aux, hook=hook)


AFTER_KS = [1, 3, 3, 3]
AFTER_SUB = [1.5, 1, 1, 1]


def _asr_head(sd, x, ks, subs, hook=None):
    """tdnn1 .. VQ layer (all of it) -> pad_input(padding_after) -> tdnnfs_after -> prefinals -> output affines
    (tdnnf_vq.py:259-284 == tdnnf_wav2vec2_vq.py:316-345 from `self.tdnn1(x)` on)"""
    # TDNNF adds its bypass when feat_dim == output_dim (chain/nn.py:279-292): never for 80 fbank bins, always for the
    # 1024-dimensional wav2vec2 features
    x = tdnnf_layer(sd, "tdnn1.", x, ks[0], subs[0], bypass=(x.shape[-1] == 1024))
    for i in range(1, len(ks) - 1):
        x = tdnnf_layer(sd, f"tdnnfs.{2 * (i - 1)}.", x, ks[i], subs[i], bypass=True)
    x = tdnnf_layer(sd, f"tdnnfs.{2 * (len(ks) - 2)}.", x, ks[-1], subs[-1], bypass=False)      # VQ layer, all of it
    if hook:
        hook("vq_layer", x)
    pad, g = 0.0, 1.0
    for k, s_ in zip(AFTER_KS, AFTER_SUB):                      # ChainE2EModel.get_padding (chain/model.py:466-473)
        pad += (k - 1) * g
        g *= s_
    x = pad_input(x, int(pad) // 2)
    for i, (k, s_) in enumerate(zip(AFTER_KS, AFTER_SUB)):
        x = tdnnf_layer(sd, f"tdnnfs_after.{2 * i}.", x, k, s_, bypass=True)
        if hook:
            hook(f"after{2 * i}", x)
    pc = tdnnf_layer(sd, "prefinal_chain.", x, 1, 1, bypass=True)
    px = tdnnf_layer(sd, "prefinal_xent.", x, 1, 1, bypass=True)
    chain = pc.matmul(sd["chain_output.weight"].t()) + sd["chain_output.bias"]
    xent = px.matmul(sd["xent_output.weight"].t()) + sd["xent_output.bias"]
    return chain, F.log_softmax(xent, dim=2)


def forward_fbank(sd, wav, hook=None):
    """the ASR half of the fbank-tag net, `Net.forward` (tdnnf_vq.py:259-284): wav [N, n] ->
    (chain_out [N, T', output_dim], log_softmax(xent_out) [N, T', output_dim]).  Eval mode (dropout = identity)."""
    x = wav.detach().clone() * 32768
    x = fb.fbank(x, 80)
    x = x - x.mean(dim=1).unsqueeze(1)
    x = pad_input(x, 19)
    return _asr_head(sd, x, FBANK_KS, FBANK_SUB, hook=hook)


def forward_w2v2(sd, wav, hook=None, model=None):
    """`Net.forward` of the wav2vec2-tag net (tdnnf_wav2vec2_vq.py:316-345): the same head behind the wav2vec2
    features (raw waveform, no 32768 scaling; replicate-pad one frame; pad_input(3))"""
    from . import wav2vec2 as w2
    if model is None:
        model = w2.Wav2Vec2Restated(24)
        model.load_state_dict({k[len("preprocessor."):]: v for k, v in sd.items() if k.startswith("preprocessor.")})
        model.eval()
    with torch.no_grad():
        x = model.extract_features(wav.detach().clone())[0][-1]
    x = F.pad(x.transpose(2, 1), (0, 1), "replicate").transpose(2, 1).to(torch.float32)
    x = pad_input(x, 3)
    return _asr_head(sd, x, W2V2_KS, W2V2_SUB, hook=hook)


def extract_bn_w2v2(sd, wav, aux=None, hook=None, model=None):
    """wav2vec2-tag bottleneck extractor (tdnnf_wav2vec2_vq.py:289-314): last transformer layer output
    [N, 249, 1024] -> replicate-pad one frame -> pad_input(3) -> tdnn1, tdnnfs[0], VQ bottleneck of
    tdnnfs[2].  `sd` = ASR-BN state dict (keys without 'bn_extractor.'); the wav2vec2 weights are its
    'preprocessor.*' entries (torchaudio key names)."""
    from . import wav2vec2 as w2
    if model is None:
        model = w2.Wav2Vec2Restated(24)
        model.load_state_dict({k[len("preprocessor."):]: v for k, v in sd.items() if k.startswith("preprocessor.")})
        model.eval()
    with torch.no_grad():
        x = model.extract_features(wav.detach().clone())[0][-1]
    if hook:
        hook("w2v2", x)
    x = F.pad(x.transpose(2, 1), (0, 1), "replicate").transpose(2, 1).to(torch.float32)
    x = pad_input(x, 3)
    return run_stack(sd, x, W2V2_KS, W2V2_SUB, aux=aux, hook=hook)
