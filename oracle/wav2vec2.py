"""wav2vec2-large as configured by the reference (egs/asr/librispeech/local/chain/tuning/
tdnnf_wav2vec2_vq.py:39-56 -> satools/satools/utils/import_fairseq_model.py:81-113 ->
torchaudio.models.wav2vec2.model.wav2vec2_model), CPU restatement.

THIRD-PARTY: torchaudio is not under /root/reference and not installed here.  This restates the published
torchaudio 2.1 module semantics and is CROSS-CHECKED against an independent implementation that IS installed
in the build container, Hugging Face transformers' `Wav2Vec2Model(do_stable_layer_norm=True,
feat_extract_norm="layer")` — the architecture torchaudio's own `import_huggingface_model` maps one-to-one
onto this configuration — by tests/golden/make_w2v2_crosscheck.py (fixtures tests/golden/fx_w2v2_hf.npz,
tests/test_oracle_w2v2.py).  Against torchaudio itself parity stays unpinned.
  * extractor_mode="layer_norm": 7 x [Conv1d(bias) -> LayerNorm over channels (affine) -> GELU]
    with (k, s) = (10,5), (3,2) x4, (2,2) x2; no waveform normalisation;
  * feature projection: LayerNorm(512) -> Linear(512, 1024);
  * positional conv: weight_norm(Conv1d(1024, 1024, 128, padding=64, groups=16), dim=2), last output
    sample dropped (even kernel), GELU, added to the input;
  * encoder_layer_norm_first=True: the 24 layers are pre-LN, x = x + Attn(LN(x)); x = x + FFN(LN(x)).
    torchaudio builds `Transformer(..., layer_norm_first=not layer_norm_first)` (`_get_encoder`), so the
    encoder-level LayerNorm(1024) runs AFTER the stack in `forward()` (as in fairseq / HF) and NOT AT ALL in
    `get_intermediate_outputs`, which is what `extract_features` returns: the reference's
    `extract_features(x)[0][-1]` (tdnnf_wav2vec2_vq.py:295-297) is the raw output of layer 24.  The
    parameter `encoder.transformer.layer_norm.*` stays in the state dict, unused by `extract_bn`.
    (Round 1 applied that LayerNorm before the stack, following SURVEY Appendix D's "from memory" note; HF
    refutes it: make_w2v2_crosscheck.py measures both placements.)
  * extract_features returns the list of per-layer outputs (the reference takes [-1]).
Parameter names follow torchaudio's state-dict keys so reference checkpoints (`preprocessor.*`) load.
It doubles as the `wav2vec2_model` factory of the fixture generator's torchaudio stand-in."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

CONV_LAYERS = [(512, 10, 5), (512, 3, 2), (512, 3, 2), (512, 3, 2), (512, 3, 2), (512, 2, 2), (512, 2, 2)]


class _ConvBlock(nn.Module):
    def __init__(self, cin, cout, k, s):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, k, s, bias=True)
        self.layer_norm = nn.LayerNorm(cout, elementwise_affine=True)

    def forward(self, x):
        x = self.conv(x)
        x = self.layer_norm(x.transpose(-2, -1)).transpose(-2, -1)
        return F.gelu(x)


class _FeatureExtractor(nn.Module):
    def __init__(self):
        super().__init__()
        blocks, cin = [], 1
        for cout, k, s in CONV_LAYERS:
            blocks.append(_ConvBlock(cin, cout, k, s))
            cin = cout
        self.conv_layers = nn.ModuleList(blocks)

    def forward(self, x):                      # [B, n] -> [B, frames, 512]
        x = x.unsqueeze(1)
        for b in self.conv_layers:
            x = b(x)
        return x.transpose(1, 2)


class _FeatureProjection(nn.Module):
    def __init__(self, cin=512, cout=1024):
        super().__init__()
        self.layer_norm = nn.LayerNorm(cin)
        self.projection = nn.Linear(cin, cout)

    def forward(self, x):
        return self.projection(self.layer_norm(x))


class _PosConv(nn.Module):
    def __init__(self, dim=1024, k=128, groups=16):
        super().__init__()
        self.conv = nn.utils.weight_norm(nn.Conv1d(dim, dim, k, padding=k // 2, groups=groups), name="weight", dim=2)
        self.k = k

    def forward(self, x):                      # [B, T, C]
        y = self.conv(x.transpose(-2, -1))
        if self.k % 2 == 0:
            y = y[..., :-1]
        return F.gelu(y).transpose(-2, -1)


class _SelfAttention(nn.Module):
    def __init__(self, dim=1024, heads=16):
        super().__init__()
        self.heads, self.hd = heads, dim // heads
        self.k_proj = nn.Linear(dim, dim)
        self.v_proj = nn.Linear(dim, dim)
        self.q_proj = nn.Linear(dim, dim)
        self.out_proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, T, C = x.shape
        sh = lambda t: t.view(B, T, self.heads, self.hd).transpose(2, 1)
        q, k, v = sh(self.q_proj(x)), sh(self.k_proj(x)), sh(self.v_proj(x))
        w = torch.softmax((q * (self.hd ** -0.5)) @ k.transpose(-2, -1), dim=-1)
        o = (w @ v).transpose(2, 1).reshape(B, T, C)
        return self.out_proj(o)


class _FeedForward(nn.Module):
    def __init__(self, dim=1024, inter=4096):
        super().__init__()
        self.intermediate_dense = nn.Linear(dim, inter)
        self.output_dense = nn.Linear(inter, dim)

    def forward(self, x):
        return self.output_dense(F.gelu(self.intermediate_dense(x)))


class _EncoderLayer(nn.Module):
    def __init__(self, dim=1024, heads=16, inter=4096):
        super().__init__()
        self.attention = _SelfAttention(dim, heads)
        self.layer_norm = nn.LayerNorm(dim)
        self.feed_forward = _FeedForward(dim, inter)
        self.final_layer_norm = nn.LayerNorm(dim)

    def forward(self, x):                      # layer_norm_first
        x = x + self.attention(self.layer_norm(x))
        return x + self.feed_forward(self.final_layer_norm(x))


class _Transformer(nn.Module):
    def __init__(self, dim=1024, layers=24, heads=16, inter=4096, k=128, groups=16):
        super().__init__()
        self.pos_conv_embed = _PosConv(dim, k, groups)
        self.layer_norm = nn.LayerNorm(dim)
        self.layers = nn.ModuleList([_EncoderLayer(dim, heads, inter) for _ in range(layers)])

    def intermediate(self, x, ln_placement="none_in_extract_features"):
        x = x + self.pos_conv_embed(x)
        if ln_placement == "before_stack":     # the refuted round-1 reading, kept only for the cross-check script
            x = self.layer_norm(x)
        else:
            assert ln_placement == "none_in_extract_features"
        outs = []
        for layer in self.layers:
            x = layer(x)
            outs.append(x)
        return outs


class _Encoder(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.feature_projection = _FeatureProjection()
        self.transformer = _Transformer(**kw)


class Wav2Vec2Restated(nn.Module):
    def __init__(self, num_layers=24):
        super().__init__()
        self.feature_extractor = _FeatureExtractor()
        self.encoder = _Encoder(layers=num_layers)

    def extract_features(self, waveforms, lengths=None, num_layers=None, _ln_placement="none_in_extract_features"):
        x = self.feature_extractor(waveforms)
        x = self.encoder.feature_projection(x)
        return self.encoder.transformer.intermediate(x, _ln_placement), None

    def forward(self, waveforms, lengths=None):
        """torchaudio's `forward()`: the encoder-level LayerNorm after the pre-LN stack"""
        return self.encoder.transformer.layer_norm(self.extract_features(waveforms)[0][-1]), None


def build_wav2vec2(*args, **cfg):
    """factory with torchaudio's `wav2vec2_model(**cfg)` signature for the configuration the reference
    passes; anything else is refused instead of silently mis-built"""
    assert cfg.get("extractor_mode") == "layer_norm" and cfg.get("encoder_layer_norm_first") is True
    assert [list(c) for c in cfg["extractor_conv_layer_config"]] == [list(c) for c in CONV_LAYERS]
    assert cfg["encoder_embed_dim"] == 1024 and cfg["encoder_num_heads"] == 16 and cfg["encoder_ff_interm_features"] == 4096
    assert cfg["encoder_pos_conv_kernel"] == 128 and cfg["encoder_pos_conv_groups"] == 16 and cfg.get("aux_num_out") is None
    return Wav2Vec2Restated(cfg["encoder_num_layers"])


def frames_out(n):
    """number of 20 ms frames the conv stack produces for n samples"""
    for _, k, s in CONV_LAYERS:
        n = (n - k) // s + 1
    return n
