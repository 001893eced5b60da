#!/usr/bin/env python3
"""Cross-check of the wav2vec2-large restatement against an INDEPENDENT implementation that is
installed in the build container: Hugging Face `transformers` (5.15) `Wav2Vec2Model` with
`do_stable_layer_norm=True, feat_extract_norm="layer", conv_bias=True` — the architecture that
torchaudio's own `import_huggingface_model` maps one-to-one onto
`torchaudio.models.wav2vec2_model(extractor_mode="layer_norm", encoder_layer_norm_first=True, ...)`,
which is what the reference builds (egs/asr/librispeech/local/chain/tuning/tdnnf_wav2vec2_vq.py:39-56
through satools/satools/utils/import_fairseq_model.py:81-113).  torchaudio itself is not installed
and cannot be fetched.

Run from the repo root:   python tests/golden/make_w2v2_crosscheck.py

What it does
  1. takes the seeded synthetic state dict of the wav2vec2-tag ASR-BN net (torchaudio key names,
     `preprocessor.*`) and loads it into the HF model through the submodule correspondence of
     torchaudio's importer: feature_extractor -> feature_extractor, feature_projection ->
     encoder.feature_projection, encoder.* -> encoder.transformer.* (pos_conv weight_g / weight_v =
     parametrizations.weight.original0 / original1);
  2. runs two seeded utterances and captures, with forward hooks: the conv feature extractor
     output, the projected features, the input of transformer layer 0 (positional conv added), the
     RAW output of encoder layers 0, 11 and 23 (what torchaudio's `extract_features(x)[0][-1]`
     returns: `Transformer.get_intermediate_outputs` applies no transformer-level LayerNorm), and
     the output after the encoder-level LayerNorm (which HF / fairseq / torchaudio's `forward()`
     apply AFTER the stack when the layers are pre-LN);
  3. evaluates the CPU restatement (oracle/wav2vec2.py) in its two candidate placements of the
     encoder-level LayerNorm and records which one HF reproduces;
  4. stores inputs-by-seed + the captured tensors (channel-subsampled) as tests/golden/fx_w2v2_hf.npz.

Fixtures are data only; nothing of transformers' source is stored.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path[:0] = [ROOT]


def hf_model(n_layers=24):
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    cfg = Wav2Vec2Config(
        hidden_size=1024, num_hidden_layers=n_layers, num_attention_heads=16, intermediate_size=4096,
        hidden_act="gelu", hidden_dropout=0.0, activation_dropout=0.0, attention_dropout=0.0,
        feat_proj_dropout=0.0, final_dropout=0.0, layerdrop=0.0, layer_norm_eps=1e-5,
        feat_extract_norm="layer", feat_extract_activation="gelu",
        conv_dim=(512,) * 7, conv_stride=(5, 2, 2, 2, 2, 2, 2), conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_bias=True,
        num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16,
        do_stable_layer_norm=True, apply_spec_augment=False, attn_implementation="eager")
    return Wav2Vec2Model(cfg).eval()


def to_hf_keys(sd_torchaudio):
    """torchaudio state-dict keys (relative to the Wav2Vec2Model) -> HF keys"""
    out = {}
    for k, v in sd_torchaudio.items():
        if k.startswith("encoder.feature_projection."):
            k2 = k[len("encoder."):]
        elif k.startswith("encoder.transformer."):
            k2 = "encoder." + k[len("encoder.transformer."):]
            k2 = k2.replace("pos_conv_embed.conv.weight_g", "pos_conv_embed.conv.parametrizations.weight.original0")
            k2 = k2.replace("pos_conv_embed.conv.weight_v", "pos_conv_embed.conv.parametrizations.weight.original1")
        else:
            k2 = k
        out[k2] = v
    return out


def main():
    import torch
    torch.set_num_threads(8)
    from oracle import wav2vec2 as ow
    from satools_amd import synthetic

    # the anonymizer tag's state dict: the same tensors tests and the GPU model get from `synthetic:<tag>`
    state, _ = synthetic.checkpoint("hifigan_bn_tdnnf_wav2vec2_vq_48_v1", conditioning=None)
    pfx = "bn_extractor.preprocessor."
    pre = {k[len(pfx):]: v for k, v in state["base_model_state_dict"].items() if k.startswith(pfx)}
    assert len(pre) > 400, len(pre)
    hf = hf_model()
    res = hf.load_state_dict(to_hf_keys(pre), strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert set(res.missing_keys) <= {"masked_spec_embed"}, res.missing_keys

    wav = synthetic.harm_batch([0, 1], 16000)
    acts = {}
    hooks = [
        hf.feature_projection.register_forward_hook(lambda m, i, o: acts.update(fe=i[0].detach(), proj=o[0].detach())),
        hf.encoder.layers[0].register_forward_pre_hook(lambda m, i: acts.__setitem__("layer0_in", i[0].detach())),
        hf.encoder.layer_norm.register_forward_hook(lambda m, i, o: acts.__setitem__("after_final_ln", o.detach())),
    ]
    for li in (0, 11, 23):
        hooks.append(hf.encoder.layers[li].register_forward_hook(lambda m, i, o, li=li: acts.__setitem__(f"layer{li}", o[0].detach())))
    with torch.no_grad():
        out = hf(wav, output_hidden_states=True)
    for h in hooks:
        h.remove()
    assert torch.equal(out.last_hidden_state, acts["after_final_ln"])

    # the restatement, in both placements of the encoder-level LayerNorm
    om = ow.Wav2Vec2Restated(24)
    om.load_state_dict(pre, strict=True)
    om.eval()
    verdict = {}
    with torch.no_grad():
        for placement in ("none_in_extract_features", "before_stack"):
            ys = om.extract_features(wav, _ln_placement=placement)[0]
            verdict[placement] = {f"layer{li}": float((ys[li] - acts[f"layer{li}"]).abs().max()) for li in (0, 11, 23)}
        y_fwd = om.forward(wav)[0]
        verdict["forward_after_stack_vs_hf_last_hidden_state"] = float((y_fwd - acts["after_final_ln"]).abs().max())
    scale = {k: float(v.abs().max()) for k, v in acts.items()}
    print(json.dumps({"max_abs_diff_vs_hf": verdict, "hf_value_scale": scale}, indent=1))
    good = verdict["none_in_extract_features"]["layer23"]
    bad = verdict["before_stack"]["layer23"]
    assert good < 1e-3 * scale["layer23"] < bad, "HF does not single out one placement"

    fx = {"fe_sub": acts["fe"][:, :, ::8].numpy(), "proj_sub": acts["proj"][:, :, ::16].numpy(),
          "layer0_in_sub": acts["layer0_in"][:, :, ::16].numpy(),
          "layer0_sub": acts["layer0"][:, :, ::16].numpy(), "layer11_sub": acts["layer11"][:, :, ::16].numpy(),
          "layer23_sub": acts["layer23"][:, :, ::16].numpy(), "layer23": acts["layer23"][0].numpy(),
          "after_final_ln_sub": acts["after_final_ln"][:, :, ::16].numpy()}
    np.savez_compressed(os.path.join(GOLD, "fx_w2v2_hf.npz"), **fx)
    json.dump({"transformers": __import__("transformers").__version__, "torch": torch.__version__,
               "input": "synthetic.harm_batch([0, 1], 16000)", "max_abs_diff_vs_hf": verdict, "hf_value_scale": scale},
              open(os.path.join(GOLD, "fx_w2v2_hf.json"), "w"), indent=1)
    print("written", os.path.join(GOLD, "fx_w2v2_hf.npz"))


if __name__ == "__main__":
    main()
