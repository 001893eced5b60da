"""torch.hub entry points with the reference's names and tag grammar (reference: hubconf.py:32-87):

    model = torch.hub.load("/path/to/this/repo", "anonymization", source="local",
                           tag_version="hifigan_bn_tdnnf_600h_vq_48_v1+f0-transformation=quant_16_awgn_2")
    wav_conv = model.convert(wav.to("cuda"), target="6081")

`tag_version` = release tag `[+key=value]*` (`-` in keys becomes `_`).  The target machines have no
network, so instead of downloading `final.pt` from GitHub releases the tag is resolved locally:
`$SATOOLS_AMD_CHECKPOINTS/<tag>/final.pt` (a reference-format checkpoint), or `synthetic:<tag>` for
the architecture with seeded random weights.  The GitHub commit check of the reference
(`exit_if_new_version`) has no offline meaning and is accepted but ignored."""
import os
import sys

dependencies = ["torch", "numpy"]

_ROOT = os.path.dirname(os.path.abspath(__file__))


def _load(tag_version):
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    os.environ["SA_JIT_TWEAK"] = "true"
    import satools_amd
    parts = tag_version.split("+")
    tag, option_args = parts[0], {}
    for o in parts[1:]:
        key, value = o.split("=")
        option_args[key.replace("-", "_")] = value
    if tag.startswith("synthetic:"):
        return satools_amd.load_model(tag, option_args=option_args)
    return satools_amd.load_model(os.path.join(tag, "final.pt"), option_args=option_args)


def asr_bn_extractor(tag_version="bn_tdnnf_wav2vec2_vq_48_v1", exit_if_new_version=False):
    """ASR-bottleneck extractor (`extract_bn`); returned in eval mode like the reference"""
    m = _load(tag_version)
    m.eval()
    return m


def anonymization(tag_version="hifigan_bn_tdnnf_wav2vec2_vq_48_v1", exit_if_new_version=False):
    """anonymization model (`convert`, `get_bn`, `get_f0`, ...)"""
    return _load(tag_version)
