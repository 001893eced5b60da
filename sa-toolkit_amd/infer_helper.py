"""Checkpoint -> model object, with the reference's `load_model` signature
(satools/satools/infer_helper.py:10-59).

The reference stores in a checkpoint the *path* of the Python model-config file that defines its
`Net` (`task_path` + `base_model_path`) and imports it; here those paths select one of the
built-in MI355X implementations of the same configs.  There is no network on the target
machines, so GitHub release downloads are replaced by local lookup:
  * a path to a checkpoint file (final.pt / conf.pt) in the reference's dict format;
  * `synthetic:<tag>[?seed=N]` — the architecture of a release tag with seeded random weights
    (released weights cannot be fetched offline);
  * a bare tag, resolved under $SATOOLS_AMD_CHECKPOINTS/<tag>/final.pt.
"""
import os
import re

import torch

_CONFIGS = {
    "local/tuning/hifigan.py": "anonymizer",
    "local/chain/tuning/tdnnf_vq.py": "tdnnf_vq",
    "local/chain/tuning/tdnnf_wav2vec2_vq.py": "tdnnf_wav2vec2_vq",
    "local/tuning/ecapa_tdnn.py": "xvector",          # egs/asv/voxceleb: the ASV x-vector extractor
}


def _builder(base_model_path):
    kind = _CONFIGS.get(base_model_path.lstrip("/"))
    if kind is None:
        raise NotImplementedError(
            f"model config '{base_model_path}' is not part of the accelerated path "
            f"(supported: {sorted(_CONFIGS)})")
    if kind == "anonymizer":
        from . import anonymizer
        return anonymizer.build
    if kind == "xvector":
        from . import xvector
        return xvector.build
    from . import asrbn

    def build(args):
        cb = int(args.codebook_size) if args.codebook_size is not None else 48
        if kind == "tdnnf_vq":
            return lambda **kw: asrbn.TdnnfVqNet(codebook_size=cb, **kw)
        from . import wav2vec2
        return lambda **kw: wav2vec2.TdnnfWav2vec2VqNet(codebook_size=cb, **kw)

    return build


def asrbn_conf_from_name(path):
    """conf.pt equivalent for an ASR-BN model referenced by name from an anonymizer checkpoint
    (`asrbn_model = ../../asr/librispeech/exp/chain/<name>/final.pt`, hifigan.py:27-29)"""
    name = os.path.basename(os.path.dirname(path))
    m = re.search(r"vq_(\d+)", name)
    if m is None:
        raise NotImplementedError(f"ASR-BN model '{name}' has no VQ bottleneck; only the *_vq_* tags are accelerated")
    w2v2 = "wav2vec2" in name
    return {
        "task_path": "/egs/asr/librispeech",
        "base_model_path": "local/chain/tuning/tdnnf_wav2vec2_vq.py" if w2v2 else "local/chain/tuning/tdnnf_vq.py",
        "base_model_params": {"output_dim": 3280},
        "base_model_args": {"freeze_encoder": "True", "codebook_size": int(m.group(1))},
    }


def _resolve(file, load_weight):
    if os.path.exists(file):
        return file
    root = os.environ.get("SATOOLS_AMD_CHECKPOINTS")
    if root:
        cand = os.path.join(root, os.path.basename(os.path.dirname(file)) or file, "final.pt" if load_weight else "conf.pt")
        if os.path.exists(cand):
            return cand
        cand = os.path.join(root, file, "final.pt")
        if os.path.exists(cand):
            return cand
    return None


def load_model(file, load_weight=True, version="v1", from_file=None, option_args=None):
    from .anonymizer import SimpleNamespace
    if file.startswith("synthetic:"):
        from . import synthetic
        return synthetic.load(file, option_args=option_args)
    if file.startswith("http"):
        raise RuntimeError("no network on the target machines: pass a local checkpoint path, set "
                           "SATOOLS_AMD_CHECKPOINTS, or use 'synthetic:<tag>'")
    if not load_weight:
        file = os.path.join(os.path.dirname(file), "conf.pt")
    path = _resolve(file, load_weight)
    if path is None and not load_weight:
        model_state = asrbn_conf_from_name(file)  # sibling conf.pt of an ASR-BN model, by name
    elif path is None:
        raise FileNotFoundError(f"checkpoint '{file}' not found locally (release downloads are unavailable offline)")
    else:
        model_state = torch.load(path, weights_only=False, map_location="cpu")
    build = _builder(model_state["base_model_path"])
    model_args = dict(model_state.get("base_model_args", {}))
    if option_args:
        model_args.update(option_args)
    net = build(SimpleNamespace(**model_args))(**model_state["base_model_params"])
    if load_weight:
        net.load_state_dict(model_state["base_model_state_dict"])
        if hasattr(net, "check_precision"):
            # a real checkpoint (not `synthetic:`): its first `.to("cuda")` runs the precision guard once (anonymizer.Net._apply)
            net.__dict__["_precision_check_pending"] = True
    return net
