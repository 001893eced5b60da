"""Frozen, kernel-ready model files: the analogue of the reference's `final.jit`
(satools/satools/hifigan/model.py:162-171, chain/model.py:167-173: a TorchScript export that loads without the training
code and without re-deriving anything from the training parameters).

`export_frozen(model, path)` stores what the HIP kernels read — weight-norm folded (97 convs of the generator), BatchNorm
folded, ConvTranspose1d rewritten as polyphase convs, everything split to hi | lo f16 and packed in the kernels' layout
with its power-of-two layer scale — plus the configuration the checkpoint named (model-config path, build arguments,
utt2spk).  `load_frozen(path, device)` rebuilds the same `Net` object WITHOUT parameters (they are replaced by empty
placeholders: a frozen model is inference-only, like `final.jit`) and installs the packed weights directly: no fold, no
pack, no 1.3 GB of f32 weights of the wav2vec2 tag on the device.  The public interface (convert / get_bn / get_f0 /
extract_features / _forward / spk / ...) is unchanged and the outputs are bit-identical to the model the file was made
from (tests/test_hip_robust.py)."""
import torch
import torch.nn as nn

from . import _lib

FORMAT = "satools_amd.frozen/2"      # 2: plain build arguments only, SAT_CONV_F16F8R second packings
OLD_FORMATS = ("satools_amd.frozen/1",)


def _enc(x):
    """cache tree -> plain python / CPU tensors (a packed weight keeps its layer scale)"""
    from .asrbn import _LayerCache
    if isinstance(x, torch.Tensor):
        d = {"__t__": x.detach().cpu()}
        if hasattr(x, "w_descale"):
            d["w_descale"] = float(x.w_descale)
        if hasattr(x, "up_zero_taps"):
            d["up_zero_taps"] = int(x.up_zero_taps)
        return d
    if isinstance(x, _LayerCache):
        return {"__lc__": {k: _enc(getattr(x, k, None)) for k in _LayerCache.__slots__}}
    if isinstance(x, dict):
        return {"__d__": {k: _enc(v) for k, v in x.items()}}
    if isinstance(x, (list, tuple)):
        return {"__l__": [_enc(v) for v in x], "tuple": isinstance(x, tuple)}
    if isinstance(x, nn.Module):
        return {"__m__": True}                      # re-linked by position at load
    if x is None or isinstance(x, (bool, int, float, str)):
        return x
    raise TypeError(f"frozen export: cannot store {type(x)}")


def _plain(x):
    """build arguments as plain python (what `torch.load(weights_only=True)` accepts).  Anything else is REFUSED at export time:
    a value stored as its str() would come back with another type into `anonymizer.build` (a lossy export nobody notices)."""
    if x is None or isinstance(x, (bool, int, float, str)):
        return x
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    if isinstance(x, dict):
        return {str(k): _plain(v) for k, v in x.items()}
    import pathlib
    if isinstance(x, pathlib.PurePath):             # (a path is its string: `build` only ever formats / opens it)
        return str(x)
    raise _lib.SatError(f"export_frozen: build argument of type {type(x).__name__} cannot be stored as plain data ({x!r})")


def _dec(x, device, mods=None):
    from .asrbn import _LayerCache
    if isinstance(x, dict):
        if "__t__" in x:
            t = x["__t__"].to(device)
            if "w_descale" in x:
                t.w_descale = x["w_descale"]
            if "up_zero_taps" in x:
                t.up_zero_taps = x["up_zero_taps"]
            return t
        if "__lc__" in x:
            c = _LayerCache()
            for k, v in x["__lc__"].items():
                setattr(c, k, _dec(v, device))
            return c
        if "__d__" in x:
            return {k: _dec(v, device, mods) for k, v in x["__d__"].items()}
        if "__l__" in x:
            out = [_dec(v, device, mods) for v in x["__l__"]]
            return tuple(out) if x["tuple"] else out
        if "__m__" in x:
            return mods.pop(0) if mods else None
    return x


def _strip_parameters(net, device):
    """every parameter / buffer becomes an empty placeholder on `device` (the kernels read the packed copies)"""
    for m in net.modules():
        for k in list(m._parameters):
            if m._parameters[k] is not None:
                m._parameters[k] = nn.Parameter(torch.empty(0, device=device), requires_grad=False)
        for k in list(m._buffers):
            if m._buffers[k] is not None:
                m._buffers[k] = torch.empty(0, device=device)


def export_frozen(model, path):
    """model: an anonymizer `Net` on the HIP device (satools_amd.load_model(...).to('cuda'))"""
    gen = model.hifigan
    dev = model._device()
    if dev.type != "cuda":
        raise _lib.SatError("export_frozen: move the model to the HIP device first (the packed layout is built there)")
    if gen.__dict__.get("_frozen"):
        raise _lib.SatError("export_frozen: the model is already a frozen one")
    ext = model.bn_extractor
    gen._prepare(dev)
    ext._prepare(dev)
    blob = {"format": FORMAT, "kind": "anonymizer",
            "build_args": _plain(dict(model._build_args)), "utt2spk": {str(k): str(v) for k, v in dict(model.utt2spk).items()},
            "generator": {"packed": _enc(gen._packed), "modes": list(gen._packed_modes), "precision": gen.precision,
                          # row order of the stride-4 upsamplers' packed weights (hifigan.py: ups_ring); absent in older files = 0
                          "ups_grouped": int(bool(gen.__dict__.get("_packed_ups_grouped", False))),
                          # second packings of the thick stages' ResBlock convs for SAT_CONV_F16F8R ("f16f8r"), {conv id: tensor}
                          "packed8": _enc({str(k): v for k, v in (gen.__dict__.get("_packed8") or {}).items()}),
                          "f8_stages": int(gen.f8_stages)},
            "extractor": {"class": type(ext).__name__, "precision": ext.precision, "cache": _enc(ext._cache)}}
    if hasattr(ext, "_prepare_full"):
        full = ext._prepare_full(dev)
        blob["extractor"]["cache_full"] = _enc({"after": [c for _, c in full["after"]], "prefinal": [c for _, c in full["prefinal"]],
                                                "out": full["out"]})
    if hasattr(ext, "_prepare_w2v2"):
        blob["extractor"]["w2"] = _enc(ext._prepare_w2v2(dev))
        blob["extractor"]["mm_mode"] = int(ext._mm_mode)
        blob["extractor"]["w2v2_precision"] = ext.w2v2_precision
    torch.cuda.synchronize(dev)
    torch.save(blob, path)
    return path


def load_frozen(path, device="cuda"):
    """-> the anonymizer `Net`, on `device`, in eval mode, ready for convert(); no parameters inside"""
    from . import anonymizer
    from .asrbn import TDNNFBatchNormParams
    # plain containers, numbers, strings and tensors only (_enc refuses anything else at export): the restricted unpickler
    # is enough, and a file from elsewhere cannot run code at load time
    try:
        blob = torch.load(path, weights_only=True, map_location="cpu")
    except Exception as e:                          # a file of an older export may hold objects the restricted unpickler refuses
        raise _lib.SatError(f"{path}: cannot be read as a {FORMAT} file ({type(e).__name__}: {e}); re-export it with this version "
                            "(export_frozen)") from e
    if isinstance(blob, dict) and blob.get("format") in OLD_FORMATS:
        raise _lib.SatError(f"{path}: written as {blob.get('format')}; re-export it with this version (export_frozen writes {FORMAT})")
    if not isinstance(blob, dict) or blob.get("format") != FORMAT or blob.get("kind") != "anonymizer":
        raise _lib.SatError(f"{path}: not a {FORMAT} anonymizer file")
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.SatError("load_frozen: the packed weights only exist for the HIP device (no CPU fallback)")
    with torch.device("meta"):                      # the architecture only: nothing is allocated or initialised
        net = anonymizer.build(anonymizer.SimpleNamespace(**blob["build_args"]))(utt2spk=blob["utt2spk"])
    _strip_parameters(net, device)
    gen, ext = net.hifigan, net.bn_extractor
    gen.precision = blob["generator"]["precision"]
    if len(blob["generator"]["packed"].get("__l__", [])) != len(blob["generator"]["modes"]):
        raise _lib.SatError(f"{path}: packed convolutions and modes differ in number")
    gen.f8_stages = int(blob["generator"].get("f8_stages", gen.f8_stages))
    p8 = _dec(blob["generator"].get("packed8"), device) or {}
    gen._install_packed(_dec(blob["generator"]["packed"], device), blob["generator"]["modes"],
                        ups_grouped=bool(blob["generator"].get("ups_grouped", 0)), packed8={int(k): v for k, v in p8.items()})
    gen.__dict__["_frozen"] = True
    e = blob["extractor"]
    if type(ext).__name__ != e["class"]:
        raise _lib.SatError(f"{path}: extractor {e['class']} does not match the configuration ({type(ext).__name__})")
    ext.precision = e["precision"]
    ext._cache = _dec(e["cache"], device)
    if "cache_full" in e:
        cf = _dec(e["cache_full"], device)
        after = [m for m in ext.tdnnfs_after if isinstance(m, TDNNFBatchNormParams)]
        ext._cache_full = {"after": list(zip(after, cf["after"])),
                           "prefinal": list(zip((ext.prefinal_chain, ext.prefinal_xent), cf["prefinal"])), "out": cf["out"]}
    if "w2" in e:
        ext._w2 = _dec(e["w2"], device)
        ext._mm_mode = e["mm_mode"]
        ext.w2v2_precision = e["w2v2_precision"]
    ext.__dict__["_frozen"] = True
    torch.cuda.synchronize(device)
    net.eval()
    return net
