"""Seeded synthetic weights, speakers and utterances.

Released weights and datasets cannot be fetched offline, so tests, fixtures and the benchmark
use (a) random weights drawn per state-dict key from a seeded generator, scaled so that
activations stay O(1) like a trained model's, optionally conditioned by a small committed file
(BatchNorm running statistics and a VQ codebook sampled from the bottleneck layer's own input:
with un-conditioned random weights the VQ collapses to one code), and (b) the
synthetic utterances of SURVEY §8d (`harm`: a harmonic source with a moving f0; `rand`:
torch.rand, the reference README's smoke input).
"""
import math
import os
import re
import zlib

import numpy as np
import torch

ASR_DIR = "../../asr/librispeech/exp/chain/{name}/final.pt"
N_SPEAKERS = 247


def utt2spk(n=N_SPEAKERS):
    """synthetic utterance->speaker map; ids chosen so that string order != numeric order"""
    return {f"utt{i}": str(100 + 37 * i) for i in range(n)}


def parse_tag(tag):
    tag = re.sub(r"_v\d+$", "", tag)
    if tag.startswith("hifigan_"):
        return "anonymizer", tag[len("hifigan_"):]
    return "asrbn", tag


def _gen(key, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) + 7919 * int(seed)) & 0x7FFFFFFF)
    return g


_UP_RATES = [5, 4, 4, 2, 2]


def _generator_gain(key, v):
    """weight_g = gain * ||v||, i.e. effective weight = gain * v.  v keeps the reference's init scale
    (N(0, 0.01), hifigan/nn.py:11-14); the gain sets a trained-like effective scale so that the
    signal neither vanishes nor saturates through 5 upsampling stages (target output RMS ~0.1)."""
    if ".ups." in key or key.startswith("ups."):
        i = int(re.search(r"ups\.(\d+)\.", key).group(1))
        c_in, _, k = v.shape
        return 1.3 / math.sqrt(c_in * k / _UP_RATES[i]) / 0.01
    if "resblocks." in key or "conv_post." in key:
        _, c_in, k = v.shape
        return 0.6 / math.sqrt(c_in * k) / 0.01
    return 1.5  # conv_pre (kaiming-uniform v)


def fill_state_dict(sd, seed=0):
    """draw every tensor of `sd` (reference key names) in place; deterministic per (key, seed)"""
    out = {}
    for k, v in sd.items():
        g = _gen(k, seed)
        shape = tuple(v.shape)
        if v.dtype != torch.float32:
            out[k] = torch.zeros_like(v)
            continue
        if k.endswith("weight_v"):
            if ".conv_pre." in k or k.startswith("conv_pre."):
                fan_in = shape[1] * shape[2]
                t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
            else:
                t = torch.randn(shape, generator=g) * 0.01
        elif k.endswith("weight_g"):
            t = None  # set from the matching weight_v below
        elif k.endswith("running_var"):
            t = torch.ones(shape)
        elif k.endswith("running_mean") or k.endswith("_ema_cluster_size"):
            t = torch.zeros(shape)
        elif "linearB.inner_nat.weight" in k:
            t = torch.randn(shape, generator=g) / math.sqrt(shape[1])
        elif "linearB.inner_nat.bias" in k:
            t = torch.randn(shape, generator=g) * 0.1
        elif "linearA.weight" in k:
            t = torch.randn(shape, generator=g) / math.sqrt(shape[1])
        elif "_embedding.weight" in k or "_ema_w" in k:
            t = torch.randn(shape, generator=g)
        elif k.endswith("bias"):
            t = (torch.rand(shape, generator=g) * 2 - 1) * 0.05
        elif k.endswith("layer_norm.weight") or k.endswith("final_layer_norm.weight"):
            t = 1.0 + 0.05 * torch.randn(shape, generator=g)
        elif v.dim() >= 2:
            fan_in = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        else:
            t = torch.randn(shape, generator=g) * 0.05
        out[k] = t
    for k in list(out):
        if k.endswith("weight_g"):
            v = out[k[:-1] + "v"]
            if tuple(sd[k].shape[:2]) == (1, 1) and v.dim() == 3 and v.shape[0] > 1:
                out[k] = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()    # weight_norm(dim=2): wav2vec2 positional conv
                continue
            norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(sd[k].shape)
            out[k] = norm * _generator_gain(k, v)
    # the VQ module is registered twice in the reference; both copies hold the same tensors
    for k in list(out):
        if ".tdnn.bottleneck_func." in k:
            out[k] = out[k.replace(".tdnn.bottleneck_func.", ".bottleneck_func.")]
    return out


def apply_conditioning(sd, cond):
    """overwrite BatchNorm statistics / codebook with calibrated values (keys = state-dict keys,
    optionally relative to the ASR-BN net)"""
    for k, v in cond.items():
        t = torch.as_tensor(np.asarray(v))
        for full in (k, "bn_extractor." + k):
            if full in sd:
                sd[full] = t.to(sd[full].dtype).reshape(sd[full].shape)
    return sd


def conditioning_path(asr_name):
    here = os.path.dirname(os.path.abspath(__file__))
    return os.path.join(here, "..", "tests", "golden", f"conditioning_{asr_name}.npz")


# (tag, seed, conditioning) -> filled state dict: drawing 327 M values takes ~15 s for the wav2vec2 tag, and bench.py /
# the tests build the same synthetic checkpoint several times per process.  ONE entry is kept (1.3 GB for the wav2vec2
# tag); callers get the SAME tensors in a fresh dict and must treat them as read-only (load_state_dict copies them);
# SATOOLS_AMD_SYNTH_CACHE=0 disables the cache.
_FILLED = {}
_FILLED_MAX = 1 if os.environ.get("SATOOLS_AMD_SYNTH_CACHE", "1") != "0" else 0


def checkpoint(tag, seed=0, conditioning="auto"):
    """a reference-format checkpoint dict for `tag` with synthetic weights"""
    from . import infer_helper
    kind, asr_name = parse_tag(tag)
    if kind == "anonymizer":
        state = {"install_path": "", "task_path": "/egs/vc/libritts", "base_model_path": "local/tuning/hifigan.py",
                 "base_model_params": {"utt2spk": utt2spk()},
                 "base_model_args": {"asrbn_model": ASR_DIR.format(name=asr_name), "f0_transformation": ""}}
    else:
        state = dict(infer_helper.asrbn_conf_from_name(ASR_DIR.format(name=asr_name)))
        state["install_path"] = ""
    build = infer_helper._builder(state["base_model_path"])
    from .anonymizer import SimpleNamespace
    net = build(SimpleNamespace(**state["base_model_args"]))(**state["base_model_params"])
    if conditioning == "auto":
        p = conditioning_path(asr_name)
        conditioning = p if os.path.exists(p) else None
    ck = (tag, int(seed), conditioning)
    sd = _FILLED.get(ck)
    if sd is None:
        sd = fill_state_dict(net.state_dict(), seed)
        if conditioning:
            apply_conditioning(sd, dict(np.load(conditioning)))
        if _FILLED_MAX:
            while len(_FILLED) >= _FILLED_MAX:
                _FILLED.pop(next(iter(_FILLED)))
            _FILLED[ck] = sd
    state["base_model_state_dict"] = dict(sd)     # same tensors (read-only by convention), a fresh dict
    return state, net


def load(spec, option_args=None):
    """`synthetic:<tag>[?seed=N]` -> model with seeded weights (see infer_helper.load_model)"""
    from . import infer_helper
    from .anonymizer import SimpleNamespace
    body = spec[len("synthetic:"):]
    tag, _, query = body.partition("?")
    seed = 0
    for kv in filter(None, query.split("&")):
        k, _, v = kv.partition("=")
        if k == "seed":
            seed = int(v)
    state, net = checkpoint(tag, seed)
    if option_args:
        args = dict(state["base_model_args"])
        args.update(option_args)
        net = infer_helper._builder(state["base_model_path"])(SimpleNamespace(**args))(**state["base_model_params"])
    net.load_state_dict(state["base_model_state_dict"])
    return net


# ---- synthetic utterances -------------------------------------------------------
def harm_utterance(seed, n=80000, sr=16000):
    """7 harmonics of f0(t) = (100 + 5*(seed mod 16)) + 60*sin(2*pi*0.7*t), 1/k amplitudes x0.3,
    on/off envelope sin(2*pi*1.5*t) > -0.3, + 0.01*N(0,1) from manual_seed(seed), whole signal x0.5"""
    t = torch.arange(n, dtype=torch.float64) / sr
    f0 = (100 + 5 * (seed % 16)) + 60 * torch.sin(2 * math.pi * 0.7 * t)
    phase = 2 * math.pi * torch.cumsum(f0, 0) / sr
    sig = torch.zeros(n, dtype=torch.float64)
    for k in range(1, 8):
        sig += torch.sin(k * phase) / k
    sig *= 0.3
    env = (torch.sin(2 * math.pi * 1.5 * t) > -0.3).to(torch.float64)
    g = torch.Generator().manual_seed(int(seed))
    noise = torch.randn(n, generator=g, dtype=torch.float32).to(torch.float64)
    return ((sig * env + 0.01 * noise) * 0.5).to(torch.float32)


def harm_batch(seeds, n=80000):
    return torch.stack([harm_utterance(s, n) for s in seeds])


def rand_batch(seed, B, n=80000):
    return torch.rand((B, n), generator=torch.Generator().manual_seed(int(seed)))


def targets(spk, idx):
    """target speaker of utterance i: spk[(7*i) mod len(spk)]"""
    return [spk[(7 * i) % len(spk)] for i in idx]


def xvector_state(seed=0, num_speakers=10):
    """seeded state dict of the x-vector extractor (reference key names): conv / linear weights N(0, 1.5/fan_in),
    BatchNorm statistics and affine terms away from the identity so that every term of the arithmetic is exercised"""
    from . import xvector
    net = xvector.build()(num_speakers=num_speakers)
    sd = {}
    for i, (k, v) in enumerate(net.state_dict().items()):
        g = torch.Generator().manual_seed(int(seed) * 100003 + i)
        if k.endswith("num_batches_tracked") or k.startswith("preprocessor."):
            sd[k] = v.clone()
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith("running_mean"):
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif ".bn" in k and k.endswith("weight"):
            sd[k] = 0.8 + 0.4 * torch.rand(v.shape, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        else:
            fan_in = v[0].numel() if v.dim() > 1 else v.numel()
            sd[k] = torch.randn(v.shape, generator=g) * math.sqrt(1.5 / fan_in)
    return sd
