"""CPU emulation (float64 sums) of the operand decompositions the generator's matrix products can
run in, on the golden generator input: how far is each from the exact-f32-operand result?
  f16x3   : hi*hi + hi*lo + lo*hi, hi/lo f16                      (shipped SAT_CONV_F16X3)
  f16+f8  : hi*hi in f16; cross terms hi*lo + lo*hi with e4m3 operands (power-of-two scales)
  f16+f6b : cross terms in fp6 e2m3 with one power-of-two scale per 32 K-elements (MX block scale)
  f16x2   : hi*hi + lo*hi  (activations rounded to f16)
  f16x1   : hi*hi
Diagnostic only (imports the oracle): run from the repo root as `python tests/diagnostics/precision_emulation.py`."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import convert as oconv, hifigan as ohg   # noqa: E402
import satools_amd                                        # noqa: E402
from satools_amd import synthetic                         # noqa: E402


def split16(x):
    hi = x.to(torch.float16).to(torch.float64)
    lo = (x.to(torch.float64) - hi).to(torch.float16).to(torch.float64)
    return hi, lo


def q8(x, scale_pow):
    s = 2.0 ** scale_pow
    return (x * s).clamp(-448, 448).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64) / s


def q6_block(x, dim):
    """fp6 e2m3 (max 7.5, 3 mantissa bits, subnormal step 0.125) with a power-of-two scale per 32
    elements along `dim` chosen from the block's max"""
    xm = x.movedim(dim, -1)
    shp = xm.shape
    n = shp[-1]
    pad = (-n) % 32
    xp = F.pad(xm, (0, pad)).reshape(*shp[:-1], -1, 32)
    mx = xp.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    e = torch.ceil(torch.log2(mx / 7.5))
    s = 2.0 ** e
    v = xp / s
    a = v.abs()
    ex = torch.floor(torch.log2(a.clamp_min(1e-300))).clamp(min=0, max=2)     # normal exponents 0..2 (1..7.5); below 1: subnormal step 0.125
    step = 2.0 ** (ex - 3)
    q = torch.round(a / step) * step
    q = torch.sign(v) * q.clamp(max=7.5)
    out = (q * s).reshape(*shp[:-1], -1)[..., :n]
    return out.movedim(-1, dim)


class Scheme:
    def __init__(self, name):
        self.name = name

    def conv(self, fn, x, w, b, **kw):
        x64, w64 = x.to(torch.float64), w.to(torch.float64)
        b64 = b.to(torch.float64)
        if self.name == "exact":
            return fn(x64, w64, b64, **kw).to(torch.float32)
        xh, xl = split16(x)
        wh, wl = split16(w)
        cdim = 1   # channel (K) axis of x; of w it is 1 for conv1d (out, in, k) and 0 for conv_transpose1d (in, out, k)
        wk = 0 if fn is F.conv_transpose1d else 1
        y = fn(xh, wh, b64, **kw)
        if self.name == "f16x3":
            y = y + fn(xl, wh, None, **kw) + fn(xh, wl, None, **kw)
        elif self.name == "f16+f8":
            y = y + fn(q8(xl, 10), q8(wh, 6), None, **kw) + fn(q8(xh, 0), q8(wl, 16), None, **kw)
        elif self.name == "f16+f6b":
            y = y + fn(q6_block(xl, cdim), q6_block(wh, wk), None, **kw) + fn(q6_block(xh, cdim), q6_block(wl, wk), None, **kw)
        elif self.name == "f16x2":
            y = y + fn(xh, wl, None, **kw)
        elif self.name == "f16x1":
            pass
        else:
            raise ValueError(self.name)
        return y.to(torch.float32)


def run(scheme, gen_sd, f0, bn, spk):
    sc = Scheme(scheme)
    real_c, real_t = F.conv1d, F.conv_transpose1d

    class Fpatched:
        def __getattr__(self, k):
            return getattr(F, k)

        def conv1d(self, x, w, b=None, **kw):
            if w.shape[0] == 1:                    # conv_post runs in exact f32 on the device
                return real_c(x, w, b, **kw)
            return sc.conv(real_c, x, w, b, **kw)

        def conv_transpose1d(self, x, w, b=None, **kw):
            return sc.conv(real_t, x, w, b, **kw)

    ohg.F = Fpatched()
    try:
        return oconv.forward(gen_sd, f0.clone(), bn, spk)
    finally:
        ohg.F = F


def main():
    torch.set_grad_enabled(False)
    tag = "hifigan_bn_tdnnf_600h_vq_48_v1"
    state, _ = synthetic.checkpoint(tag)
    _, gen_sd = oconv.split_state_dict(state["base_model_state_dict"])
    fx = np.load(os.path.join(ROOT, "tests", "golden", "fx_gen.npz"))
    model = satools_amd.load_model("synthetic:" + tag)
    spk = F.one_hot(torch.from_numpy(fx["spk_argmax"]), len(model.spk))
    f0, bn = torch.from_numpy(fx["f0_raw"].copy()), torch.from_numpy(fx["bn"])
    ref = run("exact", gen_sd, f0, bn, spk).double()
    print(f"signal RMS {ref.pow(2).mean().sqrt():.4f};  vs golden reference waveform: {(ref - torch.from_numpy(fx['y']).double()).pow(2).mean().sqrt():.2e}")
    for s in ("f16x3", "f16+f8", "f16+f6b", "f16x2", "f16x1"):
        y = run(s, gen_sd, f0, bn, spk).double()
        print(f"{s:8s}: RMS error {(y - ref).pow(2).mean().sqrt():.3e}   max {(y - ref).abs().max():.3e}")


if __name__ == "__main__":
    main()
