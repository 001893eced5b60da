"""Minimal stand-in for the `torchaudio` package, used ONLY by tests/golden/make_fixtures.py
when it imports the reference (SA-toolkit) in the build container, where torchaudio is not
installed and cannot be fetched (no network).  It never ships with the product and is never
imported on the GPU box.  What it restates (published torchaudio 2.1 algorithms, un-vendored
third-party code => "parity unpinned" for these rows):
  * functional.lowpass_biquad / highpass_biquad  (RBJ cookbook biquads through lfilter)
  * models.wav2vec2.model.wav2vec2_model         (installed by make_fixtures.py from oracle/)
"""
from . import functional, models, transforms  # noqa: F401


def load(*a, **k):
    raise RuntimeError("torchaudio stand-in: load() is not available")


def save(*a, **k):
    raise RuntimeError("torchaudio stand-in: save() is not available")
