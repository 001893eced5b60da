"""CPU oracle of the SA-toolkit `anonymize` / `model.convert()` hot path.

TEST INFRASTRUCTURE ONLY.  This package is a from-scratch CPU restatement (plain PyTorch f32 ops
and numpy; the path is floating point) of the reference algorithms; each function cites the
reference file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import it — as the checker, never as the thing measured or shipped.  The
product package (`sa-toolkit_amd/`) never imports it and has no CPU fallback.

Pinning: the reference ships no tests or golden vectors for this path, so the oracle
is pinned against outputs of the reference itself, produced by importing it in the build
container (tests/golden/make_fixtures.py) and committed as small fixtures under tests/golden/.
Exceptions ("parity unpinned", third-party code absent from /root/reference): torchaudio's
biquad filters inside YAAPT and torchaudio's wav2vec2 model — restated from the published
torchaudio 2.1 algorithms (see oracle/biquad.py, oracle/wav2vec2.py).
"""
