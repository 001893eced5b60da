from . import wav2vec2  # noqa: F401
