"""satools_amd — MI355X-native `anonymize` / `model.convert()` hot path of SA-toolkit.

Host side (this package) mirrors the reference's Python interface; all arithmetic runs in
libsatools_hip.so (hand-written HIP for gfx950, see csrc/ and include/satools_hip.h)."""
from . import _lib, infer_helper  # noqa: F401
from .infer_helper import load_model  # noqa: F401
from .frozen import export_frozen, load_frozen  # noqa: F401

__all__ = ["load_model", "infer_helper", "export_frozen", "load_frozen"]
