from . import import_fairseq  # noqa: F401
