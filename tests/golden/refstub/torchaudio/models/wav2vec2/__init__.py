from . import model, utils  # noqa: F401
