from . import gan_inference, mmd  # noqa: F401
