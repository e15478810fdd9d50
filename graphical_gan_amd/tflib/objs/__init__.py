from . import gan_inference  # noqa: F401
