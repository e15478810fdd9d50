from . import gan_inference, mmd, kl_aggregated  # noqa: F401
