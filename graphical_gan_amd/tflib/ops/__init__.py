from . import linear, conv2d, deconv2d, batchnorm, act  # noqa: F401
