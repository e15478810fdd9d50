from . import linear, conv2d, deconv2d, conv3d, batchnorm, act  # noqa: F401
