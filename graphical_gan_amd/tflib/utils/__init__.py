from . import distance  # noqa: F401
