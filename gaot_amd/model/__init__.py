from .gaot import GAOT  # noqa: F401
