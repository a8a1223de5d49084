from slim_amd.quantization.quantization import *  # noqa: F401,F403
from slim_amd.quantization import quantization as _impl


def __getattr__(name):
    return getattr(_impl, name)
