from slim_amd.quantization.utils import *  # noqa: F401,F403
from slim_amd.quantization import utils as _impl


def __getattr__(name):
    return getattr(_impl, name)
