from slim_amd.utils import *  # noqa: F401,F403
from slim_amd import utils as _impl


def __getattr__(name):
    return getattr(_impl, name)
