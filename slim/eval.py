from slim_amd.eval import *  # noqa: F401,F403
from slim_amd import eval as _impl


def __getattr__(name):
    return getattr(_impl, name)
