from slim_amd.prune import *  # noqa: F401,F403
from slim_amd import prune as _impl


def __getattr__(name):
    return getattr(_impl, name)
