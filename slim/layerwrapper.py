from slim_amd.layerwrapper import *  # noqa: F401,F403
from slim_amd import layerwrapper as _impl


def __getattr__(name):
    return getattr(_impl, name)
