from slim_amd.data import *  # noqa: F401,F403
from slim_amd import data as _impl


def __getattr__(name):
    return getattr(_impl, name)
