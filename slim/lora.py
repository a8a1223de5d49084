from slim_amd.lora import *  # noqa: F401,F403
from slim_amd import lora as _impl


def __getattr__(name):
    return getattr(_impl, name)
