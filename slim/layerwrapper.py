"""Alias: `slim.layerwrapper` IS `slim_amd.layerwrapper` (same module object, so attributes patched on one are seen by both)."""
import sys

import slim_amd.layerwrapper as _impl

sys.modules[__name__] = _impl
