"""Alias: `slim.data` IS `slim_amd.data` (same module object, so attributes patched on one are seen by both)."""
import sys

import slim_amd.data as _impl

sys.modules[__name__] = _impl
