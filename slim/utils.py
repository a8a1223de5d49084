"""Alias: `slim.utils` IS `slim_amd.utils` (same module object, so attributes patched on one are seen by both)."""
import sys

import slim_amd.utils as _impl

sys.modules[__name__] = _impl
