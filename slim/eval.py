"""Alias: `slim.eval` IS `slim_amd.eval` (same module object, so attributes patched on one are seen by both)."""
import sys

import slim_amd.eval as _impl

sys.modules[__name__] = _impl
