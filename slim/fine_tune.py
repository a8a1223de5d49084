"""Alias: `slim.fine_tune` IS `slim_amd.fine_tune` (same module object, so attributes patched on one are seen by both)."""
import sys

import slim_amd.fine_tune as _impl

sys.modules[__name__] = _impl
