"""Alias: `slim.prune` IS `slim_amd.prune` (same module object, so attributes patched on one are seen by both)."""
import sys

import slim_amd.prune as _impl

sys.modules[__name__] = _impl
