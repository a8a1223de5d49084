"""Alias: `slim.lora` IS `slim_amd.lora` (same module object, so attributes patched on one are seen by both)."""
import sys

import slim_amd.lora as _impl

sys.modules[__name__] = _impl
