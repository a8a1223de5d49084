"""Alias: `slim.quantization.utils` IS `slim_amd.quantization.utils` (same module object)."""
import sys

import slim_amd.quantization.utils as _impl

sys.modules[__name__] = _impl
