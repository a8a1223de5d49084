"""Alias: `slim.quantization.quantization` IS `slim_amd.quantization.quantization` (same module object)."""
import sys

import slim_amd.quantization.quantization as _impl

sys.modules[__name__] = _impl
