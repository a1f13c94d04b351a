"""Alias of `transforms.hip.gpu_pipeline` (the module object itself, so every name -- private helpers included -- is shared)."""
import sys

from .hip import gpu_pipeline as _impl

sys.modules[__name__] = _impl
