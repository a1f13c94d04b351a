"""Alias of `transforms.hip.normalization` (the module object itself, so every name -- private helpers included -- is shared)."""
import sys

from .hip import normalization as _impl

sys.modules[__name__] = _impl
