"""Alias of `transforms.hip.functional_hip` (the module object itself, so every name -- private helpers included -- is shared)."""
import sys

from .hip import functional_hip as _impl

sys.modules[__name__] = _impl
