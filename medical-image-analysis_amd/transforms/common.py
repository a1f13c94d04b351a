"""Alias of `transforms.hip.common` (the module object itself, so every name -- private helpers included -- is shared)."""
import sys

from .hip import common as _impl

sys.modules[__name__] = _impl
