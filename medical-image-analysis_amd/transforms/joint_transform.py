"""Alias of `transforms.hip.joint_transform` (the module object itself, so every name -- private helpers included -- is shared)."""
import sys

from .hip import joint_transform as _impl

sys.modules[__name__] = _impl
