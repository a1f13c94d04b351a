"""`scheduler` of the MI355X drop-in.  Modules this repo does not define (the reference's other files under
`src/scheduler/`) keep resolving to the reference when it sits later on sys.path -- see mia_hip/dropin.py."""
from mia_hip.dropin import extend_over_reference

__path__ = extend_over_reference(__path__, __name__)
