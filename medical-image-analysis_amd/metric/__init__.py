"""`metric` of the MI355X drop-in: `metric.segmentation` (argmax + hard Dice on the GPU) is this repo's; the reference's
`metric/metric.py` (Hausdorff distance through SimpleITK, `src/metric/metric.py`) stays the reference's and is reached
through the merged package path (mia_hip/dropin.py).  `from metric import cal_hd` (`src/training/al_trainer.py:84`)
keeps working: the name is forwarded when the reference and SimpleITK are present, and raises at CALL time otherwise."""
from mia_hip.dropin import extend_over_reference

__path__ = extend_over_reference(__path__, __name__)

try:
    from .metric import cal_hd  # noqa: F401  (the reference's module, behind this repo on sys.path)
except ImportError as _e:  # reference or SimpleITK absent: Hausdorff distance is host-side work outside the hot path
    _why = str(_e)

    def cal_hd(*args, **kwargs):
        raise ImportError(f"metric.cal_hd is the reference's SimpleITK-based Hausdorff metric and is not available here ({_why})")
