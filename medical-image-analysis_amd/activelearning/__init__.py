"""`activelearning` of the MI355X drop-in: same eight names the reference exports (`src/activelearning/__init__.py`)."""
from mia_hip.dropin import extend_over_reference

__path__ = extend_over_reference(__path__, __name__)

from .scores import selector_scores  # noqa: E402
from .selectors import (ActiveSelector, BADGESelector, ConfidenceSelector, CoresetSelector, EntropySelector,  # noqa: E402
                        KMeanSelector, MarginSelector, RandomSelector, kcenter_greedy)
