from .scores import selector_scores
from .selectors import (ActiveSelector, BADGESelector, ConfidenceSelector, CoresetSelector, EntropySelector, KMeanSelector,
                        MarginSelector, RandomSelector, kcenter_greedy)
