"""Measurement scales for non-metric data (reference plspm/scale.py:92-104).

The enum is kept so that model specifications written for the reference still parse; the non-metric
(optimal scaling) solver itself is not part of the MI355X hot path yet, so estimating a model that sets
any scale raises ``NotImplementedError``."""
from enum import Enum

from plspm.util import Value


class Scale(Enum):
    NUM = Value(1)
    RAW = Value(2)
    ORD = Value(3)
    NOM = Value(4)
