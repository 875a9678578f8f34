"""Measurement scales for non-metric data (reference plspm/scale.py:92-104).

Setting a scale on any MV (or a ``default_scale``) selects the non-metric solver, as in the reference: NUM / RAW run the
correlation-matrix solver, ORD / NOM the optimal-scaling solver on indicator columns.  The per-scale
arithmetic (scale.py:22-89 of the reference) lives in the device solver sources, not in this enum."""
from enum import Enum

from plspm.util import Value


class Scale(Enum):
    NUM = Value(1)
    RAW = Value(2)
    ORD = Value(3)
    NOM = Value(4)
