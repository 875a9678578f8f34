"""Outer-weight modes (reference plspm/mode.py:64-69).

On this backend a Mode member is a *descriptor*: its ``code`` is what the HIP solver receives per latent
variable (``PLSPM_MODE_A`` / ``PLSPM_MODE_B`` in include/plspm_hip.h).  The arithmetic itself --
Mode A ``w_k = X_k' z_k / N`` (mode.py:28-29) and Mode B ``w_k = argmin |X_k w - z_k|`` (mode.py:50-52) --
lives in csrc/solver_core.h (``iterate``)."""
from enum import Enum

from plspm.util import Value


class _OuterMode(Value):
    def __init__(self, tag, code):
        super().__init__(tag)
        self.code = code


class Mode(Enum):
    """Whether a latent variable is reflective (A) or formative (B) with respect to its manifest variables."""
    A = _OuterMode("A", 0)
    B = _OuterMode("B", 1)
