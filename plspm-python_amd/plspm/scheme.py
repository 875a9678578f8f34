"""Inner weighting schemes (reference plspm/scheme.py:57-63).

``code`` is the scheme id of the C-ABI (``PLSPM_SCHEME_*``).  The operators -- centroid
``sign(corr(Y) * (C + C'))`` (scheme.py:27-28), factorial ``cov(Y) * (C + C')`` (scheme.py:36-37) and path
(OLS on predecessors, correlations with successors, scheme.py:45-54) -- are evaluated on the device from the
latent covariance matrix (csrc/solver_core.h ``inner_weights``)."""
from enum import Enum

from plspm.util import Value


class _InnerScheme(Value):
    def __init__(self, tag, code):
        super().__init__(tag)
        self.code = code


class Scheme(Enum):
    """The scheme used to calculate inner weights."""
    CENTROID = _InnerScheme("C", 0)
    PATH = _InnerScheme("P", 2)
    FACTORIAL = _InnerScheme("F", 1)
