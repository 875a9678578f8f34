"""Inner weighting schemes (reference plspm/scheme.py:57-63).

``code`` is the scheme id of the C-ABI (``PLSPM_SCHEME_*``).  Inside a fit the operators -- centroid
``sign(corr(Y) * (C + C'))`` (scheme.py:27-28), factorial ``cov(Y) * (C + C')`` (scheme.py:36-37) and path
(OLS on predecessors, correlations with successors, scheme.py:45-54) -- are evaluated by the fused solver kernel from the latent
covariance matrix (csrc/solver_core.h ``inner_weights``).  The members also keep the reference's plug-in method
``Scheme.X.value.calculate(path, y)``: one device call (``plspm_op_inner_weights``: upload of y, MFMA Gram, the same
``inner_weights`` code), returning what the reference returns -- a DataFrame labelled like ``path`` for centroid / factorial, an
ndarray for path."""
from enum import Enum

import numpy as np
import pandas as pd

from plspm.util import Value


class _InnerScheme(Value):
    def __init__(self, tag, code):
        super().__init__(tag)
        self.code = code

    def calculate(self, path: pd.DataFrame, y: np.ndarray, device_id: int = 0):
        """Inner weights E [L x L] for the LV scores ``y`` [N x L] (columns in the order of ``path``)."""
        from plspm import _native
        E = _native.op_inner_weights(self.code, path.values, np.asarray(y, dtype=np.float64), device_id)
        if self.code == 2:                                   # the reference's path calculator returns a bare ndarray (scheme.py:54)
            return E
        return pd.DataFrame(E, index=path.index, columns=path.columns)


class Scheme(Enum):
    """The scheme used to calculate inner weights."""
    CENTROID = _InnerScheme("C", 0)
    PATH = _InnerScheme("P", 2)
    FACTORIAL = _InnerScheme("F", 1)
