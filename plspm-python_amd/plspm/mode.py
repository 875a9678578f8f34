"""Outer-weight modes (reference plspm/mode.py:64-69).

A Mode member's ``code`` is what the HIP solver receives per latent variable (``PLSPM_MODE_A`` / ``PLSPM_MODE_B`` in
include/plspm_hip.h).  Inside a fit the arithmetic -- Mode A ``w_k = X_k' z_k / N`` (mode.py:28-29) and Mode B
``w_k = argmin |X_k w - z_k|`` (mode.py:50-52, minimum norm for a rank-deficient block) -- is part of the fused solver kernel
(csrc/solver_core.h ``iterate``).  The members also keep the reference's plug-in method
``Mode.X.value.outer_weights_metric(data, Z, lv, mvs)``: one device call (``plspm_op_outer_weights``: upload of the block and z,
MFMA Gram, one small kernel) returning the reference's k x 1 DataFrame (index = mvs, column = lv).  The non-metric variant
(``outer_weights_nonmetric``, mode.py:31-42, 54-61) exists only fused inside the non-metric solver kernels."""
from enum import Enum

import pandas as pd

from plspm.util import Value


class _OuterMode(Value):
    def __init__(self, tag, code):
        super().__init__(tag)
        self.code = code

    def outer_weights_metric(self, data: pd.DataFrame, Z: pd.DataFrame, lv: str, mvs: list, device_id: int = 0) -> pd.DataFrame:
        from plspm import _native
        w = _native.op_outer_weights(self.code, data.loc[:, mvs].values, Z.loc[:, lv].values, device_id)
        return pd.DataFrame(w, columns=[lv], index=mvs)

    def outer_weights_nonmetric(self, *args, **kwargs):
        raise NotImplementedError("the non-metric outer-weight step runs only fused inside the device solver (csrc/solver_core.h nm_step)")


class Mode(Enum):
    """Whether a latent variable is reflective (A) or formative (B) with respect to its manifest variables."""
    A = _OuterMode("A", 0)
    B = _OuterMode("B", 1)
