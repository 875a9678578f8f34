"""Outer-weight modes (reference plspm/mode.py:64-69).

A Mode member's ``code`` is what the HIP solver receives per latent variable (``PLSPM_MODE_A`` / ``PLSPM_MODE_B`` in
include/plspm_hip.h).  Inside a fit the arithmetic -- Mode A ``w_k = X_k' z_k / N`` (mode.py:28-29) and Mode B
``w_k = argmin |X_k w - z_k|`` (mode.py:50-52, minimum norm for a rank-deficient block) -- is part of the fused solver kernel
(csrc/solver_core.h ``iterate``).  The members also keep the reference's plug-in method
``Mode.X.value.outer_weights_metric(data, Z, lv, mvs)``: one device call (``plspm_op_outer_weights``: upload of the block and z,
MFMA Gram, one small kernel) returning the reference's k x 1 DataFrame (index = mvs, column = lv), and the non-metric variant
``Mode.X.value.outer_weights_nonmetric(mv_grouped_by_lv, mv_grouped_by_lv_missing, Z, lv, correction)`` (mode.py:31-42, 54-61; one device
call, ``plspm_op_outer_weights_nonmetric``) returning the reference's ``(weights, Y)`` pair."""
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

    def outer_weights_nonmetric(self, mv_grouped_by_lv, mv_grouped_by_lv_missing, Z, lv: str, correction: float, device_id: int = 0):
        """(weights [k], Y [N]) of the LV's block: ``mv_grouped_by_lv[lv]`` the quantified MVs (N x k, NaN where missing),
        ``mv_grouped_by_lv_missing[lv]`` their 0/1 presence mask when the block has missing cells (the reference keeps the key only then)."""
        from plspm import _native
        present = mv_grouped_by_lv_missing[lv] if lv in mv_grouped_by_lv_missing else None
        if present is not None and self.code == 1:
            raise Exception("Missing nonmetric data is not supported in mode B. LV with missing data: " + lv)         # mode.py:55-56
        return _native.op_outer_weights_nonmetric(self.code, mv_grouped_by_lv[lv], present, Z, correction, device_id)


class Mode(Enum):
    """Whether a latent variable is reflective (A) or formative (B) with respect to its manifest variables."""
    A = _OuterMode("A", 0)
    B = _OuterMode("B", 1)
