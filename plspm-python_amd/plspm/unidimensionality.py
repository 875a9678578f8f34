"""Block unidimensionality diagnostics (reference plspm/unidimensionality.py:30-58).

The reference runs sklearn PCA on every standardised block; the same numbers follow from the block's
correlation matrix R_b, which is a sub-block of the treated covariance matrix the device already holds
(``plspm_fit_result_t.cov``): eigenvalues of R_b, Cronbach's alpha = k/(k-1) * 2 sum_{i>j} R_ij / sum_ij R_ij,
Dillon-Goldstein rho from the first principal component's loadings.  Host arithmetic on k x k matrices."""
import numpy as np
import pandas as pd

from plspm.mode import Mode


class Unidimensionality:
    def __init__(self, config, result, incomplete_mvs=()):
        self._config = config
        self._result = result
        self._incomplete = set(incomplete_mvs)        # MVs with missing values: their blocks report NaN (unidimensionality.py:39)

    def summary(self) -> pd.DataFrame:
        cm = self._result.compiled
        cov = self._result.raw["cov"]
        lvs = list(self._config.path())
        out = pd.DataFrame(np.nan, index=lvs, columns=["mode", "mvs", "cronbach_alpha", "dillon_goldstein_rho", "eig_1st", "eig_2nd"])
        out["mode"] = out["mode"].astype(object)
        for l, lv in enumerate(lvs):
            a, b = cm.block_offset[l], cm.block_offset[l + 1]
            k = b - a
            out.loc[lv, "mode"] = self._config.mode(lv).name
            out.loc[lv, "mvs"] = k
            if self._incomplete.intersection(self._config.mvs(lv)):
                continue
            block = cov[a:b, a:b]
            d = np.sqrt(np.diag(block))
            R = block / np.outer(d, d)
            evals, evecs = np.linalg.eigh(R)
            out.loc[lv, "eig_1st"] = evals[-1]
            out.loc[lv, "eig_2nd"] = evals[-2] if k > 1 else np.nan
            if self._config.mode(lv) == Mode.A:
                if k > 1:
                    off = R.sum() - np.trace(R)
                    out.loc[lv, "cronbach_alpha"] = max(0.0, off / R.sum() * (k / (k - 1)))
                load = evecs[:, -1] * np.sqrt(evals[-1])
                num = load.sum() ** 2
                out.loc[lv, "dillon_goldstein_rho"] = num / (num + (k - np.sum(load ** 2)))
        return out
