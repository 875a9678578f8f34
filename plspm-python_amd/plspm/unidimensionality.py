"""Block unidimensionality diagnostics (reference plspm/unidimensionality.py:30-58).

The reference runs sklearn PCA on every standardised block; the same numbers follow from the block's
correlation matrix R_b, which is a sub-block of the treated covariance matrix the device already holds
(``plspm_fit_result_t.cov``): eigenvalues of R_b, Cronbach's alpha = k/(k-1) * 2 sum_{i>j} R_ij / sum_ij R_ij,
Dillon-Goldstein rho from the first principal component's loadings.  Host arithmetic on k x k matrices.

Data with Scale.ORD / NOM columns are the exception: the reference standardises the FILTERED data of the block (unidimensionality.py:40: the raw category codes), not the
quantified MVs the estimate ended on, so the device's covariance -- that of the quantified MVs -- is the wrong matrix there (1-2 % off on five- to ten-point items: golden g17);
such models hand the observations in and the block's correlation matrix is taken from them."""
import numpy as np
import pandas as pd

from plspm.mode import Mode


class Unidimensionality:
    def __init__(self, config, result, incomplete_mvs=(), raw=None):
        self._config = config
        self._result = result
        self._incomplete = set(incomplete_mvs)        # MVs with missing values: their blocks report NaN (unidimensionality.py:39)
        self._raw = raw                               # the filtered observations of a model with Scale.ORD / NOM columns (else None: the device covariance serves)

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
            mvs = list(self._config.mvs(lv))
            if self._raw is not None and all(mv in self._raw.columns for mv in mvs):
                R = np.corrcoef(self._raw[mvs].values.astype(np.float64), rowvar=False).reshape(k, k) if k > 1 else np.ones((1, 1))
            else:
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
