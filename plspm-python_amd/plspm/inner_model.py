"""Inner (structural) model statistics (reference plspm/inner_model.py).

Path coefficients, R^2 and the direct / indirect / total effects arrive from the device solver
(csrc/solver_core.h ``solve_problem``: normal equations on the LV score covariance, inner_model.py:58-75, and
the matrix-power effects of inner_model.py:33-53).  The regression tables (std error, t, p) are O(L^3) host
arithmetic on the same L x L covariance matrix -- SURVEY.md section 2 keeps them off the GPU.
"""
import numpy as np
import pandas as pd
from scipy import stats


class InnerModel:
    """``InnerModel(path, scores)`` as in the reference, or ``InnerModel.from_device(...)`` (what Plspm uses)."""

    def __init__(self, path: pd.DataFrame, scores: pd.DataFrame = None, _device=None):
        lvs = list(path)
        L = len(lvs)
        C = path.values.astype(int)
        if _device is None:
            # API parity for direct callers: everything from the population covariance of the supplied scores.
            s = scores.loc[:, lvs].values.astype(np.float64)
            n = s.shape[0]
            cov = np.cov(s, rowvar=False, bias=True).reshape(L, L)
            B = np.zeros((L, L))
            r2 = np.zeros(L)
            for i in range(L):
                f = np.flatnonzero(C[i])
                if f.size:
                    B[i, f] = np.linalg.solve(cov[np.ix_(f, f)], cov[f, i])
                    r2[i] = B[i, f] @ cov[f, i] / cov[i, i]
            indirect = np.zeros((L, L))
            if L != 2:
                power = B.copy()
                for _ in range(1, L):
                    power = power @ B
                    indirect += power
            pairs = [(f, t) for f in range(L) for t in range(L) if f != t and (B + indirect)[t, f] != 0]
            direct = np.array([B[t, f] for f, t in pairs])
            ind = np.array([indirect[t, f] for f, t in pairs])
            _device = dict(n=n, path_coef=B, r2=r2, lv_cov=cov, pairs=pairs, direct=direct, indirect=ind, total=direct + ind)
        n = _device["n"]
        B, r2, cov = _device["path_coef"], _device["r2"], _device["lv_cov"]
        self._endogenous = [lv for lv, row in zip(lvs, C) if row.sum() > 0]
        self._path_coefficients = pd.DataFrame(B, index=lvs, columns=lvs)
        self._r_squared = pd.Series(r2, index=lvs, name="r_squared")
        k = C.sum(axis=1)
        adj = np.where(k > 0, 1 - (1 - r2) * (n - 1) / (n - k - 1), 0.0)
        self._r_squared_adj = pd.Series(adj, index=lvs, name="r_squared_adj")
        rows = []
        for i, dv in enumerate(lvs):
            f = np.flatnonzero(C[i])
            if not f.size:
                continue
            dof = n - f.size - 1
            sigma2 = n * cov[i, i] * (1.0 - r2[i]) / dof
            se = np.sqrt(sigma2 * np.diag(np.linalg.inv(n * cov[np.ix_(f, f)])))
            for j, s in zip(f, se):
                t = B[i, j] / s
                rows.append({"from": lvs[j], "to": dv, "estimate": B[i, j], "std error": s, "t": t,
                             "p>|t|": 2.0 * stats.t.sf(abs(t), dof), "index": lvs[j] + " -> " + dv})
        cols = ["from", "to", "estimate", "std error", "t", "p>|t|", "index"]
        self._summaries = pd.DataFrame(rows, columns=cols)
        labels = [lvs[f] + " -> " + lvs[t] for f, t in _device["pairs"]]
        self._effects = pd.DataFrame({"from": [lvs[f] for f, _ in _device["pairs"]], "to": [lvs[t] for _, t in _device["pairs"]],
                                      "direct": _device["direct"], "indirect": _device["indirect"], "total": _device["total"]},
                                     index=labels, columns=["from", "to", "direct", "indirect", "total"])

    @classmethod
    def from_device(cls, path: pd.DataFrame, result):
        raw, native = result.raw, result.native
        pairs = list(zip(native.eff_from.tolist(), native.eff_to.tolist()))
        dev = dict(n=native.N, path_coef=raw["path_coef"], r2=raw["r2"], lv_cov=raw["lv_cov"], pairs=pairs, direct=raw["direct"],
                   indirect=raw["indirect"], total=raw["total"])
        return cls(path, None, _device=dev)

    def path_coefficients(self) -> pd.DataFrame:
        return self._path_coefficients

    def r_squared(self) -> pd.Series:
        return self._r_squared

    def r_squared_adj(self) -> pd.Series:
        return self._r_squared_adj

    def inner_model(self) -> pd.DataFrame:
        return self._summaries.set_index(["index"])

    def effects(self) -> pd.DataFrame:
        return self._effects

    def endogenous(self) -> list:
        return self._endogenous
