"""Inner-model summary table and goodness-of-fit.

Mirrors what reference plspm/inner_summary.py:26-65 reports -- per LV: type, R^2, adjusted R^2, block communality,
mean redundancy, AVE; overall: GoF = sqrt(weighted mean communality * mean R^2 of the endogenous LVs).  All of it is
O(P) host arithmetic on the outer-model table built from device outputs, so it stays off the GPU."""
import math

import numpy as np
import pandas as pd

from plspm.mode import Mode


def _block_stats(config, outer, lv):
    rows = outer.loc[config.mvs(lv)]
    comm = rows["communality"].to_numpy(dtype=float)
    red = rows["redundancy"].to_numpy(dtype=float)
    ave = comm.sum() / (comm.sum() + (1.0 - comm).sum()) if config.mode(lv) == Mode.A else np.nan
    return comm.mean(), red.mean(), ave, comm.size


class InnerSummary:
    def __init__(self, config, r_squared: pd.Series, r_squared_adj: pd.Series, outer_model: pd.DataFrame):
        structure = config.path()
        names = list(structure)
        has_predecessor = structure.sum(axis=1).astype(bool)
        table = {"type": ["Endogenous" if has_predecessor[lv] else "Exogenous" for lv in names]}
        stats = [_block_stats(config, outer_model, lv) for lv in names]
        table["r_squared"] = [r_squared[lv] for lv in names]
        table["r_squared_adj"] = [r_squared_adj[lv] for lv in names]
        table["block_communality"] = [s[0] for s in stats]
        table["mean_redundancy"] = [s[1] for s in stats]
        table["ave"] = [s[2] for s in stats]
        self._table = pd.DataFrame(table, index=names).sort_index()
        # GoF: communality averaged over multi-item blocks (weighted by block size) times the mean R^2 of endogenous LVs
        multi = [(s[0], s[3]) for s in stats if s[3] > 1]
        if multi:
            weight = float(sum(k for _, k in multi))
            communality = sum(c * k for c, k in multi) / weight
            r2 = np.array([r_squared[lv] for lv in names if has_predecessor[lv] and r_squared[lv] != 0])
            self._gof = float(np.sqrt(communality * r2.mean())) if r2.size else float("nan")
        else:
            self._gof = float("nan")

    def summary(self) -> pd.DataFrame:
        return self._table

    def goodness_of_fit(self) -> float:
        if math.isnan(self._gof):
            raise ValueError("Cannot calculate goodness-of-fit if all constructs are single-item.")
        return self._gof
